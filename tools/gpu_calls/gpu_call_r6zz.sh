#!/bin/bash
# round 6: the measurement set of the round on the final kernels (bench line, kernel stats, six PMC passes at batch 256 / 4096 and N = 40 / 1024, phase stamps, RCCL one-rank line)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 bash tools/collect_profiles.sh r6zz 256 4096 > gpurun_out/r6zz_collect.log 2>&1
HORIZON=40 timeout 900 bash tools/collect_profiles.sh r6zz_N40 1024 >> gpurun_out/r6zz_collect.log 2>&1
( cd /tmp && rm -rf /tmp/prof_ro && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -- python $OLDPWD/tools/robustness_sweep.py 1024 2 12 > $OLDPWD/gpurun_out/r6zz_rollout_run.txt 2>&1; cp $(find /tmp/prof_ro -name "*kernel_stats.csv" | head -1) $OLDPWD/gpurun_out/r6zz_rollout_kernel_stats.csv )
timeout 200 python tools/dropin_time.py > gpurun_out/r6zz_dropin_time.txt 2>&1
ls gpurun_out | grep r6zz | wc -l; head -5 gpurun_out/r6zz_B256_kernel_stats.csv | cut -c1-200; python tools/show_bench.py gpurun_out/r6zz_bench.json | cut -c1-300
