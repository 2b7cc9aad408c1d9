#!/bin/bash
# final verification of round 5: the whole GPU suite, the rollout kernels' statistics, the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
( timeout 1200 python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5zz_pytest.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_ro && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -- python $R/tools/robustness_sweep.py 1024 2 12 > $R/$O/r5zz_rollout_run.txt 2>&1; cp $(find /tmp/prof_ro -name "*kernel_stats.csv" | head -1) $R/$O/r5zz_rollout_kernel_stats.csv )
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r5zz_bench.json 2> $O/r5zz_bench.err
tail -3 $O/r5zz_pytest.txt; grep -n "FAILED" $O/r5zz_pytest.txt | head; head -8 $O/r5zz_rollout_kernel_stats.csv
