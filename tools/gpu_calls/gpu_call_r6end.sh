#!/bin/bash
# round 6, end of round on the FINAL kernels: measurement set (r6zz), robustness sweeps (r6q), GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
timeout 1500 bash tools/collect_profiles.sh r6zz 256 4096 > $O/r6zz_collect.log 2>&1
HORIZON=40 timeout 900 bash tools/collect_profiles.sh r6zz_N40 1024 >> $O/r6zz_collect.log 2>&1
( cd /tmp && rm -rf /tmp/prof_ro && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -- python $R/tools/robustness_sweep.py 1024 2 12 > $R/$O/r6zz_rollout_run.txt 2>&1; cp $(find /tmp/prof_ro -name "*kernel_stats.csv" | head -1) $R/$O/r6zz_rollout_kernel_stats.csv )
timeout 200 python tools/dropin_time.py > $O/r6zz_dropin_time.txt 2>&1
bash tools/gpu_calls/gpu_call_r6q.sh > /dev/null 2>&1
( timeout 1000 python -m pytest tests/ -x -q -m gpu -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6_final_pytest.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/r6_final_smoke.txt 2>&1
( time timeout 600 python bench.py ) > $O/r6_final_bench.json 2> $O/r6_final_bench.err
grep -n "passed\|failed" $O/r6_final_pytest.txt | tail -2; cat $O/r6_final_smoke.txt | cut -c1-250; head -3 $O/r6zz_B256_kernel_stats.csv | cut -c1-160; python tools/show_bench.py $O/r6_final_bench.json | head -2 | cut -c1-300
for f in $O/r6q_robustness_*.txt; do tail -2 $f | tr '\n' ' '; echo; done
