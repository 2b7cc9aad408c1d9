#!/bin/bash
# round 6: re-tuned step-rule constants -- GPU suite, bench, rollout kernel statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
( timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6t_pytest.txt 2>&1
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r6t_bench.json 2> $O/r6t_bench.err
( cd /tmp && rm -rf /tmp/prof_ro && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -- python $R/tools/robustness_sweep.py 1024 2 12 > $R/$O/r6t_rollout_run.txt 2>&1; cp $(find /tmp/prof_ro -name "*kernel_stats.csv" | head -1) $R/$O/r6t_rollout_kernel_stats.csv )
grep -n "passed\|failed" $O/r6t_pytest.txt | tail -2; grep -n "FAILED" $O/r6t_pytest.txt | head; python tools/show_bench.py $O/r6t_bench.json | cut -c1-330; head -3 $O/r6t_rollout_kernel_stats.csv | cut -c1-170
grep -n "seed .*IPM iterations max\|N = 1[24], \(drop\|1024\)" $O/r6t_pytest.txt | cut -c1-330 | head -12
