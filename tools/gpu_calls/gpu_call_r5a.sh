#!/bin/bash
# round 5, GPU call A: N = 40 experiments + EXEC audit, full GPU test suite, bench line, closed-loop kernel trace, REG_SINGULAR sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 420 python tools/n40_experiments.py run ) > $O/r5a_n40.txt 2>&1
( timeout 240 python tools/n40_experiments.py audit ) > $O/r5a_audit.txt 2>&1
( timeout 600 python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5a_pytest.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/r5a_bench.json 2> $O/r5a_bench.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/r5a_prof_rollout -o ro -- python $OLDPWD/tools/robustness_sweep.py 1024 2 12 ) > $O/r5a_rollout_prof.txt 2>&1
( timeout 200 python tools/robustness_sweep.py 768 10 12 ) > $O/r5a_robust_768x10.txt 2>&1
tail -3 $O/r5a_pytest.txt; tail -c 600 $O/r5a_n40.txt; tail -2 $O/r5a_audit.txt; tail -2 $O/r5a_robust_768x10.txt
find $O/r5a_prof_rollout -name "*kernel_stats.csv" | head -2
