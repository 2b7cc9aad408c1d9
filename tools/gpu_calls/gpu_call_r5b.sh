#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 300 python tools/n40_experiments.py run trace40 gram40 ) > $O/r5b_n40.txt 2>&1
( timeout 600 python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5b_pytest.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/r5b_bench.json 2> $O/r5b_bench.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/r5b_prof_rollout -o ro -- python $OLDPWD/tools/robustness_sweep.py 1024 2 12 ) > $O/r5b_rollout_prof.txt 2>&1
tail -3 $O/r5b_pytest.txt; grep "^N40" $O/r5b_n40.txt | cut -c1-600
