#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 700 python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5f_pytest.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/r5f_bench.json 2> $O/r5f_bench.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/r5f_prof_rollout -o ro -- python $OLDPWD/tools/robustness_sweep.py 1024 2 12 ) > $O/r5f_rollout_prof.txt 2>&1
tail -4 $O/r5f_pytest.txt; grep -n "FAILED" $O/r5f_pytest.txt | head
