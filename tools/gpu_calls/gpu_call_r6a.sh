#!/bin/bash
# round 6, first call: the one termination rule (step_bound_ok) through the whole GPU suite + the new N = 12 closed-loop tests on both routes, then the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
( timeout 1500 python -m pytest tests -m gpu -q -rA -s 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6a_pytest.txt 2>&1
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r6a_bench.json 2> $O/r6a_bench.err
tail -3 $O/r6a_pytest.txt; grep -n "FAILED\|Error" $O/r6a_pytest.txt | head -20; grep -n "N = 12\|N = 14" $O/r6a_pytest.txt | cut -c1-400 | head -20
python tools/show_bench.py $O/r6a_bench.json 2>/dev/null | head -30
