#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5x_pytest.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/r5x_bench.json 2> $O/r5x_bench.err
tail -4 $O/r5x_pytest.txt; grep -n "FAILED" $O/r5x_pytest.txt | head
