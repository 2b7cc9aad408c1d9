#!/bin/bash
# round 6, final GPU suite (summary lines kept)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1000 python -m pytest tests/ -x -q -m gpu -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6_final_pytest.txt 2>&1
grep -n "passed\|failed" $O/r6_final_pytest.txt | tail -2
