#!/bin/bash
# round 6: whole GPU suite (per-test limits from conftest)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6h_pytest.txt 2>&1
grep -n "passed\|failed" $O/r6h_pytest.txt | tail -3; grep -n "FAILED\|Timeout" $O/r6h_pytest.txt | head
