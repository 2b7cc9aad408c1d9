#!/bin/bash
# round 6: which GPU test hangs? (per-test timeout, verbose)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -v -x --timeout=150 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > $O/r6g_pytest.txt 2>&1
grep -n "PASSED\|FAILED\|Timeout\|ERROR" $O/r6g_pytest.txt | tail -8 | cut -c1-200; tail -5 $O/r6g_pytest.txt | cut -c1-300
