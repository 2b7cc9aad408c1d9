#!/bin/bash
# round 6: whole GPU suite after the fork-safety fix (second-order weight 1.1 in the kernels)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1000 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6p_pytest.txt 2>&1
grep -n "passed\|failed" $O/r6p_pytest.txt | tail -2; grep -n "FAILED\|Fatal" $O/r6p_pytest.txt | head; grep -o "seed [0-9]: .*IPM iterations max [0-9]*" $O/r6p_pytest.txt | sed 's/: \[.*IPM/ IPM/'
grep -n "N = 1[24], \(drop\|1024\)" $O/r6p_pytest.txt | cut -c1-300 | head -8
