#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_runtime_kernel.py -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6s_pytest_rt.txt 2>&1
grep -n "passed\|failed\|through the runtime\|FAILED\|Error" $O/r6s_pytest_rt.txt | cut -c1-250 | tail -14
