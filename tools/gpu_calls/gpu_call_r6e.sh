#!/bin/bash
# round 6: the runtime-(N, S) solve kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_runtime_kernel.py -m gpu -q -rA -s -x 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6e_pytest_rt.txt 2>&1
tail -40 $O/r6e_pytest_rt.txt | cut -c1-300
