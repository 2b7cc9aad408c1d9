#!/bin/bash
# round 6: device-memory test (create / solve / rollout session / destroy cycles)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "non_finite or diverged" -rA -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -30 ) > $O/r6u_pytest.txt 2>&1
cat $O/r6u_pytest.txt | cut -c1-300
