#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 300 python tools/n40_experiments.py run trace40 gram40 ) > $O/r5c_n40.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -rA -k "every" 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r5c_pytest.txt 2>&1
tail -3 $O/r5c_pytest.txt; grep "^N40" $O/r5c_n40.txt | cut -c1-700
