#!/bin/bash
# round 6: guard zones behind every device buffer (liblmpc_hip_guard.so) under the tests that touch the new entry points too (runtime kernel, rollout capture, regress_points, context pool)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
export LMPC_LIB=$(pwd)/racinglmpc_amd/liblmpc_hip_guard.so LMPC_GUARD_REPORT=1
( timeout 1200 python -m pytest tests/test_gpu_stores.py tests/test_gpu_retry.py tests/test_gpu_dropin_main.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_runtime_kernel.py tests/test_gpu_closed_loop.py \
    -m gpu -q -k "stores or retry or dropin or rollout or generations or status or 30_lap or horizons or edge or runtime or compiler or points or pool or n12_rollouts" -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -8 ) > $O/r6m_guard.log 2>&1
cat $O/r6m_guard.log | cut -c1-200
