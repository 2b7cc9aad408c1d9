#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python tools/rt_kernel_rate.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|warn" > $O/r6j_rt_kernel_rate.txt
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "context_pool" -p no:cacheprovider 2>&1 | tail -3
cat $O/r6j_rt_kernel_rate.txt
