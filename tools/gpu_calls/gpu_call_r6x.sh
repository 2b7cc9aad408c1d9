#!/bin/bash
# round 6: the full step at batch sizes far beyond the tested 8192 (size arithmetic, work-buffer sizing against 288 GB)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( EXP_CERT=0 timeout 600 python tools/exp_bench.py 16384 65536 262144 2>&1 | tail -5 ) > $O/r6x_big_batches.txt
( EXP_CERT=0 EXP_N=40 timeout 600 python tools/exp_bench.py 32768 2>&1 | tail -5 ) >> $O/r6x_big_batches.txt
cat $O/r6x_big_batches.txt | cut -c1-600
