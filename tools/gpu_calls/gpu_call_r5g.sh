#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/r5g_ab.txt
for rep in 1 2; do for L in ab0 ab1 ab2 ab3; do echo "== $L rep $rep" >> $O/r5g_ab.txt; LMPC_LIB=$(pwd)/racinglmpc_amd/liblmpc_hip_$L.so EXP_CERT=0 timeout 200 python tools/exp_bench.py 1 256 1024 4096 >> $O/r5g_ab.txt 2>&1; done; done
cat $O/r5g_ab.txt
