#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; rm -f $O/r5k_ab.txt
for rep in 1 2; do for L in p0 p1 p2 p3; do echo "== $L rep $rep" >> $O/r5k_ab.txt; LMPC_LIB=$(pwd)/racinglmpc_amd/liblmpc_hip_$L.so EXP_N=40 EXP_CERT=0 timeout 200 python tools/exp_bench.py 256 512 1024 >> $O/r5k_ab.txt 2>&1; done; done
cat $O/r5k_ab.txt
