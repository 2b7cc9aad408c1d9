#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; rm -f $O/r5p_ab.txt
for rep in 1 2; do for L in q3 q0; do echo "== $L rep $rep" >> $O/r5p_ab.txt; LMPC_LIB=$(pwd)/racinglmpc_amd/liblmpc_hip_$L.so EXP_CERT=1 timeout 200 python tools/exp_bench.py 1 256 1024 4096 8192 >> $O/r5p_ab.txt 2>&1; done; done
cat $O/r5p_ab.txt
