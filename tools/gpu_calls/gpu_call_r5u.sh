#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; T=${TAG:-r5u}; rm -f $O/${T}_ab.txt
for rep in 1 2; do for L in ${LIBS:-v0 v1 v6}; do echo "== $L rep $rep" >> $O/${T}_ab.txt; LMPC_LIB=$(pwd)/racinglmpc_amd/liblmpc_hip_$L.so EXP_N=${EXP_N:-40} EXP_CERT=0 timeout 200 python tools/exp_bench.py ${SIZES:-512 1024 4096} >> $O/${T}_ab.txt 2>&1; done; done
cat $O/${T}_ab.txt
