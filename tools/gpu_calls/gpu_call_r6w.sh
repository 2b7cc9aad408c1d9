#!/bin/bash
# round 6 A / B: one four-wave work-group per CU forced by a padded LDS request (-DLMPC_AB_L4PAD) -- does the dispatcher already spread 256 work-groups over 256 CUs?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
rm -f $O/r6w_ab.txt
for rep in 1 2; do
  bash tools/ab_bench.sh r6w racinglmpc_amd/liblmpc_hip.so racinglmpc_amd/liblmpc_hip_l4pad.so 64 128 192 256 > /dev/null
done
cat $O/r6w_ab.txt
