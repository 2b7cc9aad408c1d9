#!/bin/bash
# round 6 A / B: the corrector's feed-forward pass of the two-wave kernel split over both waves (-DLMPC_AB_PHI2)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=$(pwd)
rm -f $O/r6y_ab.txt
for rep in 1 2; do
  bash tools/ab_bench.sh r6y racinglmpc_amd/liblmpc_hip.so racinglmpc_amd/liblmpc_hip_phi2.so 257 512 768 1024 2048 > /dev/null
done
for L in liblmpc_hip.so liblmpc_hip_phi2.so; do
  ( LMPC_LIB=$R/racinglmpc_amd/$L timeout 300 python bench.py --rollouts-only --steps 10 --warmup 3 ) > $O/r6y_ro_$L.json 2> $O/r6y_ro_$L.err
  python tools/show_bench.py $O/r6y_ro_$L.json | grep -i "rollout\|closed" | cut -c1-300
done
( LMPC_LIB=$R/racinglmpc_amd/liblmpc_hip_phi2.so timeout 900 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ) > $O/r6y_pytest.txt 2>&1
cat $O/r6y_ab.txt; cat $O/r6y_pytest.txt
