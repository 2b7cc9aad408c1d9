#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
for r in 10 400; do
  echo "=== N = 40, problem $r, [A_k | B_k] in global memory (the batch-1024 route)"; PT_N=40 timeout 300 python tools/phase_timing.py $r
  echo "=== N = 40, problem $r, [A_k | B_k] in LDS (LMPC_NO_ABG=1)"; PT_N=40 LMPC_NO_ABG=1 timeout 300 python tools/phase_timing.py $r
done > $O/r5t_phase_timing_n40.txt 2>&1
cat $O/r5t_phase_timing_n40.txt
