#!/bin/bash
# round 6: closed-loop robustness sweeps on the kernels with the one termination rule (status histograms over millions of closed-loop QPs; every route)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python tools/robustness_sweep.py 8192 10 12 > $O/r6q_robustness_N12_8192x10.txt 2>&1
timeout 600 python tools/robustness_sweep.py 1024 10 12 > $O/r6q_robustness_N12_1024x10.txt 2>&1
timeout 600 python tools/robustness_sweep.py 256 20 12 > $O/r6q_robustness_N12_256x20.txt 2>&1
timeout 900 python tools/robustness_sweep.py 4096 20 14 > $O/r6q_robustness_N14_4096x20.txt 2>&1
timeout 600 python tools/robustness_sweep.py 1024 40 14 > $O/r6q_robustness_N14_1024x40.txt 2>&1
for f in $O/r6q_robustness_*.txt; do echo $f; tail -2 $f; done
