#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout 400 python tools/robustness_sweep.py 8192 10 12 > $O/r5y_robustness_N12_8192x10.txt 2>&1
timeout 400 python tools/robustness_sweep.py 1024 10 12 > $O/r5y_robustness_N12_1024x10.txt 2>&1
timeout 400 python tools/robustness_sweep.py 256 20 12 > $O/r5y_robustness_N12_256x20.txt 2>&1
timeout 500 python tools/robustness_sweep.py 4096 20 14 > $O/r5y_robustness_N14_4096x20.txt 2>&1
timeout 400 python tools/robustness_sweep.py 1024 40 14 > $O/r5y_robustness_N14_1024x40.txt 2>&1
timeout 400 python tools/robustness_sweep.py 256 20 14 > $O/r5y_robustness_N14_256x20.txt 2>&1
tail -n 3 $O/r5y_robustness_*.txt
