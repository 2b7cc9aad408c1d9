#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout 600 python tools/robustness_sweep.py 8192 25 12 > $O/r5z_robustness_N12_8192x25.txt 2>&1
for s in 5 6 7; do timeout 400 python tools/lmpc_40laps.py --flow dropin --laps 40 --horizon 14 --seed $s --out $O/r5z_40laps_dropin_N14_seed$s.json > $O/r5z_40laps_dropin_N14_seed$s.txt 2>&1; done
timeout 400 python tools/lmpc_40laps.py --flow dropin --laps 40 --horizon 12 --seed 5 --out $O/r5z_40laps_dropin_N12_seed5.json > $O/r5z_40laps_dropin_N12_seed5.txt 2>&1
timeout 300 python tools/dropin_time.py > $O/r5z_dropin_time.txt 2>&1
tail -n 3 $O/r5z_robustness_N12_8192x25.txt; tail -n 2 $O/r5z_40laps_dropin_*.txt; tail -5 $O/r5z_dropin_time.txt
