#!/bin/bash
# round 6: the closed-loop tests with the oracle's active-set finish; the 40-lap probes for profiles/r6_closed_loop_oracle_probe.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q -rA -s 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6c_pytest_closed_loop.txt 2>&1
( for N in 12 14; do for s in 5 6 7; do timeout 900 python tools/closed_loop_oracle_probe.py $s 10 40 $N 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|developer knob"; done; done
  for s in 5 6 7; do timeout 900 python tools/closed_loop_oracle_probe.py rollouts $s 12 1024 3 4 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|developer knob"; done ) > $O/r6c_closed_loop_oracle_probe.txt 2>&1
tail -3 $O/r6c_pytest_closed_loop.txt; grep -n "^N = \|FAILED" $O/r6c_pytest_closed_loop.txt | cut -c1-420; cat $O/r6c_closed_loop_oracle_probe.txt | cut -c1-300
