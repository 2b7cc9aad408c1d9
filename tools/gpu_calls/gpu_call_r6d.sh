#!/bin/bash
# round 6: LMPC_ACC_TOL = 1e-7 -- whole GPU suite, closed-loop probes (profiles/r6_closed_loop_oracle_probe.txt), bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -rA -s 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6d_pytest.txt 2>&1
( for N in 12 14; do for s in 5 6 7; do timeout 900 python tools/closed_loop_oracle_probe.py $s 10 40 $N 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|developer knob"; done; done
  for s in 5 6 7; do timeout 900 python tools/closed_loop_oracle_probe.py rollouts $s 12 1024 3 4 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|developer knob"; done ) > $O/r6d_closed_loop_oracle_probe.txt 2>&1
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r6d_bench.json 2> $O/r6d_bench.err
tail -3 $O/r6d_pytest.txt; grep -n "^N = \|FAILED" $O/r6d_pytest.txt | cut -c1-420; grep "worst" $O/r6d_closed_loop_oracle_probe.txt | cut -c1-200
python tools/show_bench.py $O/r6d_bench.json 2>/dev/null | head -30
