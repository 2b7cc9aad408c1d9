#!/bin/bash
# round 6: predictor post in one reduction round (multi-wave kernels): phase timing, bench, GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
LMPC_TIMING_MW=1 timeout 300 python tools/phase_timing.py > $O/r6f_phase_timing_mw4.txt 2>&1
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r6f_bench.json 2> $O/r6f_bench.err
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/r6f_pytest.txt 2>&1
grep -v "^ROCm\|^Hostname\|^Librccl" $O/r6f_phase_timing_mw4.txt | head -24; python tools/show_bench.py $O/r6f_bench.json | head -3 | cut -c1-400; tail -3 $O/r6f_pytest.txt
