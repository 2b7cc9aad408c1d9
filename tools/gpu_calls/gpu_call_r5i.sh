#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 300 python -m pytest tests -m gpu -q -k "plant or rollout or every or certificate or closed_loop or reference_path" 2>&1 | tail -5 ) > $O/r5i_pytest.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/r5i_bench.json 2> $O/r5i_bench.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/r5i_prof_rollout -o ro -- python $OLDPWD/tools/robustness_sweep.py 1024 2 12 ) > $O/r5i_rollout_prof.txt 2>&1
tail -4 $O/r5i_pytest.txt
