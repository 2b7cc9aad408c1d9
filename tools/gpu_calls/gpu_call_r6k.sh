#!/bin/bash
# round 6: rollout groups on separate streams -- identity test, rollout tests, closed-loop rate at 1 / 2 / 3 / 4 groups, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_stores.py -m gpu -q -k "rollout or exchange or multirank or stores or owner" -p no:cacheprovider 2>&1 | tail -4 ) > $O/r6k_pytest.txt 2>&1
cat $O/r6k_pytest.txt
for G in 1 2 3 4; do echo "== LMPC_RO_GROUPS=$G"; LMPC_RO_GROUPS=$G timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --rollouts-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for gen in d.get('config_rollouts', {}).get('generations', []): print('  generation %d: %.4f s, %d steps, %.0f closed-loop solves/s' % (gen['generation'], gen['seconds'], gen['simulated_steps'], gen['closed_loop_solves_per_s']))
"; done
