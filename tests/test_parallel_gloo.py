"""CPU, world_size 2, gloo: sharding and the per-lap exchange give identical, deterministic results on all ranks."""
import os
import subprocess
import sys

import numpy as np

from tests import common

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from racinglmpc_amd import parallel
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = parallel.shard(11, rank, world)
rng = np.random.default_rng(100 + rank)
laps = []
for i in range(lo, hi):
    T = 120 + (7 * i) % 23                       # deterministic lap lengths, some ties across ranks
    laps.append((rng.normal(size=(T, 6)) + i, rng.normal(size=(T, 2)), rng.normal(size=(T, 6))))
best = parallel.exchange_laps(laps, K=4, T_max=160)
bc = parallel.broadcast_array(np.arange(6.0) + 10 * rank, src=0)
assert np.array_equal(bc, np.arange(6.0))
mx = parallel.allreduce_max(float(rank + 1))
np.savez(os.path.join(sys.argv[2], "rank%d.npz" % rank), lens=np.array([b[4] for b in best]), src=np.array([b[3] for b in best]),
         x0=np.array([b[0][0, 0] for b in best]), chk=np.array([b[0].sum() + b[1].sum() + b[2].sum() for b in best]), mx=mx, lo=lo, hi=hi)
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_covers_everything():
    from racinglmpc_amd import parallel
    for total in (0, 1, 7, 8, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_exchange_single_process():
    from racinglmpc_amd import parallel
    rng = np.random.default_rng(0)
    laps = [(rng.normal(size=(T, 6)), rng.normal(size=(T, 2)), rng.normal(size=(T, 6))) for T in (50, 40, 45, 40)]
    best = parallel.exchange_laps(laps, K=3, T_max=64)
    assert [b[4] for b in best] == [40, 40, 45]
    assert np.array_equal(best[0][0], laps[1][0]) and np.array_equal(best[1][0], laps[3][0])      # ties: lower local index first


def test_exchange_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", str(script), common.ROOT, str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    for k in ("lens", "src", "x0", "chk", "mx"):
        assert np.array_equal(r0[k], r1[k]), k
    assert float(r0["mx"]) == 2.0
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 6, 6, 11)
    lens_all = sorted(120 + (7 * i) % 23 for i in range(11))
    assert list(r0["lens"]) == lens_all[:4]
