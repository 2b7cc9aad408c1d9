"""BASELINE config 4 ("batch=8192 parallel rollouts sharded over the GPUs of a node, all-gather per lap"):
device-resident closed-loop LMPC laps + the per-lap exchange.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/rollout_bench.py --rollouts 8192
    python tools/rollout_bench.py --rollouts 1024            # single GPU
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=1024)
    ap.add_argument("--laps", type=int, default=2)
    ap.add_argument("--keep", type=int, default=4, help="K fastest laps exchanged and appended to the stores per lap")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from racinglmpc_amd import _capi, parallel, rollout
    from tests import common
    g = common.load_lmpc_golden()
    lo, hi = parallel.shard(args.rollouts, rank, world)
    cfg, par = common.lmpc_config(g, 12, max_batch=max(hi - lo, 1), device=local)
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=100 + rank)
    B = args.rollouts
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.1, 0.1, B); x0[:, 0] += np.linspace(0.0, 0.1, B)
    xLin0 = g["xPID"][1:14]; uLin0 = g["uPID"][1:13]
    out = []
    gen = rollout.LmpcGeneration(ro, B, K=args.keep, T_max=400, ext=40, rank=rank, world=world)
    for lap in range(args.laps):
        t0 = time.perf_counter()
        best = gen.run(x0, xLin0, uLin0)
        dt = parallel.allreduce_max(time.perf_counter() - t0)
        steps = max(b[4] for b in best)
        out.append(dict(lap=lap, seconds=dt, best_lap_steps=[b[4] for b in best], src_ranks=[b[3] for b in best],
                        approx_qp_solves_per_s=args.rollouts * steps / dt))
    if rank == 0:
        print(json.dumps(dict(rollouts=args.rollouts, n_gpus=world, laps=out)))
    ctx.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
