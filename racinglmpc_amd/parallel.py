"""Multi-GPU layer: one process per GPU, problems sharded by contiguous blocks, ONE exchange step per lap.

The QPs / rollouts of a batch are independent given a read-only safe set, so the data path needs no
collective; the only exchange is after a lap: every rank contributes its K fastest finished rollouts as fixed-stride
padded records, an all-gather (RCCL over xGMI when the process group's backend is "nccl", gloo on CPU) makes the
union visible everywhere, and every rank runs the same deterministic top-K selection and the same addTrajectory
inserts, which leaves identical lap stores on all ranks.  torch.distributed is used for the process group only.
"""
import numpy as np

REC_COLS = 14            # x (6) | u (2) | x_glob (6)


def shard(total, rank, world):
    """Contiguous block of `total` items owned by `rank` (first `total % world` ranks get one more)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    try:
        import torch.distributed as dist
    except Exception:
        return None
    return dist if (dist.is_available() and dist.is_initialized()) else None


def pack_laps(laps, K, T_max):
    """laps: list of (x (T,6), u (T,2), x_glob (T,6)[, extra (<=14,)]).  Keeps the K shortest (ties: lower local index),
    pads to T_max rows + 1 row for `extra` (e.g. the state right after the finish line).
    Returns (records float64 [K, T_max + 1, 14], lengths int64 [K]); unused slots have length -1."""
    order = sorted(range(len(laps)), key=lambda i: (laps[i][0].shape[0], i))[:K]
    rec = np.zeros((K, T_max + 1, REC_COLS)); ln = -np.ones(K, dtype=np.int64)
    for j, i in enumerate(order):
        x, u, xg = laps[i][0], laps[i][1], laps[i][2]
        T = x.shape[0]
        if T > T_max:
            raise ValueError("lap of %d steps exceeds the exchange record size %d" % (T, T_max))
        rec[j, :T, 0:6] = x; rec[j, :T, 6:8] = u; rec[j, :T, 8:14] = xg; ln[j] = T
        if len(laps[i]) > 3:
            e = np.asarray(laps[i][3], float).reshape(-1)
            rec[j, T_max, :e.shape[0]] = e
    return rec, ln


def exchange_laps(laps, K, T_max):
    """All-gather every rank's K fastest laps and return the global K fastest as [(x, u, x_glob, src_rank, T, extra)],
    ordered by (T, src_rank, local order) -- identical on every rank."""
    rec, ln = pack_laps(laps, K, T_max)
    dist = _dist()
    if dist is None:
        recs, lens = rec[None], ln[None]
    else:
        import torch
        world = dist.get_world_size()
        on_gpu = dist.get_backend() == "nccl"                   # "nccl" is RCCL on ROCm
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        t_rec = torch.from_numpy(rec).to(dev); t_len = torch.from_numpy(ln).to(dev)
        g_rec = [torch.empty_like(t_rec) for _ in range(world)]
        g_len = [torch.empty_like(t_len) for _ in range(world)]
        dist.all_gather(g_rec, t_rec)                           # ncclAllGather (RCCL) / gloo allgather
        dist.all_gather(g_len, t_len)
        recs = np.stack([t.cpu().numpy() for t in g_rec]); lens = np.stack([t.cpu().numpy() for t in g_len])
    cand = [(int(lens[r, j]), r, j) for r in range(recs.shape[0]) for j in range(K) if lens[r, j] >= 0]
    cand.sort()
    out = []
    for T, r, j in cand[:K]:
        out.append((recs[r, j, :T, 0:6].copy(), recs[r, j, :T, 6:8].copy(), recs[r, j, :T, 8:14].copy(), r, T, recs[r, j, T_max].copy()))
    return out


def broadcast_array(arr, src=0):
    """Every rank receives rank `src`'s array (same shape/dtype everywhere)."""
    dist = _dist()
    if dist is None:
        return np.array(arr)
    import torch
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def allreduce_max(value):
    dist = _dist()
    if dist is None:
        return float(value)
    import torch
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
