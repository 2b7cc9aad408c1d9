"""CPU stand-in for the RCCL communicator of racinglmpc_amd/parallel.py: the same interface over torch.distributed's gloo
backend, so that the multi-rank logic (sharding, record layout, deterministic top-K, owner gathers) runs with world_size 2 in a
container without GPUs.  Test infrastructure only: the product package does not import torch."""
import numpy as np


class GlooComm:
    backend = "gloo"

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    def allreduce_max(self, values):
        import torch
        t = torch.from_numpy(np.atleast_1d(np.asarray(values, dtype=np.float64)).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()
