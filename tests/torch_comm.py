"""all_gather over a torch.distributed gloo group -- TEST INFRASTRUCTURE: drives
hashgan_amd.sharded.evaluate_shard with the NumPy stand-in engine in two CPU
processes (tests/test_sharded_gloo.py).  The product's communicator is
hashgan_amd.sharded.RcclComm (native RCCL through the C ABI, no torch)."""
import torch
import torch.distributed as dist


class TorchComm:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        flat = t.contiguous().view(-1)
        out = torch.empty(self.world * flat.numel(), dtype=t.dtype)
        dist.all_gather_into_tensor(out, flat, group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def all_to_all(self, t):
        """t = [world, ...]: block r goes to rank r -> [world, ...], block r = what rank r sent here."""
        t = t.contiguous()
        out = torch.empty_like(t)
        dist.all_to_all_single(out.view(-1), t.view(-1), group=self.group)
        return out

    def barrier(self):
        dist.barrier(group=self.group)

    def all_gather_host(self, arr):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return self.all_gather(t).numpy()
