"""Multi-GPU plumbing for the batch-split path: one process per GPU, weights replicated, images sharded,
NO data-path collective.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests) only carries the
barrier around the timed region, the MAX-reduction of the elapsed time and (optionally) a gather of per-rank checksums."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: GPU g of G gets images [g*B/G, (g+1)*B/G) (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Group:
    def __init__(self, backend=None, device=None, local_device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world = env_rank_world()
        self.device = device
        # the GPU this rank runs on: LOCAL_RANK under the driver's launch (one rank per GPU); several gloo ranks may share a device
        self.local_device = self.local_rank if local_device is None else local_device
        self.active = self.world > 1
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_device)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
        self.backend = dist.get_backend() if self.active else None

    def _dev(self):
        if self.backend == "nccl":
            return self.torch.device("cuda", self.local_device)
        return self.torch.device("cpu")

    def barrier(self):
        if self.active:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_device])
            else:
                self.dist.barrier()
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._dev())
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._dev())
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_arrays(self, arr):
        """all_gather of equally shaped float32 numpy arrays (test / checksum use only, never on the timed path)."""
        import numpy as np

        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self._dev())
        if not self.active:
            return [arr]
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]

    def close(self):
        if self.active and self.dist.is_initialized():
            self.dist.destroy_process_group()
