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


class SharedDeviceError(RuntimeError):
    """two ranks of an RCCL job resolved to the same physical GPU"""


def device_identity(ordinal):
    """What tells two physical GPUs apart on one node: the PCI address (domain:bus:device) and, where the runtime gives one, the UUID."""
    import torch

    if not torch.cuda.is_available():
        return {"ordinal": ordinal, "pci_bus_id": None, "uuid": None, "name": None}
    p = torch.cuda.get_device_properties(ordinal)
    pci = None
    if hasattr(p, "pci_bus_id"):
        pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    uuid = str(getattr(p, "uuid", "")) or None
    return {"ordinal": ordinal, "pci_bus_id": pci, "uuid": uuid, "name": p.name}


def check_one_rank_per_device(census, backend):
    """RCCL needs one rank per physical GPU (it aborts, or hangs, on a duplicate); so does the batch split's throughput claim.  `census` = every rank's
    {rank, host, device: device_identity()}.  Raises SharedDeviceError naming the ranks that collide -- on every rank alike, BEFORE the first RCCL call.
    gloo jobs may share a device (that is how the N > 1 path is exercised on a one-GPU box)."""
    if backend != "nccl":
        return
    seen = {}
    for c in census:
        d = c["device"]
        key = (c.get("host"), d.get("pci_bus_id") or d.get("uuid") or ("ordinal", d.get("ordinal")))
        if key in seen:
            raise SharedDeviceError("ranks %d and %d both run on GPU %s (ordinal %s) of host %s: RCCL needs one rank per device -- launch one rank per GPU "
                                    "(LOCAL_RANK = device ordinal) or use --backend gloo" % (seen[key], c["rank"], key[1], d.get("ordinal"), c.get("host")))
        seen[key] = c["rank"]


class Group:
    def __init__(self, backend=None, device=None, local_device=None, identity=None):
        """identity: this rank's device_identity() (tests hand in made-up ones); default = that of `local_device`."""
        import socket

        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world = env_rank_world()
        self.device = device
        # the GPU this rank runs on: LOCAL_RANK under the driver's launch (one rank per GPU); several gloo ranks may share a device
        self.local_device = self.local_rank if local_device is None else local_device
        self.active = self.world > 1
        me = {"rank": self.rank, "local_rank": self.local_rank, "host": socket.gethostname(), "pid": os.getpid(),
              "device": identity if identity is not None else device_identity(self.local_device)}
        self.census = [me]
        self.collective_ranks = 1
        if self.active and not dist.is_initialized():
            import json

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            # the rendezvous store first (plain TCP; under torch.distributed.run it is the agent's): every rank publishes which GPU it took, and the
            # one-rank-per-device rule is checked before RCCL is touched -- no device_id here, so the communicator is built by the first collective
            store, _, _ = next(iter(dist.rendezvous("env://", rank=self.rank, world_size=self.world)))
            store.set("snn_census/%d" % self.rank, json.dumps(me))
            self.census = [json.loads(store.get("snn_census/%d" % r).decode()) for r in range(self.world)]
            check_one_rank_per_device(self.census, backend)
            dist.init_process_group(backend=backend, store=dist.PrefixStore("snn_pg", store), rank=self.rank, world_size=self.world)
        self.backend = dist.get_backend() if self.active else None
        if self.active:
            # proof on the line that the collective backend really spans the job: a SUM of ones through it (RCCL when backend == "nccl")
            self.collective_ranks = int(round(self.sum_over_ranks(1.0)))
            if self.collective_ranks != self.world or dist.get_world_size() != self.world:
                raise RuntimeError("%s all-reduce saw %d ranks, WORLD_SIZE is %d" % (self.backend, self.collective_ranks, self.world))

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

    def gather_values(self, value):
        """every rank's own float, in rank order (the per-rank ms_per_step beside the MAX)"""
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._dev())
        if not self.active:
            return [float(value)]
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [float(o.item()) for o in outs]

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
