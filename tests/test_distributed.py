"""world_size-2 CPU tests (gloo) of the multi-GPU path: batch sharding, barrier, MAX-reduction of the step time, and that the
sharded result equals the unsharded one.  The per-rank compute stand-in is the CPU oracle (there is no GPU here); on the GPU
box bench.py runs the same Group/shard code with the HIP path and backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from shadernn_amd import dist, models

    g = dist.Group(backend="gloo")
    assert g.world == world and g.rank == rank and g.backend == "gloo"
    net = models.espcn_weights(seed=1)
    batch = np.random.default_rng(5).random((5, 12, 16, 1), dtype=np.float32)  # 5 images over 2 ranks: 3 + 2
    lo, hi = dist.shard_range(len(batch), world, rank)
    g.barrier()
    y_local = O.forward(net, batch[lo:hi])
    g.barrier()
    t_max = g.max_over_ranks(1.0 + rank)          # the slowest rank defines the step time
    n_total = g.sum_over_ranks(hi - lo)
    # gather (padded to the largest shard) only to CHECK the sharding; the timed path never gathers images
    pad = np.zeros((3,) + y_local.shape[1:], np.float32)
    pad[: hi - lo] = y_local
    parts = g.gather_arrays(pad)
    if rank == 0:
        full = O.forward(net, batch)
        got = np.concatenate([parts[r][: dist.shard_range(5, world, r)[1] - dist.shard_range(5, world, r)[0]] for r in range(world)])
        q.put((t_max, n_total, float(np.abs(got - full).max()), (lo, hi)))
    g.barrier()
    g.close()


def test_two_rank_batch_split_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t_max, n_total, err, rng0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t_max == 2.0          # MAX over ranks of (1 + rank)
    assert n_total == 5.0        # every image processed exactly once
    assert rng0 == (0, 3)
    assert err == 0.0            # sharded == unsharded, bit for bit (no cross-image dependence)


def test_shard_range_covers_everything_once():
    from shadernn_amd import dist

    for n in (1, 5, 8, 64, 255, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [dist.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
