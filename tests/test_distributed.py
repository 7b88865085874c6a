"""world_size-2 CPU tests (gloo) of the multi-GPU path: batch sharding, barrier, MAX-reduction of the step time, and that the
sharded result equals the unsharded one.  The per-rank compute stand-in is the CPU oracle (there is no GPU here); on the GPU
box bench.py runs the same Group/shard code with the HIP path and backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
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


# ---- bench.py --config c4 / c5: a FIXED global batch sharded over the ranks (BASELINE configs[3], [4]) ----

def _c4_worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import bench
    import oracle_lib as O
    from shadernn_amd import dist, models

    g = dist.Group(backend="gloo")
    shard = bench.shard_plan("c4", world, rank, micro=32)  # the split bench.py times: 256 images -> 128 per rank (here as micro-batches of 32)
    assert shard["global_batch"] == 256 and shard["images"] == 128 and shard["micro_sizes"] == [32] * 4 and shard["first_image"] == 128 * rank
    assert bench.shard_plan("c4", world, rank)["micro_sizes"] == [128]  # default: a rank's share in one pass
    # compute stand-in (no GPU here): the oracle on a width-reduced MobileNetV2 at 16x16, image i = f(i) so the shard identity is checkable
    net = models.mobilenetv2(seed=1, num_classes=4, width_mult=0.25)
    rng = np.random.default_rng(7)
    batch = rng.random((shard["global_batch"], 16, 16, 3), dtype=np.float32)
    sums = np.zeros(shard["global_batch"], np.float32)
    pos = shard["first_image"]
    g.barrier()
    for mb in shard["micro_sizes"]:
        y = O.forward(net, batch[pos : pos + mb]).reshape(mb, -1)
        sums[pos : pos + mb] = (y * np.arange(1, y.shape[1] + 1)).sum(axis=1)
        pos += mb
    g.barrier()
    parts = g.gather_arrays(sums)  # check only: the timed path gathers nothing
    n_total = g.sum_over_ranks(shard["images"])
    if rank == 0:
        merged = np.sum(parts, axis=0)  # every image is written by exactly one rank, zeros elsewhere
        probe = [0, 31, 32, 127, 128, 200, 255]
        want = np.array([(O.forward(net, batch[i : i + 1]).reshape(-1) * np.arange(1, 5)).sum() for i in probe], np.float32)
        q.put((n_total, float(np.abs(merged[probe] - want).max()), int((merged != 0).sum())))
    g.barrier()
    g.close()


def test_c4_global_batch_shards_over_two_ranks_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c4_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n_total, err, nonzero = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n_total == 256.0 and nonzero == 256
    assert err < 1e-5            # an image's result does not depend on the rank / micro-batch it ran in


def test_shard_plan_every_config_and_world():
    import bench

    for cfgname, cfg in bench.CONFIGS.items():
        for world in (1, 2, 4, 8):
            plans = [bench.shard_plan(cfgname, world, r) for r in range(world)]
            assert sum(p["images"] for p in plans) == plans[0]["global_batch"]
            assert [p["first_image"] for p in plans] == [sum(q["images"] for q in plans[:r]) for r in range(world)]
            assert all(max(p["micro_sizes"]) <= cfg["micro"] and sum(p["micro_sizes"]) == p["images"] for p in plans)
            if "global" in cfg:
                assert plans[0]["global_batch"] == cfg["global"]  # strong scaling: total work fixed
            else:
                assert plans[0]["global_batch"] == world * cfg["per_rank"]
    assert bench.shard_plan("c4", 8, 3) == {"images": 32, "global_batch": 256, "first_image": 96, "micro_sizes": [32]}
    assert bench.shard_plan("c5", 8, 7) == {"images": 8, "global_batch": 64, "first_image": 56, "micro_sizes": [8]}
    assert bench.shard_plan("c5", 1, 0)["micro_sizes"] == [32, 32] and bench.shard_plan("c5", 1, 0, micro=16)["micro_sizes"] == [16] * 4
    with pytest.raises(SystemExit, match="REFUSED"):  # 64 images per pass: single tensors of 2^31 elements
        bench.shard_plan("c5", 1, 0, micro=64)


def test_bench_self_spawns_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes under torch.distributed.run: both ranks start (and, with no GPU in this
    container, both refuse to run: the HIP path has no CPU fallback)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    elif not torch.cuda.is_available():
        assert r.returncode != 0
        # (torchrun tears the other rank down as soon as the first one exits: both lines usually arrive, one always does)
        assert r.stderr.count("no GPU visible") >= 1, r.stderr[-2000:]


def test_bench_kernel_table_groups_by_function_and_prices_the_dominant_one(monkeypatch, tmp_path):
    """bench.py's `roofline` names the kernel FUNCTION with the most GPU time over all its launches (never a multi-launch step, never the longest
    single launch), prices it on the algorithmic work booked on it, and takes `traffic` from profiles/pmc_latest.json only for the INSTANTIATIONS
    the run launched and only from records taken with the kernel sources of this build."""
    import json

    import bench
    from shadernn_amd import fingerprint

    trace = {"launches": 14, "kernels": [
        {"function": "conv2d_wide_kernel", "launches": 8, "main_launches": 8, "total_ms": 4.0, "flops": 8e12, "bytes": 4e9, "instances": [
            {"name": "void snnhip::(anonymous namespace)::conv2d_wide_kernel<2, 2, 2, 2>(snnhip::(anonymous namespace)::WideParams)", "launches": 6, "main_launches": 6,
             "total_ms": 3.0, "flops": 6e12, "bytes": 3e9, "plans": ["conv2d_mfma_wide_f16 a"]},
            {"name": "void snnhip::(anonymous namespace)::conv2d_wide_kernel<4, 1, 1, 1>(snnhip::(anonymous namespace)::WideParams)", "launches": 2, "main_launches": 2,
             "total_ms": 1.0, "flops": 2e12, "bytes": 1e9, "plans": ["conv2d_mfma_wide_f16 b"]}]},
        {"function": "conv2d_upconv_kernel", "launches": 2, "main_launches": 2, "total_ms": 1.4, "flops": 1e12, "bytes": 3.5e9, "instances": [
            {"name": "void snnhip::(anonymous namespace)::conv2d_upconv_kernel<8, 2>(P)", "launches": 2, "main_launches": 2, "total_ms": 1.4, "flops": 1e12, "bytes": 3.5e9, "plans": ["up"]}]},
        {"function": "splitk_reduce_kernel", "launches": 4, "main_launches": 0, "total_ms": 0.1, "flops": 0.0, "bytes": 0.0, "instances": [
            {"name": "void snnhip::(anonymous namespace)::splitk_reduce_kernel<true, float>(int)", "launches": 4, "main_launches": 0, "total_ms": 0.1, "flops": 0, "bytes": 0, "plans": []}]},
    ]}
    rows = bench.kernel_table(trace, 2, bench.PEAK_F16_MFMA_TFLOPS)
    assert [r["function"] for r in rows] == ["conv2d_wide_kernel", "conv2d_upconv_kernel", "splitk_reduce_kernel"]  # by total time: the longest LAUNCH (0.7 ms, upconv) is not first
    assert rows[0]["bound"] == "mfma" and abs(rows[0]["frac"] - 8e12 / 4e-3 / 2500e12) < 1e-12
    assert rows[1]["bound"] == "hbm" and abs(rows[1]["frac"] - 3.5e9 / 1.4e-3 / 8e12) < 1e-12
    assert rows[2]["bound"] == "aux" and rows[2]["frac"] is None
    assert abs(sum(r["share_of_gpu_time"] for r in rows) - 1.0) < 1e-12 and rows[0]["launches_per_step"] == 4

    sha = fingerprint.csrc_sha16()
    pmc = {"conv2d_wide_kernel<2,2,2,2>": {"hbm_bytes_per_launch": 6e8, "csrc_sha16": sha, "git_head": "x"},
           "conv2d_wide_kernel<4,1,1,1>": {"hbm_bytes_per_launch": 2e8, "csrc_sha16": sha, "git_head": "x"},
           "conv2d_wide_kernel": {"hbm_bytes_per_launch": 1.0, "csrc_sha16": sha},  # the bare key is never used: it is one instantiation's record
           "conv2d_upconv_kernel<8,2>": {"hbm_bytes_per_launch": 9e8, "csrc_sha16": "0" * 16}}
    os.makedirs(tmp_path / "profiles")
    json.dump(pmc, open(tmp_path / "profiles" / "pmc_latest.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, src = bench.pmc_traffic(rows[0])
    assert abs(t - (6 * 6e8 + 2 * 2e8) / 8) < 1e-3 and "launch-weighted" in src
    t, src = bench.pmc_traffic(rows[1])
    assert t is None and src.startswith("stale")   # a kernel edit turns the field to null until the profile is re-taken, it does not lie
    t, src = bench.pmc_traffic(rows[2])
    assert t is None and "no PMC record" in src
    rec = bench.roofline_record(rows, bench.PEAK_F16_MFMA_TFLOPS, single_launch_step=False)
    assert rec["kernel"] == "conv2d_wide_kernel" and rec["bound"] == "mfma" and abs(rec["frac"] - rows[0]["frac"]) < 1e-12
    assert abs(rec["algorithmic_bytes_per_launch"] - 0.5e9) < 1 and abs(rec["traffic_over_algorithmic_bytes"] - 5e8 / 0.5e9) < 1e-9
    monkeypatch.undo()
    # a Winograd kernel: `frac` on the algorithmic count, `frac_executed` on what the matrix pipe multiplies ("mfma_flops=" of the plans, summed by the
    # trace), and the PMC pipe utilisation beside them when the committed record matches the build
    wino = {"launches": 2, "kernels": [{"function": "conv2d_wino_kernel", "launches": 2, "main_launches": 2, "total_ms": 0.1, "flops": 14.8e9, "bytes": 1e8, "mfma_flops": 6.6e9,
                                        "instances": [{"name": "void snnhip::(anonymous namespace)::conv2d_wino_kernel<2>(P)", "launches": 2, "main_launches": 2, "total_ms": 0.1,
                                                       "flops": 14.8e9, "bytes": 1e8, "mfma_flops": 6.6e9, "plans": ["conv2d_mfma_wino mfma_flops=3.3e9"]}]}]}
    wrows = bench.kernel_table(wino, 1, bench.PEAK_F32_MFMA_TFLOPS)
    assert abs(wrows[0]["frac"] - 14.8e9 / 1e-4 / 157.3e12) < 1e-9 and abs(wrows[0]["frac_executed"] - 6.6e9 / 1e-4 / 157.3e12) < 1e-9
    wrec = bench.roofline_record(wrows, bench.PEAK_F32_MFMA_TFLOPS, single_launch_step=False)
    assert abs(wrec["frac_executed"] - wrows[0]["frac_executed"]) < 1e-12 and abs(wrec["executed_mfma_flops_per_launch"] - 3.3e9) < 1
    assert "frac_executed" not in rows[0]  # a direct kernel executes what it is priced on
    # the committed file: one source fingerprint per record
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert committed and all(len(v.get("csrc_sha16", "")) == 16 and v["hbm_bytes_per_launch"] >= 0 for v in committed.values())
    assert len(fingerprint.csrc_sha16()) == 16


# ---- the N > 1 line is self-verifying: a census of (rank, device, PCI address) through the rendezvous store, RCCL refused on a shared device ----

def _census_worker(rank, world, port, q, backend, same_device):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    from shadernn_amd import dist

    ident = {"ordinal": 0 if same_device else rank, "pci_bus_id": "0000:05:00" if same_device else "0000:%02x:00" % (5 + rank), "uuid": None, "name": "test"}
    try:
        g = dist.Group(backend=backend, identity=ident)
    except dist.SharedDeviceError as e:
        q.put((rank, "refused", str(e)))
        return
    vals = g.gather_values(10.0 + rank)
    q.put((rank, "ok", {"census": g.census, "seen": g.collective_ranks, "vals": vals, "backend": g.backend}))
    g.barrier()
    g.close()


def _run_census(backend, same_device):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_census_worker, args=(r, 2, port, q, backend, same_device)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_census_reports_every_rank_and_the_collective_sees_them_all():
    got = _run_census("gloo", same_device=False)
    for rank, status, rec in got:
        assert status == "ok" and rec["backend"] == "gloo" and rec["seen"] == 2
        assert [c["rank"] for c in rec["census"]] == [0, 1]
        assert [c["device"]["pci_bus_id"] for c in rec["census"]] == ["0000:05:00", "0000:06:00"]
        assert rec["vals"] == [10.0, 11.0]  # every rank's own figure, in rank order


def test_two_gloo_ranks_may_share_a_device():
    got = _run_census("gloo", same_device=True)
    assert all(status == "ok" for _, status, _ in got)


def test_two_rccl_ranks_on_one_device_are_refused_before_rccl_is_touched():
    # backend "nccl" cannot even initialise on this CPU box: the refusal must come first, on both ranks, naming the colliding ranks
    got = _run_census("nccl", same_device=True)
    for rank, status, msg in got:
        assert status == "refused"
        assert "ranks 0 and 1" in msg and "0000:05:00" in msg and "one rank per device" in msg


def test_check_one_rank_per_device_rules():
    import pytest

    from shadernn_amd import dist

    mk = lambda r, host, pci, ordinal=0: {"rank": r, "host": host, "device": {"ordinal": ordinal, "pci_bus_id": pci, "uuid": None}}
    dist.check_one_rank_per_device([mk(0, "a", "0000:05:00"), mk(1, "a", "0000:06:00", 1)], "nccl")
    dist.check_one_rank_per_device([mk(0, "a", "0000:05:00"), mk(1, "b", "0000:05:00")], "nccl")  # same address on another host is another GPU
    dist.check_one_rank_per_device([mk(0, "a", "0000:05:00"), mk(1, "a", "0000:05:00")], "gloo")
    with pytest.raises(dist.SharedDeviceError):
        dist.check_one_rank_per_device([mk(0, "a", "0000:05:00"), mk(1, "a", "0000:06:00", 1), mk(2, "a", "0000:05:00")], "nccl")
    with pytest.raises(dist.SharedDeviceError):  # no PCI address known: fall back on the ordinal
        dist.check_one_rank_per_device([mk(0, "a", None, 3), mk(1, "a", None, 3)], "nccl")
