"""bench.py with more than one rank on the ONE GPU of the test box: the driver's N > 1 launch form (python -m torch.distributed.run ... bench.py --gpus N,
one process per rank, barrier + MAX-reduction around the timed region, rank 0 prints one JSON line) run end to end with `--backend gloo`, which lets
two ranks share device 0 (RCCL refuses that).  Everything but the collective backend is the code the 8-GPU run executes.  The reference is
single-device (core/src/ic2/vulkanBackend.cpp:30-31): this is the MI355X-side batch split of SURVEY 8e."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(ranks, extra, under_torchrun=True, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    tail = ["--gpus", str(ranks), "--also", "none", "--no-cpu-baseline", "--layer-table", "0"] + extra
    if under_torchrun:  # the driver's command
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's): %r" % (r.stdout[-2000:],)
    return json.loads(lines[0])


def test_two_ranks_on_one_device_c2_weak_scaling_line(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(2, ["--backend", "gloo", "--config", "c2", "--steps", "20", "--warmup", "5"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["global_batch"] == 2 and d["config"]["images_per_rank_per_step"] == 1 and d["config"]["backend"] == "gloo"
    assert d["parity"]["ok"] and d["parity"]["max_abs_err"] <= 1e-4
    # whole-job value: both ranks' images over the slowest rank's time; two ranks share one GPU here, so it is about the one-rank figure, not twice it
    assert abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["whole_step_frac"] <= 1.0


def test_two_ranks_on_one_device_c4_shards_the_global_batch(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(2, ["--backend", "gloo", "--config", "c4", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 256 and d["config"]["images_per_rank_per_step"] == 128 and d["config"]["micro_batches_per_rank"] == [128]
    assert d["parity"]["ok"]
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # every image of the global batch counted once


def test_one_rank_under_torch_distributed_run_matches_the_plain_run(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    args = ["--config", "c2", "--steps", "20", "--warmup", "5"]
    plain = _bench(1, args, under_torchrun=False)
    dist1 = _bench(1, args, under_torchrun=True)
    assert plain["n_gpus"] == dist1["n_gpus"] == 1
    assert abs(plain["ms_per_step"] - dist1["ms_per_step"]) <= 0.1 * plain["ms_per_step"], (plain["ms_per_step"], dist1["ms_per_step"])
