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
    assert abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 2e-4 * d["value"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["whole_step_frac"] <= 1.0
    # the self-verifying part of the N > 1 line: who ran where, what the collective backend saw, every rank's own step time beside the MAX
    assert d["config"]["collective_ranks_seen"] == 2 and d["config"]["rccl_ranks"] is None  # gloo here; under the driver's nccl launch rccl_ranks == N
    assert [r["rank"] for r in d["config"]["ranks"]] == [0, 1] and all(r["device"] == 0 for r in d["config"]["ranks"])
    assert d["config"]["ranks"][0]["pci"] == d["config"]["ranks"][1]["pci"] and d["config"]["ranks"][0]["pci"]
    assert len(d["ms_per_step_of_each_rank"]) == 2 and all(0.3 * d["ms_per_step"] < t < 3 * d["ms_per_step"] for t in d["ms_per_step_of_each_rank"])
    assert len(json.dumps(d, separators=(",", ":"))) < 8192


def test_two_ranks_on_one_device_c4_shards_the_global_batch(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(2, ["--backend", "gloo", "--config", "c4", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 256 and d["config"]["images_per_rank_per_step"] == 128
    detail = json.load(open(os.path.join(ROOT, d["detail"])))
    assert detail["config"]["micro_batches_per_rank"] == [128]
    assert d["parity"]["ok"]
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 2e-4 * d["value"]  # every image of the global batch counted once


def test_one_rank_under_torch_distributed_run_matches_the_plain_run(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    args = ["--config", "c2", "--steps", "20", "--warmup", "5"]
    plain = _bench(1, args, under_torchrun=False)
    dist1 = _bench(1, args, under_torchrun=True)
    assert plain["n_gpus"] == dist1["n_gpus"] == 1
    assert abs(plain["ms_per_step"] - dist1["ms_per_step"]) <= 0.1 * plain["ms_per_step"], (plain["ms_per_step"], dist1["ms_per_step"])


def test_two_rccl_ranks_on_the_one_device_exit_with_a_one_line_reason(built):
    """the driver's launch form with backend nccl on a one-GPU box: both ranks resolve to device 0 -> refused before RCCL is touched, exit code 5"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a one-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--also", "none", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    reasons = [l for l in r.stderr.splitlines() if l.startswith("bench.py: REFUSED:")]
    assert reasons and "ranks 0 and 1" in reasons[0] and "one rank per device" in reasons[0], r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no line: nothing was measured


def test_the_drivers_exact_command_prints_one_parsable_line_under_8_kb(built):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (BENCH_rNN.json's cmd): ONE line on stdout, < 8192 bytes (BENCH_r05.json: 22.7 KB -> parsed: null), with
    the standard keys + roofline + cpu_baseline + the four other configs; the full record lands in bench_detail.json"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1, r.stdout[-1000:]
    assert len(out[0]) < 8192 and len(r.stdout) < 8081  # the whole of stdout fits the driver's tail window
    d = json.loads(out[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "images/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["metric"].startswith("images/sec (1080p ESPCN 2x)") and "configs[1]" in d["config"]["workload"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] in ("mfma", "hbm") and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["whole_step_frac"] <= 1.0
    assert rf["sum_of_kernel_durations_ms"] <= d["ms_per_step"] * 1.05
    assert [c["id"] for c in rf["other_configs"]] == ["c1", "c3", "c4", "c5"] and all(c["parity_ok"] for c in rf["other_configs"])
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert d["parity"]["ok"] and d["parity"]["max_abs_err"] <= 1e-4
    detail = json.load(open(os.path.join(ROOT, d["detail"])))
    for k in ("kernels", "layer_table", "wait_semantics", "configs"):
        assert k in detail and k not in d
    assert abs(detail["value"] - d["value"]) <= 1e-4 * d["value"]
