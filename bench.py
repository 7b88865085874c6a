#!/usr/bin/env python
"""bench.py -- headline benchmark: images/sec for ESPCN 2x super-resolution, 1080p -> 4K, fp32 (BASELINE configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5] [--through host|capi]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` (N > 1) from a plain shell re-executes itself under torch.distributed.run with N ranks (one per GPU, RCCL).

Configs (BASELINE.json `configs`, SURVEY 8d shapes; default c2 = the configuration the metric is quoted on):
  c1  single 3x3 Conv2D, 1x224x224x3 -> 64, relu, fp32; one layer launch per step and rank                        (weak)
  c2  ESPCN 2x 1080p -> 4K, batch 1 per rank, fp32 (conv5x5 1->16 relu, conv3x3 16->16 relu, conv3x3 16->4, d2s+tanh)  (weak)
  c3  ResNet-18 224x224 fp32, batch 32 per rank                                                                    (weak)
  c4  MobileNetV2 224x224 fp32, GLOBAL batch 256 sharded over the ranks (dist.shard_range), a rank's share in one pass (strong)
  c5  Candy (fast-neural-style, the zoo graph) 720p fp16, GLOBAL batch 64 sharded over the ranks, micro-batches <= 32 (strong)
      (--micro N overrides; measured on one MI355X: c4 as 8 x 32 images 9.4 ms, 4 x 64 6.6 ms, 2 x 128 5.6 ms, 1 x 256 5.0 ms per step --
      the 14x14 / 7x7 layers of a 32-image batch do not fill 256 CUs; c5 as 4 x 16 images 24.8 ms, 2 x 32 23.6 ms -- half the launches, and the
      persistent 128 -> 128 kernel's 512 blocks walk 16.5 tiles each instead of 8.25, i.e. 17 / 16.5 rounds instead of 9 / 8.25; 64 images in one
      pass would put 2^31 elements into single tensors, beyond the 32-bit element offsets of the fp16 kernels: refused)
One "step" = one pass of the path over that batch, inputs already resident in HBM.  Every config runs through the C++ host mirror by
default (libsnn_core.so: the net written as the reference's .json + .bin model -> ModelParser -> MixedInferenceCore::create / run, fusion by
HipBackend::finalizeStages, the inference replayed as one recorded hipGraph); `--through capi` drives per-layer plans from Python instead.
`value` is quoted with the K inferences of the timed region IN FLIGHT on the rank's stream (RunParameters::deferSync, one wait at the end:
the serving mode of a batch-split replica); the reference's own semantics -- MixedInferenceCore::run waits once per inference, core.cpp:203 --
are timed right after it on the same model and reported as `sync_per_inference` (polling wait and blocking wait).
Before anything is timed, rank 0 downloads the GPU result of its first image and compares it with the CPU oracle's result for the same
input: `parity` {max_abs_err, max_rel_err, ...} goes on the line and a mismatch makes the run exit non-zero.
Multi-GPU = embarrassingly parallel batch split: no data-path collective; RCCL carries only the barrier / MAX-reduction of the elapsed time.
Rank 0 prints ONE JSON line.

Before the driver's `--warmup` steps a fixed time-based pre-heat (>= --preheat-ms of the workload, default 150 ms, reported as
`preheat_ms`) brings the clocks up, so a 20-step run does not time the ramp.

Timing: the timed region (K steps between two barriers + synchronisations, MAX over ranks) is repeated R times (--repeats; default 7 when K steps
take less than 50 ms, else 3) and `value` / `ms_per_step` are the MEDIAN; `timing` carries min / max / every repeat.

Extra objects on the line (prompt section 4):
  roofline     -- the kernel FUNCTION the step spends most of its GPU time in (summed over all its launches and template instantiations;
                  a step that launches a convolution and a statistics sweep is two entries, never one): the algorithmic flops (or HBM
                  bytes) of those launches / their summed duration, against the dense MFMA peak of the dtype (fp32 157.3 TFLOP/s, fp16
                  2500 TFLOP/s) or 8 TB/s HBM, whichever bounds that kernel.  Durations are measured live: a launch trace
                  (snnhip_trace_begin, include/snnhip.h) over a fixed number of inferences (--event-launches) run launch by launch right
                  after the timed region, every kernel stamped with its own dispatch start / end (what rocprofv3 --kernel-trace reports).
                  A fused plan is priced on what the fused launch itself has to move (its `hbm_bytes`), never on the per-layer sum.
                  `traffic` = HBM bytes per launch from the committed PMC passes (profiles/pmc_latest.json), only when that file was
                  taken with the kernel sources of this build (fingerprint match), else null.
  kernels      -- every kernel function of the step by share of GPU time, each with its own bound and fraction.
  configs      -- (default run: --config c2 on one GPU) compact records of the OTHER BASELINE configs c1, c3, c4, c5 measured the same way
                  in the same process: ms_per_step / images/s (median of R), in-run parity against the oracle, whole-step fraction,
                  dominant kernel with fraction and PMC traffic ratio (--also none switches them off).
  cpu_baseline -- the CPU oracle (kind "port": this repo's C restatement of the reference shaders; the reference has no CPU conv
                  path) timed on this host on a bounded sample of the same workload; for the configs with a Dense head also the
                  reference's own Eigen dense path (oracle/_ref/ref_dense, kind "reference") timed beside it.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 MFMA peak
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)

CONFIGS = {
    "c1": {"workload": "single 3x3 Conv2D 1x224x224x3->64 relu fp32 (BASELINE configs[0])", "dtype": "f32", "scaling": "weak", "per_rank": 1, "micro": 1,
           "hw": (224, 224), "cin": 3},
    "c2": {"workload": "ESPCN 2x super-resolution 1080p->4K, batch 1 per GPU, fp32 (BASELINE configs[1])", "dtype": "f32", "scaling": "weak", "per_rank": 1,
           "micro": 1, "hw": (1080, 1920), "cin": 1},
    "c3": {"workload": "ResNet-18 224x224 fp32, batch 32 per GPU (BASELINE configs[2])", "dtype": "f32", "scaling": "weak", "per_rank": 32, "micro": 32,
           "hw": (224, 224), "cin": 3},
    "c4": {"workload": "MobileNetV2 224x224 fp32, global batch 256 sharded over the GPUs (BASELINE configs[3])", "dtype": "f32", "scaling": "strong",
           "global": 256, "micro": 256, "hw": (224, 224), "cin": 3},
    "c5": {"workload": "Candy fast-neural-style (zoo graph) 720p fp16, global batch 64 sharded over the GPUs (BASELINE configs[4])", "dtype": "f16",
           "scaling": "strong", "global": 64, "micro": 32, "micro_max": 32, "hw": (720, 1280), "cin": 3},
}


# ------------------------------------------------------------------------------------------------ workloads

def make_net(config):
    from shadernn_amd import models

    if config == "c1":
        return models.single_conv(seed=1)
    if config == "c2":
        return models.espcn_weights(seed=1)
    if config == "c3":
        return models.resnet18(seed=1)
    if config == "c4":
        return models.mobilenetv2(seed=1)
    return models.zoo("candy-9_simplified-opt", (720, 1280, 3), seed=1)


def shard_plan(config, world, rank, micro=0):
    """(images of this rank per step, global batch, micro-batch sizes): c4 / c5 split a FIXED global batch with dist.shard_range
    (GPU g of G gets images [g*B/G, (g+1)*B/G)), the other configs give every rank the same share."""
    from shadernn_amd import dist as sdist

    cfg = CONFIGS[config]
    if "global" in cfg:
        lo, hi = sdist.shard_range(cfg["global"], world, rank)
        images, global_batch, first = hi - lo, cfg["global"], lo
    else:
        images, global_batch, first = cfg["per_rank"], cfg["per_rank"] * world, rank * cfg["per_rank"]
    if micro and micro > cfg.get("micro_max", 1 << 30):
        raise SystemExit("bench.py: REFUSED: --micro %d for %s: more than %d images per pass put 2^31 or more elements into single tensors (beyond the fp16 "
                         "kernels' 32-bit element offsets; a 64-image pass took a GPU box down)" % (micro, config, cfg["micro_max"]))
    sizes, left = [], images
    while left > 0:
        sizes.append(min(micro or cfg["micro"], left))
        left -= sizes[-1]
    return {"images": images, "global_batch": global_batch, "first_image": first, "micro_sizes": sizes}


class Workload:
    """The rank's share of one step: `images` images run as micro-batches through pre-built plans; run_device() only enqueues kernels."""

    def __init__(self, ctx, config, net, sizes, unfused=False):
        import shadernn_amd as snn

        cfg = CONFIGS[config]
        self.ctx = ctx
        self.images = sum(sizes)
        H, W = cfg["hw"]
        dtype = snn.F16 if cfg["dtype"] == "f16" else snn.F32
        self.runners, self.counts = [], []
        for mb in sorted(set(sizes), reverse=True):
            if config in ("c1", "c2"):
                r = snn.ChainRunner(ctx, net, mb, H, W, fused=(config == "c2" and not unfused))
                plans = list(r.plans)
            else:
                r = snn.GraphRunner(ctx, net, mb, H, W, dtype=dtype, fuse=not unfused)
                plans = [st[0] for st in r.steps]
            self.runners.append((r, plans))
            self.counts.append(sizes.count(mb))
        self.micro_sizes = sizes

    def upload(self, rng):
        import numpy as np

        for r, _ in self.runners:
            r.x.upload(rng.random(r.in_shape, dtype=np.float32))

    def first_input_output(self):
        """output of the first micro-batch after one synchronous pass (parity leg)."""
        r = self.runners[0][0]
        r.run_device()
        self.ctx.sync()
        return r.y.numpy()

    def run_device(self):
        for (r, _), n in zip(self.runners, self.counts):
            for _ in range(n):
                r.run_device()

    run_inflight = run_device

    def run_sync(self):
        self.run_device()
        self.ctx.sync()

    def sync(self):
        self.ctx.sync()

    def launch_by_launch(self, on):
        pass  # the ctypes runners never replay a graph

    def cost(self):
        f = b = 0.0
        for (r, _), n in zip(self.runners, self.counts):
            rf, rb = r.cost()
            f += n * rf
            b += n * rb
        return f, b

    def launches(self):
        return sum(n * sum(p.num_steps() for p in plans) for (_, plans), n in zip(self.runners, self.counts))

    def close(self):
        pass


class HostWorkload:
    """The same share of a step through the C++ host mirror (libsnn_core.so): the net is written as the reference's .json + .bin model,
    loaded by ModelParser / MixedInferenceCore at the micro-batch size (snn_model_create4), fused by HipBackend::finalizeStages, and every
    inference is MixedInferenceCore::run -- one recorded hipGraph launch + the reference's one sync per inference."""

    def __init__(self, config, net, sizes, device, tmpdir, unfused=False, capture=True):
        from shadernn_amd import host, models

        cfg = CONFIGS[config]
        self.images = sum(sizes)
        H, W = cfg["hw"]
        path = models.write_json(net, W, H, os.path.join(tmpdir, "%s_rank%d.json" % (config, device)), bin_weights=True)
        self.models, self.counts = [], []
        for mb in sorted(set(sizes), reverse=True):
            m = host.Model(path, W, H, cfg["cin"], device=device, fuse_chains=not unfused, prefer_half=cfg["dtype"] == "f16", capture_graph=capture, batch=mb)
            self.models.append(m)
            self.counts.append(sizes.count(mb))
        self.micro_sizes = sizes
        self.json_path = path
        self.model_args = dict(w=W, h=H, c=cfg["cin"], device=device, fuse_chains=not unfused, prefer_half=cfg["dtype"] == "f16")

    def upload(self, rng):
        import numpy as np

        for m in self.models:
            m.upload(rng.random((m.batch,) + tuple(m.in_shape[-3:]), dtype=np.float32))

    def layer_table(self, loops, drop=5):
        """The reference's benchmark table (demo/common/inferenceProcessor.cpp:84-86,143-199): `loops` inferences of a model built with the
        per-stage device timers (core.cpp:140-153,392-404, exported by MixedInferenceCore::writeTimeStat :437-442), the first `drop` (5)
        discarded, per-layer mean and POPULATION standard deviation in milliseconds.  A fused plan's time is booked on the last stage of its group."""
        import numpy as np

        from shadernn_amd import host

        a = self.model_args
        m = host.Model(self.json_path, a["w"], a["h"], a["c"], device=a["device"], fuse_chains=a["fuse_chains"], profiling=True, prefer_half=a["prefer_half"],
                       batch=self.models[0].batch)
        m.upload(np.random.default_rng(7767517).random((m.batch,) + tuple(m.in_shape[-3:]), dtype=np.float32))
        rows = {}
        for i in range(loops + drop):
            m.run()
            if i >= drop:
                for k, v in m.time_stats().items():
                    rows.setdefault(k, []).append(v)
        m.close()
        return [{"layer": k, "mean_ms": float(np.mean(v)), "std_ms": float(np.std(v)), "min_ms": float(np.min(v)), "max_ms": float(np.max(v))}
                for k, v in rows.items()]

    def first_input_output(self):
        m = self.models[0]
        m.run()
        return m.output()

    def run_inflight(self):
        """one step, enqueued only (RunParameters::deferSync): the stream keeps as many inferences in flight as the caller enqueues"""
        for m, n in zip(self.models, self.counts):
            for _ in range(n):
                m.run_async()

    def run_sync(self):
        """one step with the reference's semantics: MixedInferenceCore::run waits once per inference (core.cpp:203)"""
        for m, n in zip(self.models, self.counts):
            for _ in range(n):
                m.run()

    run_device = run_inflight

    def sync(self):
        for m in self.models:
            m.sync()

    def launch_by_launch(self, on):
        """a replayed hipGraph never calls the plans: traced inferences run launch by launch"""
        for m in self.models:
            m.suspend_replay(on)

    def cost(self):
        f = b = 0.0
        for m, n in zip(self.models, self.counts):
            mf, mb = m.cost()
            f += n * mf
            b += n * mb
        return f, b

    def launches(self):
        return sum(n * len(m.plan_steps()) for m, n in zip(self.models, self.counts))

    def close(self):
        for m in self.models:
            m.close()
        self.models = []


# ------------------------------------------------------------------------------------------------ CPU baseline

def oracle_input(config):
    """image 0 of rank 0's synthetic batch (U(0,1), seed 7767517): numpy fills a [B,H,W,C] draw image by image, so a 1-image draw IS image 0"""
    import numpy as np

    cfg = CONFIGS[config]
    H, W = cfg["hw"]
    return np.random.default_rng(7767517).random((1, H, W, cfg["cin"]), dtype=np.float32)


def cpu_baseline(config, net, timed_legs=True, compact=False):
    """The oracle ("port") on a bounded sample of the workload, single thread and all host threads; dense heads also on the reference's
    own Eigen path (oracle/_ref/ref_dense).  Returns (record or None, the oracle's result for image 0 of rank 0 -- the parity leg's expectation).
    compact = the `configs` block's leg: the parity pass itself is the all-thread sample (plus a short single-thread one where that is under a
    few seconds), the better of the two is `value`, labelled with the threads that gave it (inferenceProcessor.cpp:84-86: a mean over whole inferences)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import oracle_lib

    cfg = CONFIGS[config]
    cores = os.cpu_count() or 1
    x = oracle_input(config)
    fp16 = cfg["dtype"] == "f16"
    oracle_lib.forward(make_net("c2"), np.zeros((1, 32, 32, 1), np.float32))  # warm the library
    keep = {}

    def timed(threads, images):
        t0 = time.perf_counter()
        for _ in range(images):
            keep["y"] = oracle_lib.forward(net, x, threads=threads, fp16=fp16)
        return (time.perf_counter() - t0) / images

    # sample sizes chosen so the whole leg stays around 10-30 s of CPU work
    n_multi = {"c1": 20, "c2": 2, "c3": 4, "c4": 8, "c5": 1}[config] if timed_legs else 1
    n_single = {"c1": 5, "c2": 2, "c3": 1, "c4": 2, "c5": 0}[config] if timed_legs else 0
    if timed_legs and compact:
        n_multi = {"c1": 5, "c2": 1, "c3": 2, "c4": 2, "c5": 1}[config]
        n_single = {"c1": 2, "c2": 1, "c3": 1, "c4": 2, "c5": 0}[config]
    tn = timed(cores, n_multi)
    if not timed_legs:
        return None, keep["y"]
    t1 = timed(1, n_single) if n_single else None
    best_t, best_c = (tn, cores) if (t1 is None or tn <= t1) else (t1, 1)
    out = {"value": 1.0 / best_t, "unit": "images/s", "cores": best_c, "kind": "port",
           "sample": "%d full-size image(s) of %s (batch 1) through oracle/liboracle.so (C restatement of the reference shaders), all %d host threads%s"
                     % (n_multi, config, cores, (" + %d single-threaded" % n_single) if n_single else ""),
           "all_threads_images_per_s": 1.0 / tn, "single_thread_images_per_s": (1.0 / t1) if t1 else None, "host_threads": cores}
    if compact:
        out["sample"] = "%d image(s), all %d host threads%s; oracle/liboracle.so" % (n_multi, cores, (", %d single-threaded" % n_single) if n_single else "")
        return out, keep["y"]
    dense = [l for l in net["layers"] if l["type"] == "Dense"]
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dense")
    if dense and os.path.exists(ref):
        l = dense[-1]
        try:
            act = l["activation"] if l["activation"] else "-"
            sec = float(subprocess.run([ref, "--time", str(l["ic"]), str(l["units"]), act, "200"], stdout=subprocess.PIPE, text=True, timeout=120,
                                       check=True).stdout.split()[0])
            out["dense_reference"] = {"value": 1.0 / sec, "unit": "images/s (dense layer only)", "cores": 1, "kind": "reference",
                                      "sample": "200 calls of the reference's CPUCommonUtil<float> (core/src/ic2/cpulayer.h:136-171, Eigen) on the %dx%d %s head, "
                                                "batch 1, oracle/_ref/ref_dense --time" % (l["ic"], l["units"], l["activation"] or "linear"),
                                      "us_per_call": sec * 1e6}
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["dense_reference"] = {"error": repr(e)}
    return out, keep["y"]




def parity_record(config, got, want):
    """GPU result of image 0 vs the oracle's: fp32 configs at the north-star tolerance (|d| <= 1e-4 + 1e-4 |want| on every element), the fp16
    config with the acceptance of tests/test_configs_gpu.py::test_c5_* (quantised oracle; 99.9 % of the elements within 6e-3 of the output
    range, none beyond 6e-2: instance norms amplify half-precision rounding)."""
    import numpy as np

    cfg = CONFIGS[config]
    got = np.asarray(got, dtype=np.float32).reshape(want.shape)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    scale = max(1.0, float(np.abs(want).max()))
    rec = {"oracle": "oracle/liboracle.so (CPU restatement of the reference shaders; parity unpinned by reference fixtures, DESIGN.md 3)",
           "compared": "image 0 of rank 0, whole output tensor %s, downloaded before the timed region" % (list(want.shape),),
           "max_abs_err": float(d.max()), "max_rel_err": float((d / np.maximum(np.abs(want), 1e-4 if cfg["dtype"] == "f32" else 1e-2)).max()),
           "max_rel_err_definition": "max |gpu - oracle| / max(|oracle|, %s)" % ("1e-4" if cfg["dtype"] == "f32" else "1e-2"),
           "finite": bool(np.isfinite(got).all())}
    if cfg["dtype"] == "f32":
        rec["tolerance"] = "|d| <= 1e-4 + 1e-4 * |oracle| (north-star fp32 bound)"
        rec["ok"] = bool(rec["finite"] and (d <= 1e-4 + 1e-4 * np.abs(want)).all())
    else:
        e = d / scale
        rec["q999_err_over_range"] = float(np.quantile(e, 0.999))
        rec["tolerance"] = "fp16 storage vs the half-quantised oracle: q99.9(|d|) < 6e-3 * range and max |d| < 6e-2 * range"
        rec["ok"] = bool(rec["finite"] and rec["q999_err_over_range"] < 6e-3 and float(e.max()) < 6e-2)
    return rec


# ------------------------------------------------------------------------------------------------ the printed line

LINE_BUDGET = 6000  # bytes; the driver keeps a bounded tail of stdout (8 081 characters in BENCH_r05.json, whose 22.7 KB line came back parsed: null)


def _r(v, sig=5):
    """floats to `sig` significant digits (the line is a summary: the full-precision record is bench_detail.json)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v))
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def compact_line(detail, detail_path="bench_detail.json"):
    """The ONE line rank 0 prints: the driver's standard keys, `roofline` (dominant kernel of the headline config + one short row per other config),
    `cpu_baseline` and the in-run parity verdict -- nothing else.  In the spirit of the reference's own benchmark output, a compact table
    (demo/common/inferenceProcessor.cpp:143-199).  Everything else (kernels, layer_table, wait_semantics, the full `configs` block) stays in `detail`,
    which main() writes to bench_detail.json.  Pure function of the detail record (tests/test_bench_line.py builds it from a stored one)."""
    d = detail
    cfg = d.get("config") or {}
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value_mode"] = d.get("value_mode")
    line["config"] = {k: cfg[k] for k in ("workload", "config_id", "global_batch", "images_per_rank_per_step", "parallelism", "backend", "ranks_per_device",
                                          "launches_per_step", "device", "compute_units", "rccl_ranks", "collective_ranks_seen", "ranks") if k in cfg}
    t = d.get("timing") or {}
    line["timing"] = {k: t[k] for k in ("repeats", "min_ms_per_step", "max_ms_per_step") if k in t}
    if d.get("ms_per_step_of_each_rank"):
        line["ms_per_step_of_each_rank"] = d["ms_per_step_of_each_rank"]
    par = d.get("parity")
    line["parity"] = {k: par[k] for k in ("ok", "max_abs_err", "max_rel_err", "q999_err_over_range") if k in par} if par else None
    if par:
        line["parity"]["vs"] = "oracle/liboracle.so, image 0 (parity unpinned by reference fixtures)"
    rf = d.get("roofline")
    if rf:
        line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic_bytes", "kernel", "share_of_gpu_time",
                                               "launches_per_step", "avg_launch_us", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "frac_executed",
                                               "mfma_pipe_util_pmc", "bound_note", "arithmetic", "frac_of_split_peak", "whole_step_frac", "sync_per_inference_images_per_s",
                                               "sync_per_inference_blocking_wait_images_per_s", "whole_step_frac_sync_per_inference") if rf.get(k) is not None or k == "traffic"}
        for k in ("sum_of_kernel_durations_ms", "sum_of_launch_rooflines_ms"):
            if k in d:
                line["roofline"][k] = d[k]
        if rf.get("other_configs"):
            line["roofline"]["other_configs"] = rf["other_configs"]
    cb = d.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "single_thread_images_per_s") if cb.get(k) is not None}
        if len(line["cpu_baseline"].get("sample", "")) > 200:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:197] + "..."
    if d.get("errors"):
        line["errors"] = d["errors"]
    line["detail"] = detail_path
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_BUDGET:  # never lose the headline to a long list: drop the per-config rows first, then the optional objects
        for victim in (("roofline", "other_configs"), ("timing",), ("config", "ranks"), ("ms_per_step_of_each_rank",)):
            node = line
            for k in victim[:-1]:
                node = node.get(k) or {}
            if victim[-1] in node:
                node.pop(victim[-1])
                line["truncated"] = line.get("truncated", []) + [".".join(victim)]
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_BUDGET:
                break
    return text


def write_detail(detail, path=None):
    """the full record of the run (what the line used to carry): bench_detail.json beside bench.py, and a copy under gpurun_out/ when that exists
    (the directory gpurun merges back)"""
    paths = [path or os.path.join(ROOT, "bench_detail.json")]
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    done = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(detail, f, indent=1)
            done.append(p)
        except OSError as e:  # a read-only tree must not cost the line
            sys.stderr.write("bench.py: could not write %s: %r\n" % (p, e))
    return done


# ------------------------------------------------------------------------------------------------ main

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def csrc_fingerprint():
    from shadernn_amd import fingerprint

    return fingerprint.csrc_sha16()


def _short_kernel_name(full):
    """`void snnhip::(anonymous namespace)::conv2d_wide_kernel<4, 1, 4, 8>(Params, ...)` -> `conv2d_wide_kernel<4,1,4,8>`: the key tools/summarize_prof.py
    writes a PMC record under"""
    n = full.replace("void snnhip::(anonymous namespace)::", "").replace("snnhip::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:110].replace(" ", "")  # ([:110]: summarize_prof.py's cut -- mangled names of the fp16 kernels, which no demangler here resolves, are long)


def pmc_traffic(kernel):
    """HBM bytes per launch of a kernel function from the committed PMC passes (profiles/pmc_latest.json, written by tools/profile_gpu.sh ->
    summarize_prof.py, one record per template instantiation): the launch-weighted mean over the instantiations this run launched, only from
    records taken with the kernel sources of this build.  (traffic or None, where it came from)"""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_latest.json is missing"
    try:
        pmc = json.load(open(path))
    except Exception as e:
        return None, "profiles/pmc_latest.json unreadable: %r" % (e,)
    sha = csrc_fingerprint()
    tot = n = 0.0
    stale = missing = 0
    head = None
    for inst in kernel["instances"]:
        ent = pmc.get(_short_kernel_name(inst["name"]))
        if ent is None:
            missing += inst["launches"]
        elif ent.get("csrc_sha16") != sha:
            stale += inst["launches"]
            head = ent.get("csrc_sha16")
        else:
            tot += ent["hbm_bytes_per_launch"] * inst["launches"]
            n += inst["launches"]
            head = ent.get("git_head")
    if n and not missing and not stale:
        return tot / n, "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, kernel sources %s, commit %s), launch-weighted over %d instantiation(s)" % (
            sha, head, len(kernel["instances"]))
    if stale:
        return None, "stale: profiles/pmc_latest.json was taken with kernel sources %s, this build is %s" % (head, sha)
    return None, "no PMC record for this kernel's instantiations in profiles/pmc_latest.json"


def pmc_field(kernel, field):
    """launch-weighted mean of one field of the committed PMC records (profiles/pmc_latest.json) over a kernel function's instantiations; None when a
    record is missing or was taken with other kernel sources"""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return None
    sha = csrc_fingerprint()
    tot = n = 0.0
    for inst in kernel["instances"]:
        ent = pmc.get(_short_kernel_name(inst["name"]))
        if ent is None or ent.get("csrc_sha16") != sha or ent.get(field) is None:
            return None
        tot += ent[field] * inst["launches"]
        n += inst["launches"]
    return (tot / n) if n else None


SPLIT_TAG = "f16x3split"  # plan descriptions of the split-precision kernels (irb_fused.hip): fp32 operands as fp16 hi + lo, three f16 MFMA products per fp32 product
SPLIT_NOTE = ("pointwise stages with fp32 operands carried as fp16 hi + lo halves: three f16 MFMA products per fp32 product, fp32 accumulate (|error| <= 2^-21 per "
              "product); `frac` stays algorithmic fp32 flops against the fp32 MFMA peak, `frac_of_split_peak` is the same flops against the f16 pipe's 2500 / 3 TFLOP/s")


def _is_split(inst):
    return any(SPLIT_TAG in (pl or "") for pl in inst.get("plans", []))


def kernel_table(trace, inferences, peak_tf):
    """The launch trace (capi.trace_end) as one row per kernel FUNCTION, most GPU time first.  `flops` / `bytes` are the algorithmic work of the plan
    invocations whose main launch the function was (a fused plan: what the fused launch moves), per traced step; a function that only ever runs
    beside another plan's main kernel (split-K reduce, statistics fold, softmax) has none and no fraction."""
    total_ms = sum(k["total_ms"] for k in trace["kernels"]) or 1.0
    rows = []
    for k in trace["kernels"]:
        t = k["total_ms"] * 1e-3
        tf, tb = k["flops"] / (peak_tf * 1e12), k["bytes"] / (PEAK_HBM_GBPS * 1e9)
        row = {"function": k["function"], "launches_per_step": k["launches"] / inferences, "us_per_step": 1e3 * k["total_ms"] / inferences,
               "avg_launch_us": 1e3 * k["total_ms"] / max(1, k["launches"]), "share_of_gpu_time": k["total_ms"] / total_ms,
               "algorithmic_flops_per_step": k["flops"] / inferences, "algorithmic_bytes_per_step": k["bytes"] / inferences,
               "instances": [{"name": i["name"], "launches": i["launches"], "total_ms": i["total_ms"], "flops": i["flops"], "bytes": i["bytes"],
                              "mfma_flops": i.get("mfma_flops", 0.0), "plans": i["plans"][:4]} for i in sorted(k["instances"], key=lambda i: -i["total_ms"])]}
        if k["main_launches"] and t > 0 and (tf > 0 or tb > 0):
            row["bound"] = "mfma" if tf >= tb else "hbm"
            row["frac"] = max(tf, tb) / t
            row["achieved_tflops"] = k["flops"] / t / 1e12
            row["achieved_hbm_gbps"] = k["bytes"] / t / 1e9
            if k.get("mfma_flops", 0) > 0 and row["bound"] == "mfma":
                # Winograd / pre-summed-tap kernels execute fewer multiplies than the algorithmic count: `frac` is throughput on the reference's work,
                # `frac_executed` what the matrix pipe really does -- read headroom from this one
                done = k["mfma_flops"] + sum(i["flops"] for i in k["instances"] if not i.get("mfma_flops"))
                row["frac_executed"] = done / t / 1e12 / peak_tf
            util = pmc_field(k, "mfma_pipe_util")
            if util is not None:
                row["mfma_pipe_util_pmc"] = util
            split_flops = sum(i["flops"] for i in k["instances"] if _is_split(i))
            if split_flops > 0:
                row["arithmetic"] = "f16x3 split products, fp32 accumulate" + ("" if split_flops >= 0.999 * k["flops"] else " (%.0f %% of this function's flops)" % (100.0 * split_flops / k["flops"]))
                row["frac_of_split_peak"] = split_flops / t / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0)
        else:
            row["bound"], row["frac"] = "aux", None
        rows.append(row)
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows


def roofline_record(rows, peak_tf, single_launch_step):
    dom = rows[0]
    launches = sum(i["launches"] for i in dom["instances"]) or 1
    t = sum(i["total_ms"] for i in dom["instances"]) * 1e-3
    flops, nbytes = sum(i["flops"] for i in dom["instances"]), sum(i["bytes"] for i in dom["instances"])
    mfma = dom.get("bound") == "mfma"
    ach = (flops / t / 1e12) if mfma else (nbytes / t / 1e9)
    peak = peak_tf if mfma else PEAK_HBM_GBPS
    traffic, src = pmc_traffic(dom)
    rec = {"bound": "mfma" if mfma else "hbm", "achieved": ach, "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s", "frac": ach / peak,
           "traffic": traffic, "traffic_source": src, "kernel": dom["function"],
           "kernel_instances": [_short_kernel_name(i["name"]) for i in dom["instances"]],
           "dominant_by": "total GPU time over all launches of the function in the traced steps", "share_of_gpu_time": dom["share_of_gpu_time"],
           "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
           "algorithmic_flops_per_launch": flops / launches, "algorithmic_bytes_per_launch": nbytes / launches,
           "traffic_over_algorithmic_bytes": (traffic / (nbytes / launches)) if (traffic and nbytes) else None,
           "hbm_gbps_of_this_kernel": nbytes / t / 1e9, "plans": dom["instances"][0]["plans"][:3]}
    if single_launch_step and dom["avg_launch_us"] < 25.0:
        # a step that is ONE kernel of a few microseconds (c1: 196 blocks on 256 CUs, 8-11 us): what bounds it is the launch itself (dispatch,
        # wave start-up, the tail of a single round of blocks), not the memory system; the fraction stays quoted against HBM
        rec["bound"] = "launch"
        rec["bound_note"] = "single %.1f us kernel per step: launch / ramp bound; achieved and peak are the HBM figures" % dom["avg_launch_us"]
    if mfma and dom.get("frac_executed") is not None:
        # `achieved` uses the ALGORITHMIC flops of the direct convolutions (2*k*k*IC*OC per output pixel).  A Winograd F(2x2,3x3) kernel
        # multiplies 2.25x less (the ESPCN kernel also recomputes conv1 on the tile halo): `frac_executed` is what the matrix pipe really executes
        ex = sum((i.get("mfma_flops") or i["flops"]) for i in dom["instances"])
        rec["executed_mfma_flops_per_launch"] = ex / launches
        rec["executed_mfma_tflops"] = ex / t / 1e12
        rec["frac_executed"] = ex / t / 1e12 / peak_tf
    if dom.get("mfma_pipe_util_pmc") is not None:
        rec["mfma_pipe_util_pmc"] = dom["mfma_pipe_util_pmc"]
    if dom.get("arithmetic"):
        rec["arithmetic"], rec["frac_of_split_peak"], rec["split_peak"], rec["arithmetic_note"] = dom["arithmetic"], dom["frac_of_split_peak"], PEAK_F16_MFMA_TFLOPS / 3.0, SPLIT_NOTE
    return rec


def run_config(config, args, env, primary):
    """Builds `config`'s workload on this rank, checks image 0 against the oracle, times it (median of R timed regions of K steps) and traces its
    kernels.  Returns the record of the JSON line (primary) or a compact one (the `configs` block)."""
    import numpy as np
    import tempfile

    import shadernn_amd as snn
    import shadernn_amd.capi as capi_mod

    cfg = CONFIGS[config]
    rank, world, group, ctx, dev, info = env["rank"], env["world"], env["group"], env["ctx"], env["dev"], env["info"]
    steps = args.steps if (primary and args.steps is not None) else {"c1": 500, "c2": 200, "c3": 50, "c4": 20, "c5": 10}[config]
    warmup = args.warmup if (primary and args.warmup is not None) else {"c1": 50, "c2": 20, "c3": 5, "c4": 3, "c5": 2}[config]
    if not primary:
        steps = {"c1": 200, "c2": 100, "c3": 20, "c4": 8, "c5": 3}[config]
        warmup = {"c1": 20, "c2": 10, "c3": 3, "c4": 2, "c5": 1}[config]
    net = make_net(config)
    shard = shard_plan(config, world, rank, args.micro if primary else 0)
    images, global_batch = shard["images"], shard["global_batch"]
    if images == 0:
        sys.stderr.write("bench.py: rank %d has no images (global batch %d over %d ranks)\n" % (rank, global_batch, world))
        sys.exit(2)
    through = args.through if args.through != "auto" else "host"
    rng = np.random.default_rng(7767517 + rank)
    tmpdir = tempfile.mkdtemp(prefix="snn_bench_")
    if through == "host":
        os.environ.setdefault("SNN_LOG_LEVEL", "2")
        wl = HostWorkload(config, net, shard["micro_sizes"], dev, tmpdir, unfused=args.unfused, capture=not args.no_capture)
    else:
        wl = Workload(ctx, config, net, shard["micro_sizes"], unfused=args.unfused)
    # synthetic input (U(0,1), seed echoing the reference's SRAND(7767517), a different stream per rank), uploaded once: resident in HBM
    wl.upload(rng)
    ctx.sync()
    env["torch"].cuda.synchronize()

    # ---- parity leg (rank 0): the GPU result of image 0 against the oracle's, before anything is timed
    parity, cpu_rec = None, None
    if rank == 0 and not args.no_parity:
        cpu_rec, want = cpu_baseline(config, net, timed_legs=(world == 1 and not args.no_cpu_baseline), compact=not primary)
        got = wl.first_input_output()
        got = np.asarray(got).reshape((-1,) + tuple(want.shape[1:]))[:1]
        parity = parity_record(config, got, want)
        if not parity["ok"]:
            sys.stderr.write("bench.py: PARITY FAILURE (%s) against the CPU oracle: %s\n" % (config, json.dumps(parity)))
    elif rank == 0 and primary and world == 1 and not args.no_cpu_baseline:
        cpu_rec, _ = cpu_baseline(config, net)

    barrier = group.barrier  # dist.barrier + torch.cuda.synchronize()
    barrier()

    def timed_region(step_fn, n):
        """exactly n steps between two barriers + synchronisations; a rank's interval ends on its own clock right after its own wait (the
        trailing barrier sits OUTSIDE it: an RCCL barrier inside a 2.4 ms region would read as 2-4 % of lost scaling), MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        wl.sync()
        dt = time.perf_counter() - t0
        barrier()
        return group.max_over_ranks(dt)

    # time-based pre-heat, then the driver's warmup steps
    t0 = time.perf_counter()
    pre_steps = 0
    preheat_target = args.preheat_ms if primary else min(args.preheat_ms, 100.0)
    while True:
        for _ in range(4):
            wl.run_inflight()
        pre_steps += 4
        wl.sync()
        if 1e3 * (time.perf_counter() - t0) >= preheat_target:
            break
    preheat_ms = 1e3 * (time.perf_counter() - t0)
    value_fn = wl.run_sync if (args.value_mode == "sync" and through == "host") else wl.run_inflight
    for _ in range(warmup):
        value_fn()
    # R timed regions of exactly `steps` steps each; the line quotes the median (a 20-step ESPCN region is 2.5 ms: one sample cannot be told from noise)
    first = timed_region(value_fn, steps)
    repeats = args.repeats if args.repeats > 0 else (7 if first < 0.05 else 3)
    if not primary:
        repeats = min(repeats, 5)
    samples = [first] + [timed_region(value_fn, steps) for _ in range(repeats - 1)]
    elapsed = float(np.median(samples))
    # each rank's OWN clock over one more region of the same steps (the samples above are already the MAX over ranks): a slow GPU shows up by rank
    own_ms = None
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            value_fn()
        wl.sync()
        own_ms = group.gather_values(1e3 * (time.perf_counter() - t0) / steps)
        barrier()

    # the other wait semantics on the same models, same step count (host path; outside `value`)
    modes = {}
    if through == "host" and primary:
        other = wl.run_inflight if value_fn == wl.run_sync else wl.run_sync
        med = lambda fn: float(np.median([timed_region(fn, steps) for _ in range(min(repeats, 3))]))
        if value_fn == wl.run_sync:
            modes["sync_per_inference_blocking_wait"] = elapsed
            modes["inflight"] = med(other)
        else:
            modes["inflight"] = elapsed
            modes["sync_per_inference_blocking_wait"] = med(other)
        capi_mod.set_option("SNNHIP_SYNC_SPIN_US", "2000")  # opt in to the polling wait (the library's default blocks at once)
        modes["sync_per_inference"] = med(wl.run_sync)
        capi_mod.set_option("SNNHIP_SYNC_SPIN_US", None)

    # per-kernel durations: a launch trace over a fixed number of steps run launch by launch on the same plans and buffers, right after the timed
    # region (clocks still up); every kernel is stamped with its own dispatch start / end
    rows, trace_steps = [], 0
    if not args.no_kernel_events and args.event_launches > 0:
        trace_steps = max(1, min(args.event_launches, {"c1": 64, "c2": 32, "c3": 8, "c4": 4, "c5": 2}[config]))
        wl.launch_by_launch(True)
        wl.run_sync() if through == "host" else (wl.run_device(), wl.sync())  # first launch-by-launch pass untraced
        capi_mod.trace_begin()
        for _ in range(trace_steps):
            wl.run_sync() if through == "host" else wl.run_device()
        wl.sync()
        trace = capi_mod.trace_end()
        wl.launch_by_launch(False)
        peak_tf = PEAK_F16_MFMA_TFLOPS if cfg["dtype"] == "f16" else PEAK_F32_MFMA_TFLOPS
        rows = kernel_table(trace, trace_steps, peak_tf)

    layer_table = None
    table_loops = args.layer_table if args.layer_table is not None else (20 if world == 1 else 0)
    if rank == 0 and through == "host" and primary and table_loops > 0:
        layer_table = wl.layer_table(table_loops)

    out = None
    if rank == 0:
        flops, bytes_unfused = wl.cost()
        flops_img, bytes_img = flops / images, bytes_unfused / images
        step_s = elapsed / steps
        value = global_batch * steps / elapsed
        peak_tf = PEAK_F16_MFMA_TFLOPS if cfg["dtype"] == "f16" else PEAK_F32_MFMA_TFLOPS
        H, W = cfg["hw"]
        value_mode = ("sync per inference" if value_fn == wl.run_sync else "inferences in flight") if through == "host" else "kernels enqueued from Python, one wait at the end"
        launches = wl.launches()
        out = {
            "metric": "images/sec (1080p ESPCN 2x) at 1/2/4/8 MI355X; achieved HBM GB/s" if config == "c2" else "images/sec (%s)" % cfg["workload"],
            "value": value, "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic", "preheat_ms": preheat_ms, "preheat_steps": pre_steps,
            "value_mode": value_mode,
            "timing": {"repeats": len(samples), "what": "each repeat = exactly %d steps between two barriers + synchronisations, MAX over ranks; value and ms_per_step are the median" % steps,
                       "median_ms_per_step": 1e3 * step_s, "min_ms_per_step": 1e3 * min(samples) / steps, "max_ms_per_step": 1e3 * max(samples) / steps,
                       "ms_per_step_of_each_repeat": [1e3 * t / steps for t in samples]},
            "config": {"workload": cfg["workload"], "config_id": config, "global_batch": global_batch, "images_per_rank_per_step": images,
                       "micro_batches_per_rank": wl.micro_sizes, "input": [images, H, W, cfg["cin"]],
                       "parallelism": "dp%d (batch split, weights replicated, no data-path collective)" % world,
                       "backend": group.backend or "none (one rank)", "launches_per_step": launches,
                       "rccl_ranks": (group.collective_ranks if group.backend == "nccl" else None),
                       "collective_ranks_seen": group.collective_ranks,
                       "ranks": [{"rank": c["rank"], "device": c["device"].get("ordinal"), "pci": c["device"].get("pci_bus_id")} for c in group.census],
                       "ranks_per_device": max(1, -(-world // max(1, env["torch"].cuda.device_count()))),
                       "path": ("C++ host mirror (libsnn_core.so, JSON + .bin model -> ModelParser -> MixedInferenceCore::create / run%s)" % ("" if args.no_capture else ", recorded hipGraph replay")
                                if through == "host" else "per-layer plans through the C-ABI") +
                               (", graph fusion (snnhip_graph_fuse)" if not args.unfused else ", one kernel per layer") + ", %d kernel launches per step" % launches,
                       "device": info["name"], "compute_units": info["compute_units"]},
            "ms_per_step_of_each_rank": own_ms,
            "parity": parity,
            "max_abs_err": parity["max_abs_err"] if parity else None, "max_rel_err": parity["max_rel_err"] if parity else None,
            "flops_per_image": flops_img, "bytes_per_image_unfused_accounting": bytes_img,
            "achieved_tflops_per_gpu": flops / step_s / 1e12,
            "achieved_hbm_gbps_unfused_accounting_per_gpu": bytes_unfused / step_s / 1e9,
            "frac_hbm_roofline_unfused_accounting": bytes_unfused / step_s / 1e9 / PEAK_HBM_GBPS,
            "frac_compute_roofline": flops / step_s / 1e12 / peak_tf,
        }
        # per-LAYER accounting (every layer reads its input and writes its output once): what the unfused graph would have to move.  Where fusion
        # removes tensors the step beats this figure (c5: 1.5x), so it is reported as a ratio, never as a roofline fraction
        unfused_ms = 1e3 * max(flops / peak_tf / 1e12, bytes_unfused / PEAK_HBM_GBPS / 1e9)
        out["unfused_accounting"] = {"roofline_ms": unfused_ms, "ratio_to_measured": unfused_ms / out["ms_per_step"],
                                     "note": "per-layer byte count of the unfused graph; not a bound on the fused step (a ratio above 1 = traffic the fusion removed)"}
        if modes:
            out["wait_semantics"] = {
                "value_is": value_mode,
                "inflight": {"images_per_s": global_batch * steps / modes["inflight"], "ms_per_step": 1e3 * modes["inflight"] / steps,
                             "what": "RunParameters::deferSync: %d steps enqueued back to back, one wait at the end" % steps},
                "sync_per_inference": {"images_per_s": global_batch * steps / modes["sync_per_inference"], "ms_per_step": 1e3 * modes["sync_per_inference"] / steps,
                                       "what": "reference semantics: MixedInferenceCore::run waits once per inference (core.cpp:203); snnhip_sync polls the stream "
                                               "for up to SNNHIP_SYNC_SPIN_US=2000 us before it blocks (opt-in)"},
                "sync_per_inference_blocking_wait": {"images_per_s": global_batch * steps / modes["sync_per_inference_blocking_wait"],
                                                     "ms_per_step": 1e3 * modes["sync_per_inference_blocking_wait"] / steps,
                                                     "what": "the same with the library default SNNHIP_SYNC_SPIN_US=0: hipStreamSynchronize at once"},
            }
        if layer_table is not None:
            out["layer_table"] = {"method": "reference benchmark table (inferenceProcessor.cpp:84-86,143-199): %d inferences, first 5 dropped, per-stage device timers "
                                            "(MixedInferenceCore::writeTimeStat), mean and population sigma in ms; launch by launch (timers need the host between stages)" % (table_loops + 5),
                                  "rows": layer_table}
        # always on the record (null without a launch trace, --no-kernel-events); since round 5 the whole-step fraction is the step against the SUM of
        # its launches' own rooflines (fused accounting), not rounds 1-4's max(flops, unfused bytes) bound
        out.update({"sum_of_launch_rooflines_ms": None, "frac_of_sum_of_launch_rooflines": None, "whole_step_roofline_ms": None, "frac_of_whole_step_roofline": None,
                    "frac_of_whole_step_roofline_definition": "v2 (round 5+): sum over the step's launches of max(flops / MFMA peak, fused-launch bytes / 8 TB/s) / ms_per_step"})
        if rows:
            # the same bound kernel by kernel, on the FUSED graph's own accounting (a fused launch counts its inputs and outputs once): fusion cannot
            # beat this one, and a compute-bound layer is not hidden behind the graph's HBM total
            per_step = sum(max(r["algorithmic_flops_per_step"] / peak_tf / 1e12, r["algorithmic_bytes_per_step"] / PEAK_HBM_GBPS / 1e9) for r in rows)
            out["sum_of_launch_rooflines_ms"] = 1e3 * per_step
            out["frac_of_sum_of_launch_rooflines"] = 1e3 * per_step / out["ms_per_step"]
            # THE whole-step fraction: the step against the sum of its launches' own rooflines (fused accounting) -- always <= 1 for a correct count
            out["whole_step_roofline_ms"] = 1e3 * per_step
            out["frac_of_whole_step_roofline"] = out["frac_of_sum_of_launch_rooflines"]
            if modes:
                out["frac_of_whole_step_roofline_sync_per_inference"] = 1e3 * per_step / (1e3 * modes["sync_per_inference"] / steps)
            out["sum_of_kernel_durations_ms"] = sum(r["us_per_step"] for r in rows) / 1e3
            out["traced_steps"] = trace_steps
            out["kernels"] = rows if args.all_kernels else [dict(r, instances=r["instances"][:3]) for r in rows[:10]]
            out["roofline"] = roofline_record(rows, peak_tf, single_launch_step=(launches == 1))
            out["roofline"]["whole_step_frac"] = out["frac_of_whole_step_roofline"]
            if modes:
                # `value` keeps the timed region's inferences in flight (RunParameters::deferSync); under the reference's one wait per inference
                # (core.cpp:203) the same models give:
                out["roofline"]["sync_per_inference_images_per_s"] = global_batch * steps / modes["sync_per_inference"]
                out["roofline"]["sync_per_inference_blocking_wait_images_per_s"] = global_batch * steps / modes["sync_per_inference_blocking_wait"]
                out["roofline"]["whole_step_frac_sync_per_inference"] = out["frac_of_whole_step_roofline_sync_per_inference"]
        if cpu_rec is not None:
            out["cpu_baseline"] = cpu_rec
        if not primary:
            # the compact form of the `configs` block
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "scaling", "value_mode", "frac_of_whole_step_roofline", "whole_step_roofline_ms", "unfused_accounting",
                    "sum_of_launch_rooflines_ms", "frac_of_sum_of_launch_rooflines", "sum_of_kernel_durations_ms", "flops_per_image", "bytes_per_image_unfused_accounting")
            comp = {k: out[k] for k in keep if k in out}
            comp["workload"] = cfg["workload"]
            comp["images_per_step"] = global_batch
            comp["micro_batches"] = wl.micro_sizes
            comp["launches_per_step"] = launches
            comp["timing"] = {k: out["timing"][k] for k in ("repeats", "median_ms_per_step", "min_ms_per_step", "max_ms_per_step")}
            comp["parity"] = {k: parity[k] for k in ("ok", "max_abs_err", "max_rel_err", "tolerance") if k in parity} if parity else None
            if parity and "q999_err_over_range" in parity:
                comp["parity"]["q999_err_over_range"] = parity["q999_err_over_range"]
            if rows:
                r = out["roofline"]
                comp["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic_bytes", "kernel", "share_of_gpu_time",
                                                      "launches_per_step", "avg_launch_us", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "frac_executed",
                                                      "mfma_pipe_util_pmc", "whole_step_frac", "arithmetic", "frac_of_split_peak", "split_peak", "arithmetic_note") if k in r}
                comp["kernels"] = [{"function": k["function"], "share_of_gpu_time": k["share_of_gpu_time"], "us_per_step": k["us_per_step"], "launches_per_step": k["launches_per_step"],
                                    "bound": k["bound"], "frac": k["frac"], "frac_executed": k.get("frac_executed"),
                                    **({"arithmetic": k["arithmetic"], "frac_of_split_peak": k["frac_of_split_peak"]} if k.get("arithmetic") else {})} for k in rows[:6]]
            if cpu_rec is not None:
                comp["cpu_baseline"] = cpu_rec
            out = comp
    wl.close()
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--also", default=None,
                    help="comma-separated configs measured in the same run and reported compactly under `configs` (default: the other four when --config c2 "
                         "runs on one GPU; `none` = off)")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions per config (value = median); 0 = 7 when a region is shorter than 50 ms, else 3")
    ap.add_argument("--preheat-ms", type=float, default=150.0, help="run the workload untimed for at least this long before --warmup (clock ramp)")
    ap.add_argument("--unfused", action="store_true", help="one kernel per layer, no chain fusion (debug / comparison)")
    ap.add_argument("--through", choices=["auto", "host", "capi"], default="auto",
                    help="host (default): the C++ host mirror (libsnn_core.so: JSON/.bin model -> MixedInferenceCore::create / run); "
                         "capi: per-layer plans driven from Python through the C-ABI")
    ap.add_argument("--value-mode", choices=["inflight", "sync"], default="inflight",
                    help="what `value` is quoted on (host path): inflight = the K inferences of the timed region enqueued back to back, one wait at the end; "
                         "sync = MixedInferenceCore::run's one wait per inference (the other mode is reported beside it)")
    ap.add_argument("--all-kernels", action="store_true", help="list every kernel function and instantiation of the step in `kernels` (default: the 10 most expensive)")
    ap.add_argument("--micro", type=int, default=0, help="micro-batch size of the rank's share (default: the config's)")
    ap.add_argument("--no-capture", action="store_true", help="host path: launch kernel by kernel instead of replaying the recorded hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the timed CPU legs (the parity check still runs the oracle once)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle check (debug only: the line then says parity: null)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the launch-trace leg (no `roofline` / `kernels`)")
    ap.add_argument("--event-launches", type=int, default=64, help="upper bound on the steps run under the launch trace after the timed region (per config: 64 / 32 / 8 / 4 / 2)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the barrier / MAX-reduction (nccl = RCCL, the driver's; gloo: several ranks may share one GPU -- rank r "
                         "runs on device r %% device_count -- which is how the N > 1 path is exercised on a one-GPU box)")
    ap.add_argument("--layer-table", type=int, default=None, help="loops of the reference-style per-layer table (first 5 dropped); default 20 at N=1, 0 = off")
    ap.add_argument("--detail-out", default=None, help="where the full record goes (default: bench_detail.json beside bench.py, plus gpurun_out/ when present)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started from a plain shell: become N ranks (one process per GPU) under torch.distributed.run
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import torch

    from shadernn_amd import dist as sdist

    rank, local_rank, world = sdist.env_rank_world()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the HIP path has no CPU fallback\n")
        sys.exit(3)
    if torch.cuda.device_count() < world and rank == 0:
        sys.stderr.write("bench.py: %d ranks on %d visible GPU(s)%s\n" % (world, torch.cuda.device_count(),
                                                                          "" if args.backend == "gloo" else " -- RCCL refuses two ranks on one device: use --backend gloo"))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    try:
        # nccl == RCCL on ROCm; only barrier + MAX-reduction of the time use it.  Group() publishes every rank's GPU (ordinal, PCI address) through the
        # rendezvous store and refuses two RCCL ranks on one device before the first collective
        group = sdist.Group(backend=args.backend, local_device=local_rank % max(1, torch.cuda.device_count()))
    except sdist.SharedDeviceError as e:
        sys.stderr.write("bench.py: REFUSED: %s\n" % (e,))
        sys.exit(5)

    import shadernn_amd as snn

    snn.load_library()
    dev = torch.cuda.current_device()
    stream = torch.cuda.Stream(device=dev)
    ctx = snn.Context(dev, stream=stream.cuda_stream)
    env = {"rank": rank, "world": world, "group": group, "ctx": ctx, "dev": dev, "info": ctx.info(), "torch": torch}

    out, parity = run_config(args.config, args, env, primary=True)
    failed = parity is not None and not parity["ok"]
    also = args.also if args.also is not None else ("c1,c3,c4,c5" if (args.config == "c2" and world == 1 and not args.unfused and args.through != "capi") else "none")
    extra = [c for c in also.split(",") if c in CONFIGS and c != args.config] if also != "none" else []
    if extra:
        t0 = time.perf_counter()
        recs = {}
        for c in extra:
            try:
                rec, par = run_config(c, args, env, primary=False)
                failed = failed or (par is not None and not par["ok"])
            except Exception as e:  # one config's failure must not cost the headline line; it is reported, and the run exits non-zero
                rec, failed = {"error": repr(e)}, True
            if rank == 0:
                recs[c] = rec
        if rank == 0:
            out["configs"] = recs
            out["configs_note"] = ("the other BASELINE configs, same process, same method (host mirror + hipGraph replay, in-run oracle parity on image 0, median of R "
                                   "timed regions, launch trace for the per-kernel roofline); CPU legs: the oracle timed on 1-5 whole images per config (the parity pass "
                                   "is one of them), better of all-threads / single-thread, labelled; %.0f s in total" % (time.perf_counter() - t0))
            if "roofline" in out:
                def brief(c, r):
                    if "error" in r:
                        return {"id": c, "error": r["error"]}
                    rf, cb, par = r.get("roofline") or {}, r.get("cpu_baseline") or {}, r.get("parity") or {}
                    return {"id": c, "dtype": r.get("dtype"), "launches": r.get("launches_per_step"), "ms_per_step": r.get("ms_per_step"), "images_per_s": r.get("value"), "whole_step_frac": r.get("frac_of_whole_step_roofline"),
                            "dominant_kernel": rf.get("kernel"), "bound": rf.get("bound"), "frac": rf.get("frac"), "frac_executed": rf.get("frac_executed"),
                            "arithmetic": rf.get("arithmetic"), "frac_of_split_peak": rf.get("frac_of_split_peak"),
                            "share_of_gpu_time": rf.get("share_of_gpu_time"), "parity_ok": par.get("ok"), "max_abs_err": par.get("max_abs_err"),
                            "cpu_images_per_s": cb.get("value"), "cpu_cores": cb.get("cores")}
                out["roofline"]["other_configs"] = [brief(c, recs[c]) for c in extra]
    if rank == 0:
        write_detail(out, args.detail_out)
        print(compact_line(out, os.path.relpath(args.detail_out or os.path.join(ROOT, "bench_detail.json"), ROOT)))
        sys.stdout.flush()
    group.barrier()
    group.close()
    if failed:
        sys.exit(4)


if __name__ == "__main__":
    main()
