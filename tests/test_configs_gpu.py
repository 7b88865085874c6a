"""Every BASELINE.json config at the size it is quoted (and timed) on, through the C-ABI, against the CPU oracle.

  configs[0] (C1)  single 3x3 Conv2D 1x224x224x3 -> 64 fp32          test_c1_*
  configs[1] (C2)  ESPCN 2x 1080p batch 1 fp32                        tests/test_espcn_gpu.py::test_espcn_full_size_properties (+ test_c2_* here)
  configs[2] (C3)  ResNet-18 224x224 fp32 batch 32                    test_c3_*
  configs[3] (C4)  MobileNetV2 224x224 fp32, batch 256 / 8 GPUs = 32  test_c4_*
  configs[4] (C5)  Candy 720p fp16, batch 64 / 8 GPUs                 test_c5_*

Full-size runs take the tile-selection, split-K and grid-size branches the benchmark takes (reduced-size tests do not).  What is
checked: (a) images of the batch against the oracle, layer by layer in the style of the reference's model tests
(resnet18Test.cpp:84-140); (b) the batch-index property: the same image at every batch position gives the same result at every
position; (c) per layer, the MFMA kernels against the direct VALU kernel (SNNHIP_CONV=generic) on the same inputs."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_golden import C1_WINDOWS, c1_inputs, check_g3  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)
THREADS = max(1, os.cpu_count() or 1)


# ------------------------------------------------------------------------------------------------ C1

def test_c1_single_conv_224x224x3_to_64_plan(ctx):
    """BASELINE configs[0] through snnhip_conv2d_plan_create: full tensor vs the oracle, vs the committed G3 vectors (oracle + torch
    values), default kernel choice and both forced families."""
    import shadernn_amd as snn

    x, w, b = c1_inputs()
    want = O.conv2d(x, w, b, 1, (1, 1, 1, 1), "constant", "relu", threads=THREADS)
    seen = set()
    for family in ("", "generic", "mfma"):
        if family:
            os.environ["SNNHIP_CONV"] = family
        try:
            plan = snn.conv2d_plan(ctx, 1, 224, 224, w, b, stride=1, pads=(1, 1, 1, 1), act="relu")
        finally:
            os.environ.pop("SNNHIP_CONV", None)
        seen.add(plan.describe().split(" ")[0])
        got = plan(snn.Tensor.from_numpy(ctx, x)).numpy()
        assert got.shape == (1, 224, 224, 64)
        np.testing.assert_allclose(got, want, err_msg=plan.describe(), **TOL)
        check_g3(got, plan.describe() + " ")
    assert len(seen) >= 2, seen  # the VALU kernel and the MFMA kernel both ran


def test_c1_single_conv_through_conv_test_with_layer(ctx, tmp_path, monkeypatch):
    """BASELINE configs[0] through the host mirror's ShaderUnitTest::snnConvTestWithLayer (convolutionTest.cpp:416-452 call path:
    hand-built InputLayer + Conv2D, C4HW4 upload, layer dump); the test harness has no activation, so relu is applied to the dump."""
    from shadernn_amd import host

    monkeypatch.setenv("SNN_OUTPUT_DIR", str(tmp_path))
    x, w, b = c1_inputs()
    f = host.conv_test_with_layer(x[0], w, b, stride=1, pad=0)
    wd, hd, d, c, px = host.read_dump(f)
    assert (wd, hd, d, c) == (224, 224, 16, 64)
    got = host.c4hw4_to_nhwc(px, 64)[None]
    want = O.conv2d(x, w, b, 1, (1, 1, 1, 1), "constant", "", threads=THREADS)
    np.testing.assert_allclose(got, want, **TOL)
    check_g3(np.maximum(got, 0.0), "conv_test_with_layer ")


def test_c1_batched_and_fp16(ctx):
    """The C1 layer at batch 4 (batch-index property) and with half tensors (vs the quantised oracle)."""
    import shadernn_amd as snn

    x, w, b = c1_inputs()
    plan = snn.conv2d_plan(ctx, 4, 224, 224, w, b, stride=1, pads=(1, 1, 1, 1), act="relu")
    got = plan(snn.Tensor.from_numpy(ctx, np.repeat(x, 4, axis=0))).numpy()
    for i in range(1, 4):
        np.testing.assert_array_equal(got[i], got[0])
    check_g3(got[:1], "batch 4 ")
    p16 = snn.conv2d_plan(ctx, 1, 224, 224, w, b, stride=1, pads=(1, 1, 1, 1), act="relu", dtype=snn.F16)
    y16 = p16(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy()
    want16 = O._h(O.conv2d(O._h(x), O._h(w), b, 1, (1, 1, 1, 1), "constant", "relu", threads=THREADS))
    np.testing.assert_allclose(y16, want16, rtol=4e-3, atol=4e-3, err_msg=p16.describe())


# ------------------------------------------------------------------------------------------------ C2

def test_c2_espcn_1080p_batch_index_and_repeatability(ctx):
    """BASELINE configs[1] at 1080p: two runs of the fused chain are bit-identical (no atomics / order dependence), and a batch of two
    identical frames gives two identical results that equal the batch-1 result."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    H, W = 1080, 1920
    x = np.random.default_rng(3).random((1, H, W, 1), dtype=np.float32)
    r1 = snn.EspcnRunner(ctx, net, 1, H, W, fused=True)
    y_a = r1(x)
    y_b = r1(x)
    np.testing.assert_array_equal(y_a, y_b)
    r2 = snn.EspcnRunner(ctx, net, 2, H, W, fused=True)
    y2 = r2(np.repeat(x, 2, axis=0))
    np.testing.assert_array_equal(y2[0], y2[1])
    np.testing.assert_array_equal(y2[0], y_a[0])


# ------------------------------------------------------------------------------------------------ C3 / C4 shared machinery

def _step_outputs(r):
    """{layer name: ndarray} for every tensor the runner really produces (folded layers have none)."""
    return {layer["name"]: out.numpy() for _, _, out, layer in r.steps}


def _check_classifier_full_size(ctx, net, batch, probe, monkeypatch):
    import shadernn_amd as snn

    H = W = 224
    rng = np.random.default_rng(20260927)
    x = rng.random((batch, H, W, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, batch, H, W)
    y = r(x).reshape(batch, -1)
    got = _step_outputs(r)
    assert np.isfinite(y).all()
    np.testing.assert_allclose(y.sum(axis=1), 1.0, atol=1e-4)  # softmax head

    # (a) layer by layer vs the oracle on the probed batch positions (first, an interior one, last)
    for n in probe:
        _, named = O.forward(net, x[n : n + 1], threads=THREADS, return_named=True)
        checked = 0
        for name, arr in got.items():
            want = named[name]
            np.testing.assert_allclose(arr[n : n + 1].reshape(want.shape), want, err_msg="image %d layer %s" % (n, name), **TOL)
            checked += 1
        assert checked == len(r.steps)

    # (c) the direct VALU convolution kernel on the same inputs, layer by layer (independent kernel family, same C-ABI)
    monkeypatch.setenv("SNNHIP_CONV", "generic")
    monkeypatch.setenv("SNNHIP_CONV_1X1", "0")
    rg = snn.GraphRunner(ctx, net, batch, H, W)
    monkeypatch.delenv("SNNHIP_CONV")
    monkeypatch.delenv("SNNHIP_CONV_1X1")
    assert any("conv2d_generic" in d for d in rg.describe()) and not any("conv2d_mfma" in d or "conv1x1_stream" in d for d in rg.describe())
    rg(x)
    gen = _step_outputs(rg)
    for name, arr in got.items():
        if name in gen:
            np.testing.assert_allclose(arr, gen[name], err_msg="mfma vs generic, layer " + name, **TOL)
    del rg, gen

    # (b) batch-index property: image `probe[1]` replicated at every position -> every position returns that image's result
    n = probe[1]
    yb = r(np.repeat(x[n : n + 1], batch, axis=0)).reshape(batch, -1)
    for i in range(batch):
        np.testing.assert_allclose(yb[i], y[n], rtol=1e-6, atol=1e-7, err_msg="batch position %d" % i)
    return r


def test_c3_resnet18_224_batch32_full_size(ctx, monkeypatch):
    """BASELINE configs[2]: ResNet-18 224x224 fp32 batch 32 -- the shape tools/bench_models.py / bench.py --config c3 time."""
    from shadernn_amd import models

    net = models.resnet18(seed=1)
    r = _check_classifier_full_size(ctx, net, 32, (0, 13, 31), monkeypatch)
    kinds = " ".join(r.describe())
    import re

    assert "conv2d_mfma" in kinds and re.search(r"splitK=([2-9]|1[0-9])", kinds), "expected the MFMA implicit-GEMM kernels incl. a split-K layer at this size"


def test_c4_mobilenetv2_224_batch32_full_size(ctx, monkeypatch):
    """BASELINE configs[3]: MobileNetV2 224x224 fp32, per-GPU share 32 of batch 256 sharded over 8 GPUs."""
    from shadernn_amd import models

    net = models.mobilenetv2(seed=1)
    r = _check_classifier_full_size(ctx, net, 32, (0, 7, 31), monkeypatch)
    kinds = " ".join(r.describe())
    assert "depthwise" in kinds and " stream:" in kinds  # the strip depthwise kernel and the streaming pointwise MFMA kernel


# ------------------------------------------------------------------------------------------------ C5

def test_c5_candy_720p_fp16_full_size(ctx):
    """BASELINE configs[4]: the zoo's candy-9 graph at 720p with half tensors (fp16 MFMA convolutions), batch 2 of the per-GPU share
    of 8: image 0 against the quantised oracle (whole 720p frame: instance-norm statistics are global, crops do not work), both
    positions identical for identical inputs, and the fp32 oracle within the reference's own fp16 bound (0.1, styleTransferTest)."""
    import shadernn_amd as snn
    from test_param_import import _zoo

    H, W = 720, 1280
    net = _zoo("candy-9_simplified-opt", input_shape=(H, W, 3))
    x = np.random.default_rng(5).random((1, H, W, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 2, H, W, dtype=snn.F16)
    y = r(np.repeat(x, 2, axis=0))
    want, named = O.forward(net, x, fp16=True, threads=THREADS, return_named=True)
    # the graph's reflect Pads grow the tensor and its "valid" convolutions do not shrink it back under the reference's size rule (Q20)
    assert want.shape == (1, 826, 1386, 3)
    assert y.shape == (2,) + want.shape[1:] and np.isfinite(y).all()
    np.testing.assert_array_equal(y[0], y[1])
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(y[:1] - want) / scale
    # same acceptance as the reduced-size fp16 graph test: the bulk within a few half ulps, the instance-norm-amplified tail small
    assert np.quantile(err, 0.999) < 6e-3 and err.max() < 6e-2, (float(np.quantile(err, 0.999)), float(err.max()))
    # the first convolution + norm (before any amplification) tightly, at full size
    got = _step_outputs(r)
    first_norm = next(l["name"] for l in net["layers"] if l["type"] == "InstanceNorm")
    if first_norm in got:
        e1 = np.abs(got[first_norm][:1] - named[first_norm])
        assert np.quantile(e1, 0.999) < 8e-3, float(np.quantile(e1, 0.999))
    full = O.forward(net, x, threads=THREADS)
    assert np.abs(y[:1] - full).max() / max(1.0, float(np.abs(full).max())) < 0.1


def test_c5_candy_720p_fp32_crop_free_layers(ctx):
    """Candy's convolutions at 720p in fp32, the first three layers (pad -> 9x9 conv -> instance norm) at the north-star tolerance."""
    import shadernn_amd as snn
    from test_param_import import _zoo

    H, W = 720, 1280
    net = _zoo("candy-9_simplified-opt", input_shape=(H, W, 3))
    net = dict(net, layers=net["layers"][:3])
    x = np.random.default_rng(6).random((1, H, W, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 1, H, W)
    y = r(x)
    want = O.forward(net, x, threads=THREADS)
    np.testing.assert_allclose(y, want, **TOL)


# ------------------------------------------------------------------------------------------------ what bench.py times, through the path it times
# bench.py runs every config through the C++ host mirror (JSON + .bin model -> ModelParser -> MixedInferenceCore::create / run, fusion by
# HipBackend::finalizeStages, recorded hipGraph replay) at the batch / micro-batch sizes below.  These tests run exactly that.

def _bench_model(tmp_path, config, batch, **kw):
    import bench
    from shadernn_amd import host, models

    cfg = bench.CONFIGS[config]
    net = bench.make_net(config)
    H, W = cfg["hw"]
    path = models.write_json(net, W, H, str(tmp_path / (config + ".json")), bin_weights=True)
    m = host.Model(path, W, H, cfg["cin"], fuse_chains=True, prefer_half=cfg["dtype"] == "f16", capture_graph=True, batch=batch, **kw)
    return net, m, (H, W, cfg["cin"])


def test_c2_espcn_1080p_whole_frame_through_host_graph_replay(ctx, tmp_path):
    """BASELINE configs[1] as bench.py times it: the whole 1080p frame (every one of the 2160 x 3840 output pixels, no crops) against the
    oracle at the north-star tolerance; first run = record + launch, second / third run = hipGraph replay, deferred-sync runs in flight."""
    import bench

    net, m, (H, W, C) = _bench_model(tmp_path, "c2", 1)
    x = bench.oracle_input("c2")
    want = O.forward(net, x, threads=THREADS)
    assert want.shape == (1, 2 * H, 2 * W, 1)
    y0 = m(x)
    np.testing.assert_allclose(y0[None], want, **TOL)
    kinds = " ".join(d for _, _, d, _, _ in m.plan_steps())
    assert "wino3x3" in kinds and "d2s" in kinds and len(m.plan_steps()) == 2, kinds  # rule A + rule B: the two kernels of the bench line
    m.run()                                  # replay
    np.testing.assert_array_equal(m.output(), y0)
    x2 = np.random.default_rng(11).random((1, H, W, 1), dtype=np.float32)
    m.upload(x2)                             # same device buffer, new contents: the recording stays valid
    for _ in range(3):
        m.run_async()                        # three inferences in flight, one wait
    m.sync()
    np.testing.assert_allclose(m.output()[None], O.forward(net, x2, threads=THREADS), **TOL)
    m.close()


def test_c3_resnet18_batch32_through_host_graph_replay(ctx, tmp_path):
    """BASELINE configs[2] as bench.py times it: the 32-image batch in one pass through the host mirror (JSON + .bin model -> ModelParser ->
    MixedInferenceCore::create / run) + hipGraph.  Images 0 / 13 / 31 against the oracle (softmax output, 1e-4), the replay of the recording
    bit-identical, a second batch through the same recording, the batch-index property.  Reference style: demo/test/unittest/resnet18Test.cpp:84-140
    (the whole network against a second implementation on the same input)."""
    net, m, (H, W, C) = _bench_model(tmp_path, "c3", 32)
    x = np.random.default_rng(7767517).random((32, H, W, C), dtype=np.float32)
    y = m(x).reshape(32, -1).copy()                 # record + launch
    assert y.shape == (32, 1000) and np.isfinite(y).all()
    np.testing.assert_allclose(y.sum(axis=1), 1.0, atol=1e-4)
    for n in (0, 13, 31):
        want = O.forward(net, x[n : n + 1], threads=THREADS).reshape(-1)
        np.testing.assert_allclose(y[n], want, err_msg="image %d" % n, **TOL)
    steps = m.plan_steps()
    kinds = " ".join(d for _, _, d, _, _ in steps)
    assert "wino" in kinds and "kgroups=2" in kinds and kinds.count("ksplit") == 6, kinds   # the kernels of the bench line: Winograd body (two K groups at 14x14), the stage entries' 3x3 and 1x1 stride-2 pairs on the K-split kernel
    assert len(steps) <= 24, len(steps)              # rule E folded the eight Adds, rule J the max pool, rule 0b the Flatten
    m.run()                                          # replay
    np.testing.assert_array_equal(m.output().reshape(32, -1), y)
    x2 = (0.5 * np.random.default_rng(12).random((32, H, W, C), dtype=np.float32)).astype(np.float32)
    m.upload(x2)                                     # same device buffer, new contents: the recording stays valid
    for _ in range(3):
        m.run_async()                                # three inferences in flight, one wait
    m.sync()
    y2 = m.output().reshape(32, -1).copy()
    np.testing.assert_allclose(y2[31], O.forward(net, x2[31:32], threads=THREADS).reshape(-1), err_msg="second batch, image 31", **TOL)
    yb = m(np.repeat(x[13:14], 32, axis=0)).reshape(32, -1)
    for i in range(32):
        np.testing.assert_allclose(yb[i], y[13], rtol=1e-6, atol=1e-7, err_msg="batch position %d" % i)
    m.close()


def test_c4_mobilenetv2_batch256_through_host_graph_replay(ctx, tmp_path):
    """BASELINE configs[3] as bench.py times it on one GPU: all 256 images in ONE pass through the host mirror + hipGraph.  Images 0 / 127 /
    255 against the oracle (final softmax output, 1e-4), then the batch-index property at this batch size."""
    net, m, (H, W, C) = _bench_model(tmp_path, "c4", 256)
    x = np.random.default_rng(7767517).random((256, H, W, C), dtype=np.float32)
    y = m(x).reshape(256, -1)
    assert y.shape == (256, 1000) and np.isfinite(y).all()
    np.testing.assert_allclose(y.sum(axis=1), 1.0, atol=1e-4)
    for n in (0, 127, 255):
        want = O.forward(net, x[n : n + 1], threads=THREADS).reshape(-1)
        np.testing.assert_allclose(y[n], want, err_msg="image %d" % n, **TOL)
    kinds = " ".join(d for _, _, d, _, _ in m.plan_steps())
    assert "irb_" in kinds, kinds            # the fused inverted-residual kernels the bench line lists
    m.run()                                  # replay of the recording
    np.testing.assert_array_equal(m.output().reshape(256, -1), y)
    xb = np.repeat(x[127:128], 256, axis=0)
    yb = m(xb).reshape(256, -1)
    for i in range(256):
        np.testing.assert_allclose(yb[i], y[127], rtol=1e-6, atol=1e-7, err_msg="batch position %d" % i)
    m.close()


def test_c4_split_precision_blocks_against_their_fp32_mfma_form_through_the_whole_network(ctx, tmp_path, monkeypatch):
    """MobileNetV2 at the benched batch, every inverted-residual block on its fused kernel: the default (split-precision pointwise stages: three f16 MFMA products per
    fp32 product) against SNNHIP_IRB_SPLIT=0 (the same kernels on the fp32 MFMA) through all seventeen blocks -- the logits' worth of difference, measured on the
    softmax output and on the last feature map's scale -- and both against the oracle."""
    x = np.random.default_rng(99).random((256, 224, 224, 3), dtype=np.float32)
    outs = {}
    for split in ("1", "0"):
        monkeypatch.setenv("SNNHIP_IRB_SPLIT", split)
        net, m, (H, W, C) = _bench_model(tmp_path, "c4", 256)
        kinds = " ".join(d for _, _, d, _, _ in m.plan_steps())
        assert ("f16x3split" in kinds) == (split == "1") and "irb_image_kernel" in kinds and "irb_band_kernel" in kinds and "irb_wave_kernel" in kinds, kinds
        outs[split] = m(x).reshape(256, -1).copy()
        m.close()
    want = O.forward(net, x[200:201], threads=THREADS).reshape(-1)
    np.testing.assert_allclose(outs["1"][200], want, **TOL)
    np.testing.assert_allclose(outs["0"][200], want, **TOL)
    # probabilities are ~1e-3: compare relative to each row's largest one
    d = np.abs(outs["1"] - outs["0"]).max(axis=1) / outs["0"].max(axis=1)
    assert d.max() < 2e-5, float(d.max())


@pytest.mark.parametrize("MB", [32, 16], ids=["micro32_as_benched", "micro16"])
def test_c5_candy_720p_fp16_microbatch_default_switches_through_host(ctx, tmp_path, MB):
    """BASELINE configs[4] as bench.py times it (micro-batches of 32 since round 6 -- the 64 -> 32 up-convolution's nominal input, the x2-upsampled 64-channel
    tensor, then has 2.3e9 elements: a description-only plan that only runs fused, rule D -- and of 16 as in rounds 3-5): through the host mirror with the DEFAULT switches, so the 128 -> 128 body layers
    take rule F (tile statistics + in-kernel fold, on from 128 MB per tensor) and rule I (normalisation applied in LDS behind the DMA) -- the
    kernels of the bench line, which the batch-2 test above does not reach.  Image 0 and the last image (different inputs) against the half-quantised
    oracle with the acceptance of the batch-2 test; then a second, different batch through the same plans (stale tile records of the previous
    launch must not leak into the fold: the relaxed-atomic hand-off of norm_fold.h)."""
    net, m, (H, W, C) = _bench_model(tmp_path, "c5", MB)
    kinds = " ".join(d for _, _, d, _, _ in m.plan_steps())
    assert "+tile-stats+fold" in kinds and "in LDS behind the DMA" in kinds, kinds
    rng = np.random.default_rng(7767517)
    x = rng.random((MB, H, W, C), dtype=np.float32)

    def check(y, xs, positions):
        for n in positions:
            want = O.forward(net, xs[n : n + 1], fp16=True, threads=THREADS)
            assert y[n : n + 1].shape == want.shape
            scale = max(1.0, float(np.abs(want).max()))
            err = np.abs(y[n : n + 1] - want) / scale
            assert np.isfinite(y[n]).all()
            assert np.quantile(err, 0.999) < 6e-3 and err.max() < 6e-2, (n, float(np.quantile(err, 0.999)), float(err.max()))

    y = m(x)
    assert y.shape[0] == MB
    check(y, x, (0, MB - 1))
    # per layer at THIS micro-batch (the reference checks layer by layer: demo/common/testutil.h:1194-1195, fp16 bound 0.1; styleTransferTest.cpp):
    # the last image's tensor right behind the first InstanceNorm that the fused graph materialises, and the output of the last residual block -- an error
    # in the statistics fold (rule F) or in the normalisation applied behind the DMA (rule I) shows here before fifteen norms have amplified it
    import re

    _, named = O.forward(net, x[MB - 1 : MB], fp16=True, threads=THREADS, return_named=True)
    stages = m.stages()

    def layer_of(st):
        k = re.search(r"layer \[(\d+)\]", st["name"])
        return net["layers"][int(k.group(1)) - 1] if k and int(k.group(1)) >= 1 else None

    first_norm = next(i for i, st in enumerate(stages) if (layer_of(st) or {}).get("type") == "InstanceNorm")
    last_add = max(i for i, st in enumerate(stages) if (layer_of(st) or {}).get("type") == "Add")
    picks, checked = [next((i for i in range(first_norm, len(stages)) if not stages[i]["fused_away"] and layer_of(stages[i])), None), last_add], 0
    for i in picks:
        if i is None or stages[i]["fused_away"]:
            continue
        lay = layer_of(stages[i])
        got = m.stage_output(i)
        if got is None or lay["name"] not in named:
            continue
        want = named[lay["name"]]
        assert got[MB - 1 : MB].shape == want.shape, (stages[i]["name"], got.shape, want.shape)
        e = np.abs(got[MB - 1 : MB] - want) / max(1.0, float(np.abs(want).max()))
        assert np.isfinite(got[MB - 1]).all() and np.quantile(e, 0.999) < 6e-3 and e.max() < 6e-2, (stages[i]["name"], float(np.quantile(e, 0.999)), float(e.max()))
        del got
        checked += 1
    assert checked >= 1, [st["name"] for st in stages if not st["fused_away"]]
    # a different batch right behind it (brighter, other statistics) through the same plans and the same record buffers
    x2 = (0.25 + 0.5 * rng.random((MB, H, W, C), dtype=np.float32)).astype(np.float32)
    y2 = m(x2)
    check(y2, x2, (MB // 2 - 1,))
    m.close()


def test_c3_stage_entries_run_their_two_branches_as_one_launch(ctx, tmp_path, monkeypatch):
    """Round 6, the default: the 3x3 stride-2 convolution of a ResNet stage entry and the 1x1 stride-2 downsample beside it (same input, both on
    conv2d_ksplit) are bracketed by a launch group (snnhip_ctx_group_begin / _end) and leave as ONE kernel launch -- in the recorded hipGraph as well.
    Same bits as the two launches (SNN_STAGE_GROUPS=0), on a batch of 4 at 224x224: first run (record), replay and launch-by-launch."""
    from shadernn_amd import host, models

    net = models.resnet18(seed=1)
    H = W = 224
    path = models.write_json(net, W, H, str(tmp_path / "resnet18.json"), bin_weights=True)
    x = np.random.default_rng(33).random((4, H, W, 3), dtype=np.float32)
    monkeypatch.setenv("SNN_STAGE_GROUPS", "0")
    plain = host.Model(path, W, H, 3, device=0, capture_graph=True, batch=4)
    want = np.asarray(plain(x)).copy()
    assert not any(s.get("group") for s in plain.stages())
    plain.close()
    monkeypatch.delenv("SNN_STAGE_GROUPS")
    m = host.Model(path, W, H, 3, device=0, capture_graph=True, batch=4)
    groups = [s for s in m.stages() if s.get("group")]
    assert len(groups) == 3 and not any(s.get("side") for s in m.stages()), [s["name"] for s in m.stages()]   # l2_b0_down, l3_b0_down, l4_b0_down
    got = np.asarray(m(x)).copy()                   # record + launch
    np.testing.assert_array_equal(got, want)
    m.run()                                          # replay
    np.testing.assert_array_equal(np.asarray(m.output()), want)
    m.suspend_replay(True)
    m.run()                                          # launch by launch
    np.testing.assert_array_equal(np.asarray(m.output()), want)
    m.close()


def test_c3_residual_branches_side_by_side_give_the_same_result(ctx, tmp_path, monkeypatch):
    """SNN_BRANCH_OVERLAP=1 (host mirror, opt-in): the 1x1 stride-2 downsample of a ResNet stage entry is issued on the context's side stream
    (snnhip_ctx_fork / _main / _join) beside the 3x3 stride-2 convolution that reads the same tensor -- in the recorded hipGraph as well.  Same
    bits as the one-stream order, on a batch of 4 at 224x224, first run (record), replay and launch-by-launch."""
    from shadernn_amd import host, models

    net = models.resnet18(seed=1)
    H = W = 224
    path = models.write_json(net, W, H, str(tmp_path / "resnet18.json"), bin_weights=True)
    x = np.random.default_rng(31).random((4, H, W, 3), dtype=np.float32)
    plain = host.Model(path, W, H, 3, device=0, capture_graph=True, batch=4)
    want = np.asarray(plain(x)).copy()
    plain.close()
    monkeypatch.setenv("SNN_BRANCH_OVERLAP", "1")
    monkeypatch.setenv("SNN_STAGE_GROUPS", "0")     # (since round 6 such a pair is ONE launch by default: the side stream only when that is off)
    monkeypatch.setenv("SNN_LOG_LEVEL", "3")
    m = host.Model(path, W, H, 3, device=0, capture_graph=True, batch=4)
    got = np.asarray(m(x)).copy()                   # record + launch
    np.testing.assert_array_equal(got, want)
    m.run()                                          # replay of the two-stream graph
    np.testing.assert_array_equal(np.asarray(m.output()), want)
    m.suspend_replay(True)
    m.run()                                          # launch by launch, events between the streams
    np.testing.assert_array_equal(np.asarray(m.output()), want)
    sides = [s for s in m.stages() if s.get("side")]
    assert len(sides) == 3, [s["name"] for s in m.stages()]   # l2_b0_down, l3_b0_down, l4_b0_down
    m.close()
