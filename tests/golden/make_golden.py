#!/usr/bin/env python
"""Generates the committed golden vectors (run in the build container, where /root/reference is mounted).

  python tests/golden/make_golden.py

Sources of truth:
  * dense_ref.json  -- outputs of the REFERENCE's own Eigen dense path (oracle/_ref/ref_dense, built from
                       /root/reference/core/src/ic2/cpulayer.h by oracle/Makefile); pins the dense oracle.
  * prng_ref.json   -- first values of the reference test generator (demo/common/prng.h, public domain) compiled where it
                       lies; pins oracle's re-implementation used for the convolutionTest KAT (G1).
  * *.npz           -- conv / depthwise / subpixel / ESPCN vectors produced by the oracle (the reference has no CPU conv
                       and its ncnn ground truth is not vendored: "parity unpinned" by reference fixtures), cross-checked
                       here against torch CPU before being written.
Only data is stored: inputs, parameters and expected outputs.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402
from shadernn_amd import models  # noqa: E402

REF = "/root/reference"

C1_WINDOWS = {"interior": (slice(100, 104), slice(100, 104)), "top_left": (slice(0, 2), slice(0, 2)), "bottom_right": (slice(222, 224), slice(222, 224)),
              "right_edge": (slice(60, 62), slice(222, 224))}


def c1_inputs():
    """C1 tensors from the reference tests' generator: SRAND(7767517); x ~ U(-1,1) [224*224*3], w ~ U(-1.2,1.2)/sqrt(27), b ~ U(-0.1,0.1)."""
    v = O.reference_rand(7767517, 224 * 224 * 3 + 64 * 27 + 64, -1.0, 1.0)
    x = v[: 224 * 224 * 3].reshape(1, 224, 224, 3).copy()
    w = (v[224 * 224 * 3 : 224 * 224 * 3 + 64 * 27] * np.float32(1.2 / np.sqrt(27.0))).astype(np.float32).reshape(64, 3, 3, 3)
    b = (v[-64:] * np.float32(0.1)).astype(np.float32)
    return x, w, b


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def bn(c, seed):
    r = np.random.default_rng(seed)
    return {"beta": r.uniform(-0.1, 0.1, c).astype(np.float32), "gamma": r.uniform(0.5, 1.5, c).astype(np.float32),
            "mean": r.uniform(-0.1, 0.1, c).astype(np.float32), "var": r.uniform(0.5, 1.5, c).astype(np.float32)}


def torch_conv(x, w, b, stride, pads, pad_mode, act, leaky, bnp, out_hw):
    """The same layer in plain torch CPU ops (F.pad + F.conv2d + the reference's BN / activation formulas): an implementation that
    shares no code with oracle/snn_oracle.c.  Stored beside the oracle's output so the GPU tests compare against BOTH."""
    import torch
    import torch.nn.functional as F

    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    pt, pb, pl, pr = pads
    extra = 2 * max(w.shape[2], w.shape[3])  # room for the float output-size rule; cropped below
    if pad_mode == "constant":
        xt = F.pad(xt, (pl, pr + extra, pt, pb + extra))
    else:
        xt = F.pad(xt, (pl, pr, pt, pb), mode=pad_mode)
        xt = F.pad(xt, (0, extra, 0, extra))
    y = F.conv2d(xt, torch.from_numpy(w), None if b is None else torch.from_numpy(b), stride=stride)
    if bnp is not None:  # vk_conv2d.comp:277-288: (v - mean) / max(sqrt(var + 1e-3), 1e-4) * gamma + beta
        g = lambda k: torch.from_numpy(bnp[k]).view(1, -1, 1, 1)
        y = (y - g("mean")) / torch.clamp(torch.sqrt(g("var") + 1e-3), min=1e-4) * g("gamma") + g("beta")
    y = {"": lambda v: v, "relu": torch.relu, "relu6": lambda v: torch.clamp(v, 0, 6), "tanh": torch.tanh, "sigmoid": torch.sigmoid,
         "leakyRelu": lambda v: torch.where(v > 0, v, v * leaky)}[act](y)
    return y.permute(0, 2, 3, 1).numpy()[:, : out_hw[0], : out_hw[1]].copy()


def main():
    # ---- prng_ref.json: the reference's generator, compiled from where it lies
    src = '#include <stdio.h>\n#include "%s/demo/common/prng.h"\nint main(){struct prng_rand_t s; prng_srand(7767517,&s); for(int i=0;i<16;i++) printf("%%llu\\n",(unsigned long long)prng_rand(&s)); return 0;}\n' % REF
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-O1", "-o", os.path.join(td, "p"), os.path.join(td, "p.c")])
        vals = [int(v) for v in subprocess.check_output([os.path.join(td, "p")]).split()]
    floats = O.reference_rand(7767517, 16)
    mine = [float(np.float32(-1.2) + np.float32(np.float32(v) / np.float32(2 ** 64)) * np.float32(2.4)) for v in vals]
    assert np.allclose(floats, mine, rtol=0, atol=1e-6), (floats, mine)
    json.dump({"seed": 7767517, "raw_u64": [str(v) for v in vals], "random_float_m1p2_p1p2": [float(v) for v in floats]},
              open(os.path.join(HERE, "prng_ref.json"), "w"), indent=1)

    # ---- dense_ref.json: reference Eigen path
    cases = []
    rng = np.random.default_rng(41)
    for In, Out in [(11, 5), (37, 3), (64, 10)]:
        w = rng.standard_normal(In * Out).astype(np.float32)
        b = rng.standard_normal(Out).astype(np.float32)
        x = rng.standard_normal(In).astype(np.float32)
        for act in ["relu", "", "sigmoid", "tanh", "softmax", "leakyRelu", "SiLU", "linear"]:
            ref = O.ref_dense(In, Out, act, 0.3, w, b, x)
            assert ref is not None, "oracle/_ref/ref_dense missing"
            cases.append({"in": In, "out": Out, "act": act, "alpha": 0.3, "w_flat": w.tolist(), "bias": b.tolist(), "x": x.tolist(),
                          "expected": ref.tolist()})
    json.dump({"source": "oracle/_ref/ref_dense == reference CPUCommonUtil<float> (Eigen), core/src/ic2/cpulayer.h:136-266", "cases": cases},
              open(os.path.join(HERE, "dense_ref.json"), "w"))

    # ---- G1: convolutionTest.cpp defaults
    w1 = O.reference_rand(7767517, 128).reshape(1, 128, 1, 1)
    x1 = np.ones((1, 8, 8, 128), np.float32)
    bn1 = {"beta": np.zeros(1, np.float32), "gamma": np.ones(1, np.float32), "mean": np.zeros(1, np.float32), "var": np.ones(1, np.float32)}
    y1 = O.conv2d(x1, w1, np.zeros(1, np.float32), 1, (0, 0, 0, 0), "constant", "", 0.0, bn1)
    y1t = torch_conv(x1, w1, None, 1, (0, 0, 0, 0), "constant", "", 0.0, bn1, y1.shape[1:3])
    assert np.allclose(y1, y1t, rtol=2e-5, atol=2e-5)
    np.savez_compressed(os.path.join(HERE, "g1_convtest_8x8x128_k1.npz"), x=x1, w=w1, y=y1, y_torch=y1t)

    # ---- G2: 3x3 conv family
    import torch
    import torch.nn.functional as F
    g2 = {}
    idx = 0
    for ic, oc in [(3, 1), (4, 4), (5, 5), (8, 12)]:
        for stride in (1, 2):
            for pm in ("constant", "replicate", "reflect"):
                for act in ("", "relu", "leakyRelu", "tanh"):
                    if (idx % 3) != 0:  # keep the file small: every third combination
                        idx += 1
                        continue
                    x = rnd((1, 8, 8, ic), 100 + idx)
                    w = rnd((oc, ic, 3, 3), 200 + idx, 1 / np.sqrt(9 * ic))
                    b = rnd((oc,), 300 + idx, 0.1)
                    bnp = bn(oc, 400 + idx)
                    y = O.conv2d(x, w, b, stride, (1, 1, 1, 1), pm, act, 0.1, bnp)
                    yt = O.conv2d_texel(x, w, b, stride, (1, 1, 1, 1), pm, act, 0.1, bnp)
                    assert np.allclose(y, yt, rtol=2e-5, atol=2e-6)
                    ytorch = torch_conv(x, w, b, stride, (1, 1, 1, 1), pm, act, 0.1, bnp, y.shape[1:3])
                    assert ytorch.shape == y.shape and np.allclose(y, ytorch, rtol=2e-5, atol=2e-5), (idx, np.abs(y - ytorch).max())
                    k = "c%d" % idx
                    g2[k + "_x"], g2[k + "_w"], g2[k + "_b"], g2[k + "_y"] = x, w, b, y
                    g2[k + "_yt"] = ytorch
                    for q in ("beta", "gamma", "mean", "var"):
                        g2[k + "_bn_" + q] = bnp[q]
                    g2[k + "_meta"] = np.array([ic, oc, stride, {"constant": 1, "replicate": 2, "reflect": 3}[pm], O.ACT[act]])
                    idx += 1
    np.savez_compressed(os.path.join(HERE, "g2_conv3x3_family.npz"), **g2)

    # ---- G4: depthwise (depthwiseConv2DTest.cpp:316 + 3x3 cases)
    g4 = {}
    for i, (c, k, s) in enumerate([(8, 1, 2), (8, 3, 1), (32, 3, 2), (96, 3, 1)]):
        x = np.ones((1, 9, 9, c), np.float32) if i == 0 else rnd((1, 9, 9, c), 500 + i)
        w = rnd((c, k, k), 510 + i, 0.3)
        b = rnd((c,), 520 + i, 0.1)
        pads = O.padding_offsets("same", k)
        y = O.depthwise(x, w, b, s, pads, "relu6")
        xt = torch.from_numpy(x).permute(0, 3, 1, 2)
        yt = F.conv2d(F.pad(xt, (pads[0], pads[0] + 2, pads[2], pads[2] + 2)), torch.from_numpy(w[:, None]), torch.from_numpy(b), stride=s, groups=c)
        yt = torch.clamp(yt, 0, 6).permute(0, 2, 3, 1).numpy()[:, : y.shape[1], : y.shape[2]]
        assert np.allclose(y, yt, rtol=2e-5, atol=2e-5)
        g4["d%d_x" % i], g4["d%d_w" % i], g4["d%d_b" % i], g4["d%d_y" % i] = x, w, b, y
        g4["d%d_yt" % i] = np.ascontiguousarray(yt)
        g4["d%d_meta" % i] = np.array([c, k, s])
    np.savez_compressed(os.path.join(HERE, "g4_depthwise.npz"), **g4)

    # ---- G6: ESPCN 32x32 end to end
    net = models.espcn_weights(seed=1)
    x = np.random.default_rng(7767517).random((1, 32, 32, 1), dtype=np.float32)
    y, layers = O.forward(net, x, return_layers=True)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    L = net["layers"]
    t = F.relu(F.conv2d(xt, torch.from_numpy(L[0]["w"]), torch.from_numpy(L[0]["b"]), padding=2))
    t = F.relu(F.conv2d(t, torch.from_numpy(L[1]["w"]), torch.from_numpy(L[1]["b"]), padding=1))
    t = F.conv2d(t, torch.from_numpy(L[2]["w"]), torch.from_numpy(L[2]["b"]), padding=1)
    t = torch.tanh(F.pixel_shuffle(t, 2)).permute(0, 2, 3, 1).numpy()
    assert np.allclose(y, t, rtol=1e-5, atol=1e-5)
    np.savez_compressed(os.path.join(HERE, "g6_espcn_32x32.npz"), x=x, y=y, y_torch=np.ascontiguousarray(t), conv1=layers[0], conv2=layers[1], conv3=layers[2],
                        **{"w%d" % i: L[i]["w"] for i in range(3)}, **{"b%d" % i: L[i]["b"] for i in range(3)})
    # ---- G3: BASELINE configs[0] (C1): 1x224x224x3 -> 64, 3x3 s1 same, relu.  Inputs come from the reference tests' generator
    # (prng.h, pinned by prng_ref.json), so only the seed is stored; expected = checksums + sampled windows, from the
    # oracle AND from torch.
    x3, w3, b3 = c1_inputs()
    y3 = O.conv2d(x3, w3, b3, 1, (1, 1, 1, 1), "constant", "relu", threads=8)
    y3t = torch_conv(x3, w3, b3, 1, (1, 1, 1, 1), "constant", "relu", 0.0, None, (224, 224))
    assert y3.shape == (1, 224, 224, 64) and np.allclose(y3, y3t, rtol=2e-5, atol=2e-5)
    g3 = {"shape": np.array(y3.shape)}
    for tag, yy in (("", y3), ("_torch", y3t)):
        g3["sum" + tag], g3["abssum" + tag] = np.float64(yy.astype(np.float64).sum()), np.float64(np.abs(yy.astype(np.float64)).sum())
        g3["rowsum" + tag] = yy.astype(np.float64).sum(axis=(0, 2, 3))   # 224 per-row sums: localises a wrong tile
        g3["chansum" + tag] = yy.astype(np.float64).sum(axis=(0, 1, 2))  # 64 per-channel sums
        for nm, (ys, xs) in C1_WINDOWS.items():
            g3["win_" + nm + tag] = yy[0, ys, xs, :].copy()
    np.savez_compressed(os.path.join(HERE, "g3_c1_224x224x3_64.npz"), **g3)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
