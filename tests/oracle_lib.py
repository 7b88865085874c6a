"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Test infrastructure: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DENSE = os.path.join(ORACLE_DIR, "_ref", "ref_dense")

ACT = {"": 0, "linear": 0, "none": 0, "relu": 1, "relu6": 2, "tanh": 3, "sigmoid": 4, "leakyRelu": 5, "SiLU": 6, "SiLU_quirk": 7}
PAD_MODE = {"none": 0, "constant": 1, "replicate": 2, "reflect": 3}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "IC", "OC", "kh", "kw", "sh", "sw", "padT", "padB", "padL", "padR", "padMode", "act")] + [
        ("leaky", C.c_float), ("useBias", C.c_int), ("useBN", C.c_int), ("OH", C.c_int), ("OW", C.c_int)]


_FP = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        l = C.CDLL(LIB)
        l.snn_oracle_out_dim.restype = C.c_int
        l.snn_oracle_out_dim.argtypes = [C.c_int] * 5
        l.snn_oracle_padding_offsets.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int * 4)]
        l.snn_oracle_to_medium_precision.restype = C.c_float
        l.snn_oracle_to_medium_precision.argtypes = [C.c_float]
        l.snn_oracle_random_float.restype = C.c_float
        l.snn_oracle_random_float.argtypes = [C.c_float, C.c_float]
        l.snn_oracle_srand.argtypes = [C.c_uint64]
        l.snn_oracle_dense_act_from_string.restype = C.c_int
        l.snn_oracle_dense_act_from_string.argtypes = [C.c_char_p]
        _lib = l
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_FP)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def padding_offsets(padding, kernel):
    o = (C.c_int * 4)()
    lib().snn_oracle_padding_offsets(padding.encode(), kernel, C.byref(o))
    return tuple(o)


def out_dim(n, k, s, pa, pb):
    return lib().snn_oracle_out_dim(n, k, s, pa, pb)


def make_desc(N, H, W, IC, OC, k, stride, pads, pad_mode="constant", act="", leaky=0.0, use_bias=True, use_bn=False):
    d = ConvDesc()
    d.N, d.H, d.W, d.IC, d.OC = N, H, W, IC, OC
    d.kh = d.kw = k
    d.sh = d.sw = stride
    d.padT, d.padB, d.padL, d.padR = pads
    d.padMode = PAD_MODE[pad_mode] if isinstance(pad_mode, str) else pad_mode
    d.act = ACT[act] if isinstance(act, str) else act
    d.leaky = leaky
    d.useBias, d.useBN = int(use_bias), int(use_bn)
    d.OH = out_dim(H, k, stride, pads[0], pads[1])
    d.OW = out_dim(W, k, stride, pads[0], pads[1])
    return d


def conv2d(x, w_oihw, bias=None, stride=1, pads=None, pad_mode="constant", act="", leaky=0.0, bn=None, threads=1):
    """NHWC walk. x [N,H,W,IC], w [OC,IC,k,k]."""
    x, w = _f(x), _f(w_oihw)
    N, H, W, IC = x.shape
    OC, _, k, _ = w.shape
    if pads is None:
        pads = padding_offsets("same", k)
    d = make_desc(N, H, W, IC, OC, k, stride, pads, pad_mode, act, leaky, bias is not None, bn is not None)
    y = np.empty((N, d.OH, d.OW, OC), dtype=np.float32)
    b = _f(bias)
    bnp = [None] * 4 if bn is None else [_f(bn[q]) for q in ("beta", "gamma", "mean", "var")]
    lib().snn_oracle_conv2d_nhwc_mt(C.byref(d), _p(x), _p(w), _p(b), _p(bnp[0]), _p(bnp[1]), _p(bnp[2]), _p(bnp[3]), _p(y), C.c_int(threads))
    return y


def _pad4(a, n4):
    out = np.zeros(n4, dtype=np.float32)
    if a is not None:
        out[: len(a)] = a
    return out


def conv2d_texel(x, w_oihw, bias=None, stride=1, pads=None, pad_mode="constant", act="", leaky=0.0, bn=None):
    """C4HW4 walk in GLSL loop order, batch 1; returns NHWC for comparison."""
    x, w = _f(x), _f(w_oihw)
    N, H, W, IC = x.shape
    assert N == 1
    OC, _, k, _ = w.shape
    if pads is None:
        pads = padding_offsets("same", k)
    d = make_desc(1, H, W, IC, OC, k, stride, pads, pad_mode, act, leaky, True, bn is not None)
    ic4, oc4 = (IC + 3) // 4 * 4, (OC + 3) // 4 * 4
    xc4 = np.empty((ic4 // 4, H, W, 4), dtype=np.float32)
    lib().snn_oracle_hwc_to_c4hw4(_p(x), H, W, IC, _p(xc4))
    wp = np.empty(oc4 * k * k * ic4, dtype=np.float32)
    lib().snn_oracle_pack_conv_weights(_p(w), IC, OC, k, k, _p(wp))
    b4 = _pad4(_f(bias), oc4)
    bn4 = [_pad4(None if bn is None else _f(bn[q]), oc4) for q in ("beta", "gamma", "mean", "var")]
    yc4 = np.zeros((oc4 // 4, d.OH, d.OW, 4), dtype=np.float32)
    lib().snn_oracle_conv2d_texel(C.byref(d), _p(xc4), _p(wp), _p(b4), _p(bn4[0]), _p(bn4[1]), _p(bn4[2]), _p(bn4[3]), _p(yc4))
    y = np.empty((1, d.OH, d.OW, OC), dtype=np.float32)
    lib().snn_oracle_c4hw4_to_hwc(_p(yc4), d.OH, d.OW, OC, _p(y))
    return y


def depthwise(x, w_chw, bias=None, stride=1, pads=None, act="", leaky=0.0, bn=None):
    x, w = _f(x), _f(w_chw)
    N, H, W, Cc = x.shape
    _, k, _ = w.shape
    if pads is None:
        pads = padding_offsets("same", k)
    d = make_desc(N, H, W, Cc, Cc, k, stride, pads, "constant", act, leaky, True, bn is not None)
    y = np.empty((N, d.OH, d.OW, Cc), dtype=np.float32)
    b = _f(bias)
    bnp = [None] * 4 if bn is None else [_f(bn[q]) for q in ("beta", "gamma", "mean", "var")]
    lib().snn_oracle_depthwise_nhwc(C.byref(d), _p(x), _p(w), _p(b), _p(bnp[0]), _p(bnp[1]), _p(bnp[2]), _p(bnp[3]), _p(y))
    return y


def depthwise_texel(x, w_chw, bias=None, stride=1, pads=None, act="", leaky=0.0, bn=None):
    x, w = _f(x), _f(w_chw)
    N, H, W, Cc = x.shape
    assert N == 1
    _, k, _ = w.shape
    if pads is None:
        pads = padding_offsets("same", k)
    d = make_desc(1, H, W, Cc, Cc, k, stride, pads, "constant", act, leaky, True, bn is not None)
    c4 = (Cc + 3) // 4 * 4
    xc4 = np.empty((c4 // 4, H, W, 4), dtype=np.float32)
    lib().snn_oracle_hwc_to_c4hw4(_p(x), H, W, Cc, _p(xc4))
    wp = np.empty(c4 * k * k, dtype=np.float32)
    lib().snn_oracle_pack_depthwise_weights(_p(w), Cc, k, k, _p(wp))
    b4 = _pad4(_f(bias), c4)
    bn4 = [_pad4(None if bn is None else _f(bn[q]), c4) for q in ("beta", "gamma", "mean", "var")]
    yc4 = np.zeros((c4 // 4, d.OH, d.OW, 4), dtype=np.float32)
    lib().snn_oracle_depthwise_texel(C.byref(d), _p(xc4), _p(wp), _p(b4), _p(bn4[0]), _p(bn4[1]), _p(bn4[2]), _p(bn4[3]), _p(yc4))
    y = np.empty((1, d.OH, d.OW, Cc), dtype=np.float32)
    lib().snn_oracle_c4hw4_to_hwc(_p(yc4), d.OH, d.OW, Cc, _p(y))
    return y


def dense(x, w_flat, out_units, bias=None, act="", leaky=0.0):
    """x [batch, In] (or any [batch, ...] flattened HWC); activation mapped like CPUCommonUtil (unknown -> relu)."""
    x = _f(x)
    batch = x.shape[0]
    x2 = x.reshape(batch, -1)
    In = x2.shape[1]
    w = _f(w_flat).reshape(-1)
    assert w.size == In * out_units
    a = lib().snn_oracle_dense_act_from_string(act.encode()) if isinstance(act, str) else act
    y = np.empty((batch, out_units), dtype=np.float32)
    b = _f(bias)
    lib().snn_oracle_dense(_p(x2), batch, In, out_units, _p(w), _p(b), a, C.c_float(leaky), _p(y))
    return y


def subpixel(x, factor=2, mode=0):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty((N, H * factor, W * factor, 1), dtype=np.float32)
    lib().snn_oracle_subpixel_nhwc(_p(x), N, H, W, Cc, factor, mode, _p(y))
    return y


def add_act(a, b=None, act="", leaky=0.0):
    a = _f(a)
    b = _f(b)
    if b is not None and b.shape != a.shape:  # the reference's max-extent Add (see snn_oracle_add_ragged)
        assert a.ndim == 4 and a.shape[0] == b.shape[0] and a.shape[3] == b.shape[3], (a.shape, b.shape)
        y = np.empty((a.shape[0], max(a.shape[1], b.shape[1]), max(a.shape[2], b.shape[2]), a.shape[3]), np.float32)
        lib().snn_oracle_add_ragged(_p(a), a.shape[1], a.shape[2], _p(b), b.shape[1], b.shape[2], a.shape[0], a.shape[3], ACT[act], C.c_float(leaky), _p(y))
        return y
    y = np.empty_like(a)
    lib().snn_oracle_add_act(_p(a), _p(b), C.c_long(a.size), ACT[act], C.c_float(leaky), _p(y))
    return y


def batchnorm(x, bn, act="", leaky=0.0):
    x = _f(x)
    Cc = x.shape[-1]
    y = np.empty_like(x)
    arrs = [_f(bn[k]) for k in ("beta", "gamma", "mean", "var")]
    lib().snn_oracle_batchnorm(_p(x), C.c_long(x.size // Cc), Cc, *[_p(a) for a in arrs], ACT[act], C.c_float(leaky), _p(y))
    return y


def pool_out_dim(n, k, s, same):
    l = lib()
    l.snn_oracle_pool_out_dim.restype = C.c_int
    return l.snn_oracle_pool_out_dim(n, k, s, int(bool(same)))


def pool2d(x, k, stride, kind="max", same=True, pad_t=0, pad_l=0, OH=0, OW=0):
    x = _f(x)
    N, H, W, Cc = x.shape
    OH = OH or pool_out_dim(H, k, stride, same)
    OW = OW or pool_out_dim(W, k, stride, same)
    y = np.empty((N, OH, OW, Cc), np.float32)
    lib().snn_oracle_pool2d(_p(x), N, H, W, Cc, k, k, stride, stride, pad_t, pad_l, OH, OW, 0 if kind == "max" else 1, _p(y))
    return y


def global_avgpool(x):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty((N, 1, 1, Cc), np.float32)
    lib().snn_oracle_pool2d(_p(x), N, H, W, Cc, H, W, H, W, 0, 0, 1, 1, 1, _p(y))
    return y


def pad(x, pads, mode="constant"):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty((N, H + pads[0] + pads[1], W + pads[2] + pads[3], Cc), np.float32)
    lib().snn_oracle_pad(_p(x), N, H, W, Cc, pads[0], pads[1], pads[2], pads[3], {"constant": 0, "replicate": 1, "reflect": 2}[mode], _p(y))
    return y


def upsample(x, scale=2.0, mode="nearest"):
    x = _f(x)
    N, H, W, Cc = x.shape
    OH, OW = int(np.float32(scale) * np.float32(H)), int(np.float32(scale) * np.float32(W))
    y = np.empty((N, OH, OW, Cc), np.float32)
    lib().snn_oracle_upsample(_p(x), N, H, W, Cc, C.c_float(scale), {"nearest": 0, "bilinear": 1}[mode], _p(y))
    return y


def instancenorm(x, beta, gamma, act="", leaky=0.0, eps=1e-5):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty_like(x)
    lib().snn_oracle_instancenorm(_p(x), N, H, W, Cc, _p(_f(beta)), _p(_f(gamma)), C.c_float(eps), ACT[act], C.c_float(leaky), _p(y))
    return y


def concat(x0, x1, OC=None):
    x0, x1 = _f(x0), _f(x1)
    N, H, W, C0 = x0.shape
    C1 = x1.shape[-1]
    OC = C0 + C1 if OC is None else OC
    y = np.empty((N, H, W, OC), np.float32)
    lib().snn_oracle_concat(_p(x0), _p(x1), C.c_long(N * H * W), C0, C1, OC, _p(y))
    return y


UNARY_OPS = {"copy": 0, "fixed": 1, "neg": 2, "rcp": 3, "square": 4, "exp": 5, "abs": 6}


def unary(x, op="copy", value=1.0):
    x = _f(x)
    y = np.empty_like(x)
    lib().snn_oracle_unary(_p(x), C.c_long(x.size), UNARY_OPS[op] if isinstance(op, str) else int(op), C.c_float(value), _p(y))
    return y


def calculate(x, OC):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty((N, H, W, OC), np.float32)
    lib().snn_oracle_calculate(_p(x), C.c_long(N * H * W), Cc, OC, _p(y))
    return y


def resize(x, OH, OW, means=(0, 0, 0, 0), norms=(1, 1, 1, 1), linear=True):
    x = _f(x)
    N, H, W, Cc = x.shape
    y = np.empty((N, OH, OW, Cc), np.float32)
    lib().snn_oracle_resize(_p(x), N, H, W, Cc, OH, OW, _p(_f(means)), _p(_f(norms)), int(bool(linear)), _p(y))
    return y


def image_u8(img, means=(0, 0, 0, 0), norms=(1, 1, 1, 1)):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    N, H, W, sc = img.shape
    y = np.empty((N, H, W, 4), np.float32)
    lib().snn_oracle_image_u8(img.ctypes.data_as(C.c_void_p), C.c_long(N * H * W), sc, _p(_f(means)), _p(_f(norms)), _p(y))
    return y


def argmax(x):
    x = _f(x).reshape(-1)
    l = lib()
    l.snn_oracle_argmax.restype = C.c_long
    return int(l.snn_oracle_argmax(_p(x), C.c_long(x.size)))


def deconv2d(x, w_oihw, bias=None, stride=2, same=True, act="", leaky=0.0, bn=None):
    x, w = _f(x), _f(w_oihw)
    N, H, W, IC = x.shape
    OC, _, k, _ = w.shape
    p = (k - stride) // 2 if same else 0
    OH, OW = (stride * H, stride * W) if same else (stride * H + k - stride, stride * W + k - stride)
    y = np.empty((N, OH, OW, OC), np.float32)
    arrs = [_f(bn[key]) for key in ("beta", "gamma", "mean", "var")] if bn is not None else [None] * 4
    b = _f(bias) if bias is not None else None
    lib().snn_oracle_deconv2d(_p(x), N, H, W, IC, OC, k, stride, p, OH, OW, _p(w), _p(b), *[_p(a) for a in arrs], ACT[act], C.c_float(leaky), _p(y))
    return y


def deconv4x4s2_shader(x_hwc, w_oihw, bias=None):
    """The k=4 s=2 compute shader of the reference restated line by line (one image)."""
    x, w = _f(x_hwc), _f(w_oihw)
    H, W, IC = x.shape
    OC = w.shape[0]
    y = np.empty((2 * H, 2 * W, OC), np.float32)
    lib().snn_oracle_deconv4x4s2_shader(_p(x), H, W, IC, OC, _p(w), _p(_f(bias)) if bias is not None else None, _p(y))
    return y


# YOLO v3 tiny decode + NMS: numpy restatement of yololayer.cpp:27-226 (the reference runs this layer on the CPU)
YOLO_ANCHORS = [10.0, 14.0, 23.0, 27.0, 37.0, 58.0, 81.0, 82.0, 135.0, 169.0, 344.0, 319.0]
YOLO_MASKS = [3, 4, 5, 1, 2, 3]


def yolo_decode(heads, conf=0.35, iou=0.45, net=416):
    """heads: [grid13 NHWC (1,13,13,18), grid26 (1,26,26,18)] -> list of [classId, score, x, y, w, h] after NMS."""
    boxes = []
    for idx, scale in enumerate((32, 16)):
        g = net // scale
        d = _f(heads[idx]).reshape(g, g, -1)  # the reference walks the 4-channel-aligned texture; NHWC with C=18 holds the same 18 values
        for gy in range(g):
            for gx in range(g):
                for gc in range(3):
                    v = d[gy, gx, gc * 6:(gc + 1) * 6]
                    cls_logit = v[5]
                    a = YOLO_MASKS[gc + idx * 3]
                    bw, bh = YOLO_ANCHORS[2 * a], YOLO_ANCHORS[2 * a + 1]
                    # yololayer.cpp:147 as written: 1 / (1 + exp(-obj) * (1 + exp(-cls)))  (NOT sigmoid(obj) * sigmoid(cls))
                    prob = np.float32(1.0) / (np.float32(1.0) + np.exp(-v[4], dtype=np.float32) * (np.float32(1.0) + np.exp(-cls_logit, dtype=np.float32)))
                    if prob > conf:
                        sig = lambda t: np.float32(1.0) / (np.float32(1.0) + np.exp(-t, dtype=np.float32))
                        cx, cy = (gx + sig(v[0])) / g, (gy + sig(v[1])) / g
                        w_, h_ = np.exp(v[2], dtype=np.float32) * bw / (scale * g), np.exp(v[3], dtype=np.float32) * bh / (scale * g)
                        boxes.append([0, float(prob), float(cx - w_ / 2), float(cy - h_ / 2), float(w_), float(h_)])
    boxes.sort(key=lambda b: -b[1])
    merged, out = [False] * len(boxes), []

    def iou_of(a, b):
        x0, y0 = max(a[2], b[2]), max(a[3], b[3])
        x1, y1 = min(a[2] + a[4], b[2] + b[4]), min(a[3] + a[5], b[3] + b[5])
        if x1 < x0 or y1 < y0:
            return 0.0
        inter = (x1 - x0) * (y1 - y0)
        return inter / (a[4] * a[5] + b[4] * b[5] - inter)

    for i, b in enumerate(boxes):
        if merged[i]:
            continue
        for j in range(i + 1, len(boxes)):
            if not merged[j] and boxes[j][0] == b[0] and iou_of(b, boxes[j]) > iou:
                merged[j] = True
        out.append(b)
    return out


def to_medium_precision(v):
    return lib().snn_oracle_to_medium_precision(C.c_float(v))


def reference_rand(seed, count, a=-1.2, b=1.2):
    """The reference tests' generator: SRAND(seed) then RandomFloat(a, b) x count (demo/common/testutil.cpp:41-46)."""
    lib().snn_oracle_srand(seed)
    return np.array([lib().snn_oracle_random_float(a, b) for _ in range(count)], dtype=np.float32)


def ref_dense(In, Out, act, alpha, w_flat, bias, x):
    """Runs the REFERENCE's own Eigen dense path (oracle/_ref/ref_dense, built from /root/reference)."""
    if not os.path.exists(REF_DENSE):
        return None
    txt = "%d %d %s %r\n" % (In, Out, act if act else "-", float(alpha))
    txt += " ".join("%.9g" % v for v in np.asarray(w_flat, dtype=np.float32).reshape(-1)) + "\n"
    txt += " ".join("%.9g" % v for v in np.asarray(bias, dtype=np.float32)) + "\n"
    txt += " ".join("%.9g" % v for v in np.asarray(x, dtype=np.float32).reshape(-1)) + "\n"
    out = subprocess.run([REF_DENSE], input=txt, stdout=subprocess.PIPE, text=True, check=True).stdout
    return np.array([float(t) for t in out.split()], dtype=np.float32)


def _h(a):
    """fp16 storage emulation: round to nearest even to half, back to float32."""
    return None if a is None else np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def quantize_net_fp16(net):
    """A copy of the net whose conv / dense weights are the fp16-representable values the fp16 kernels use (RNE conversion of the
    given fp32 weights; bias / BN / instance-norm parameters stay fp32 -- the epilogue runs in fp32)."""
    import copy

    q = copy.deepcopy(net)
    for l in q["layers"]:
        if l["type"] in ("Conv2D", "DepthwiseConv2D", "Dense", "Conv2DTranspose"):
            l["w"] = _h(l["w"])
    return q


def forward(net, x, threads=1, return_layers=False, return_named=False, fp16=False):
    """Runs a models.py net (chain or graph: layers may name their producers in "inputs") on the oracle.
    fp16=True emulates the SNNHIP_F16 path: weights and every stored activation are rounded to half, arithmetic stays fp32
    (what an RGBA16F texture chain with >= fp16 shader arithmetic computes)."""
    if fp16:
        net = quantize_net_fp16(net)
        x = _h(x)
    outs, named, prev = [], {"input": _f(x)}, "input"
    for l in net["layers"]:
        ins = [named[n] for n in l.get("inputs", [prev])]
        x = ins[0]
        t = l["type"]
        plain = lambda a: "" if a in ("linear", "none", None) else a
        if t == "Conv2D":
            pads = padding_offsets(l["padding"], l["kernel"])
            x = conv2d(x, l["w"], l["b"], l["stride"], pads, l.get("pad_mode", "constant"), l["activation"], l.get("alpha", 0.0), l["bn"], threads)
        elif t == "DepthwiseConv2D":
            pads = padding_offsets(l["padding"], l["kernel"])
            x = depthwise(x, l["w"], l["b"], l["stride"], pads, l["activation"], l.get("alpha", 0.0), l["bn"])
        elif t == "Dense":
            x = dense(x.reshape(x.shape[0], -1), l["w"], l["units"], l["b"], l["activation"]).reshape(x.shape[0], 1, 1, -1)
        elif t == "Subpixel":
            x = subpixel(x, 2, l.get("mode", 0))
        elif t in ("MaxPooling2D", "AveragePooling2D"):
            x = pool2d(x, l["pool"], l["stride"], "max" if t == "MaxPooling2D" else "avg", l["padding"] not in ("valid", "none", "0"))
        elif t == "AdaptiveAvgPool2d":
            x = global_avgpool(x)
        elif t == "Add":
            x = add_act(ins[0], ins[1], plain(l.get("activation", "")), l.get("alpha", 0.0))
        elif t == "Activation":
            x = add_act(x, None, plain(l.get("activation", "")), l.get("alpha", 0.0))
        elif t == "Flatten":
            x = add_act(x, None, plain(l.get("activation", "")), l.get("alpha", 0.0)).reshape(x.shape[0], 1, 1, -1)
        elif t == "BatchNormalization":
            x = batchnorm(x, l["bn"], plain(l.get("activation", "")), l.get("alpha", 0.0))
        elif t == "Pad":
            (pt, pb), (pl, pr) = l["padding"]
            x = pad(x, (pt, pb, pl, pr), l["mode"])
        elif t == "InstanceNorm":
            x = instancenorm(x, l["beta"], l["gamma"], plain(l.get("activation", "")), l.get("alpha", 0.0))
        elif t == "UpSampling2D":
            x = upsample(x, l["scaleFactor"], l["interpolation"])
        elif t == "Concatenate":
            x = concat(ins[0], ins[1], l.get("oc"))
        elif t == "Unary":
            x = unary(x, l.get("op", "copy"), l.get("value", 1.0))
        elif t == "Calculate":
            x = calculate(x, l["oc"])
        elif t == "Conv2DTranspose":
            x = deconv2d(x, l["w"], l["b"], l["stride"], l["padding"] == "same", plain(l["activation"]), l.get("alpha", 0.0), l["bn"])
        else:
            raise ValueError(t)
        if fp16:
            x = _h(x)
        outs.append(x)
        named[l["name"]] = x
        prev = l["name"]
    if return_named:
        return x, named
    return (x, outs) if return_layers else x


def espcn_forward(net, x, threads=1):
    return forward(net, x, threads)
