"""ctypes binding of the C-ABI in include/snnhip.h (libsnnhip.so).

This is host-side plumbing only: every operator call goes through the C entry points, exactly what a cgo/JNI/C++
caller would bind.  The library is mandatory -- there is no CPU or torch fallback; a missing or broken library raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNNHIP_LIB_PATH") or os.path.join(_HERE, "lib", "libsnnhip.so")  # the override serves ablation builds (tools/)

OK, E_INVALID, E_HIP, E_UNSUPPORTED, E_NOMEM, E_GUARD = 0, -1, -2, -3, -4, -5

ACT = {"": 0, "linear": 0, "none": 0, "relu": 1, "relu6": 2, "tanh": 3, "sigmoid": 4, "leakyRelu": 5, "SiLU": 6, "SiLU_quirk": 7}
PAD_MODE = {"none": 0, "constant": 1, "replicate": 2, "reflect": 3}
DENSE_ACT = {"identity": 0, "": 0, "relu": 1, "leakyRelu": 2, "sigmoid": 3, "softmax": 4, "tanh": 5, "SiLU": 6}


class SnnHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("snnhip error %d: %s" % (code, msg))
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "IC", "OC", "kh", "kw", "sh", "sw", "padT", "padB", "padL", "padR", "padMode", "act")] + [
        ("leaky", C.c_float), ("useBias", C.c_int), ("useBN", C.c_int), ("dtype", C.c_int), ("OH", C.c_int), ("OW", C.c_int)]


class DenseDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("in_units", C.c_int), ("out_units", C.c_int), ("act", C.c_int), ("leaky", C.c_float), ("useBias", C.c_int)]


class SubpixelDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "factor", "mode")]


class EltwiseDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("act", C.c_int), ("leaky", C.c_float)]


class Pool2dDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "kh", "kw", "sh", "sw", "padT", "padL", "OH", "OW", "same", "type")]


class PadDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "padT", "padB", "padL", "padR", "mode")]


class UpsampleDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("scale", C.c_float), ("mode", C.c_int)]


class InstanceNormDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("act", C.c_int), ("leaky", C.c_float), ("eps", C.c_float)]


class ConcatDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C0", C.c_int), ("C1", C.c_int), ("OC", C.c_int)]


class UnaryDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("op", C.c_int), ("value", C.c_float)]


class CalculateDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("OC", C.c_int)]


class ResizeDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("means", C.c_float * 4),
                ("norms", C.c_float * 4), ("linear", C.c_int)]


class ImageU8Desc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("src_channels", C.c_int), ("means", C.c_float * 4), ("norms", C.c_float * 4)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("compute_units", C.c_int), ("lds_bytes_per_cu", C.c_int), ("hbm_bytes", C.c_size_t), ("device", C.c_int)]


_lib = None

_P = C.c_void_p
_FP = C.POINTER(C.c_float)


class GraphNode(C.Structure):
    _fields_ = [("plan", _P), ("n_inputs", C.c_int), ("inputs", C.c_int * 4), ("keep", C.c_int)]


class FusedNode(C.Structure):
    _fields_ = [("plan", _P), ("owned", C.c_int), ("n_inputs", C.c_int), ("inputs", C.c_int * 4)]


# name -> (restype, argtypes); must list every symbol include/snnhip.h declares (tests/test_abi.py checks that)
SIGNATURES = {
    "snnhip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "snnhip_ctx_create_on_stream": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "snnhip_ctx_destroy": (C.c_int, [_P]),
    "snnhip_ctx_info": (C.c_int, [_P, C.POINTER(DeviceInfo)]),
    "snnhip_ctx_stream": (_P, [_P]),
    "snnhip_ctx_fork": (C.c_int, [_P]),
    "snnhip_ctx_main": (C.c_int, [_P]),
    "snnhip_ctx_join": (C.c_int, [_P]),
    "snnhip_ctx_group_begin": (C.c_int, [_P]),
    "snnhip_ctx_group_end": (C.c_int, [_P]),
    "snnhip_plan_groupable": (C.c_int, [_P]),
    "snnhip_sync": (C.c_int, [_P]),
    "snnhip_last_error": (C.c_char_p, []),
    "snnhip_version": (C.c_char_p, []),
    "snnhip_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "snnhip_guard_active": (C.c_int, []),
    "snnhip_guard_check": (C.c_int, [_P]),
    "snnhip_guard_selftest": (C.c_int, [_P, _P, C.c_long]),
    "snnhip_get_option": (C.c_char_p, [C.c_char_p]),
    "snnhip_tensor_alloc": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snnhip_tensor_wrap": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snnhip_tensor_free": (C.c_int, [_P]),
    "snnhip_tensor_dims": (C.c_int, [_P, C.POINTER(C.c_int * 4)]),
    "snnhip_tensor_data": (_P, [_P]),
    "snnhip_tensor_bytes": (C.c_size_t, [_P]),
    "snnhip_tensor_dtype": (C.c_int, [_P]),
    "snnhip_tensor_upload": (C.c_int, [_P, _FP]),
    "snnhip_tensor_download": (C.c_int, [_P, _FP]),
    "snnhip_tensor_upload_c4hw4": (C.c_int, [_P, _FP]),
    "snnhip_tensor_download_c4hw4": (C.c_int, [_P, _FP]),
    "snnhip_tensor_fill": (C.c_int, [_P, C.c_float]),
    "snnhip_conv2d_plan_create": (C.c_int, [_P, C.POINTER(ConvDesc), _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(_P)]),
    "snnhip_depthwise_plan_create": (C.c_int, [_P, C.POINTER(ConvDesc), _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(_P)]),
    "snnhip_dense_plan_create": (C.c_int, [_P, C.POINTER(DenseDesc), _FP, _FP, C.POINTER(_P)]),
    "snnhip_subpixel_plan_create": (C.c_int, [_P, C.POINTER(SubpixelDesc), C.POINTER(_P)]),
    "snnhip_add_plan_create": (C.c_int, [_P, C.POINTER(EltwiseDesc), C.POINTER(_P)]),
    "snnhip_activation_plan_create": (C.c_int, [_P, C.POINTER(EltwiseDesc), C.POINTER(_P)]),
    "snnhip_batchnorm_plan_create": (C.c_int, [_P, C.POINTER(EltwiseDesc), _FP, _FP, _FP, _FP, C.POINTER(_P)]),
    "snnhip_pool2d_plan_create": (C.c_int, [_P, C.POINTER(Pool2dDesc), C.POINTER(_P)]),
    "snnhip_pad_plan_create": (C.c_int, [_P, C.POINTER(PadDesc), C.POINTER(_P)]),
    "snnhip_upsample_plan_create": (C.c_int, [_P, C.POINTER(UpsampleDesc), C.POINTER(_P)]),
    "snnhip_instancenorm_plan_create": (C.c_int, [_P, C.POINTER(InstanceNormDesc), _FP, _FP, C.POINTER(_P)]),
    "snnhip_concat_plan_create": (C.c_int, [_P, C.POINTER(ConcatDesc), C.POINTER(_P)]),
    "snnhip_unary_plan_create": (C.c_int, [_P, C.POINTER(UnaryDesc), C.POINTER(_P)]),
    "snnhip_deconv2d_plan_create": (C.c_int, [_P, C.POINTER(ConvDesc), _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(_P)]),
    "snnhip_calculate_plan_create": (C.c_int, [_P, C.POINTER(CalculateDesc), C.POINTER(_P)]),
    "snnhip_resize_plan_create": (C.c_int, [_P, C.POINTER(ResizeDesc), C.POINTER(_P)]),
    "snnhip_image_u8_plan_create": (C.c_int, [_P, C.POINTER(ImageU8Desc), C.POINTER(_P)]),
    "snnhip_tensor_upload_raw": (C.c_int, [_P, _P, C.c_size_t]),
    "snnhip_tensor_argmax": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "snnhip_chain_plan_create": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "snnhip_graph_fuse": (C.c_int, [_P, C.POINTER(GraphNode), C.c_int, C.POINTER(FusedNode)]),
    "snnhip_plan_run": (C.c_int, [_P, _P, _P]),
    "snnhip_plan_run_n": (C.c_int, [_P, C.POINTER(_P), C.c_int, _P]),
    "snnhip_plan_output_dims": (C.c_int, [_P, C.POINTER(C.c_int * 4)]),
    "snnhip_plan_describe": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "snnhip_plan_cost": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "snnhip_plan_destroy": (C.c_int, [_P]),
    "snnhip_plan_num_steps": (C.c_int, [_P]),
    "snnhip_plan_step_describe": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t]),
    "snnhip_plan_step_cost": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "snnhip_plan_profile_enable": (C.c_int, [_P, C.c_int]),
    "snnhip_plan_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "snnhip_graph_begin_capture": (C.c_int, [_P]),
    "snnhip_graph_end_capture": (C.c_int, [_P, C.POINTER(_P)]),
    "snnhip_graph_launch": (C.c_int, [_P]),
    "snnhip_graph_num_nodes": (C.c_int, [_P]),
    "snnhip_graph_destroy": (C.c_int, [_P]),
    "snnhip_timer_create": (C.c_int, [_P, C.POINTER(_P)]),
    "snnhip_timer_start": (C.c_int, [_P]),
    "snnhip_timer_stop": (C.c_int, [_P]),
    "snnhip_timer_elapsed_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "snnhip_timer_destroy": (C.c_int, [_P]),
    "snnhip_trace_begin": (C.c_int, []),
    "snnhip_trace_end": (C.c_int, []),
    "snnhip_trace_report": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
}


def load_library():
    """Loads libsnnhip.so (built in-tree by __graft_entry__.build()).  torch is imported first so that the process has
    ONE HIP runtime: torch's bundled libamdhip64 has SONAME libamdhip64.so.7, which satisfies this library's
    DT_NEEDED; loading in the other order would map a second runtime."""
    global _lib
    if _lib is not None:
        return LIB_PATH
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first" % LIB_PATH)
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return LIB_PATH


def lib():
    load_library()
    return _lib


def set_option(name, value):
    """snnhip_set_option: override a kernel-selection / fusion switch for this process; None removes the override (the environment
    variable of the same name is the fallback)."""
    check(lib().snnhip_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    v = lib().snnhip_get_option(name.encode())
    return None if v is None else v.decode()


def trace_begin():
    """snnhip_trace_begin: from now on every kernel the library launches is stamped with its own dispatch start / end and booked on its plan."""
    check(lib().snnhip_trace_begin())


def trace_end():
    """snnhip_trace_end + snnhip_trace_report: the launches since trace_begin grouped by kernel function (parsed JSON)."""
    import json

    check(lib().snnhip_trace_end())
    need = C.c_size_t(0)
    check(lib().snnhip_trace_report(None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value + 16)
    check(lib().snnhip_trace_report(buf, len(buf), C.byref(need)))
    return json.loads(buf.value.decode("utf-8", "replace"))


def check(rc):
    if rc != OK:
        raise SnnHipError(rc, lib().snnhip_last_error().decode("utf-8", "replace"))


def _fptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(_FP)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class Context:
    def __init__(self, device=0, stream=None):
        h = _P()
        if stream is None:
            check(lib().snnhip_ctx_create(device, C.byref(h)))
        else:
            check(lib().snnhip_ctx_create_on_stream(device, _P(stream), C.byref(h)))
        self.h = h
        self.device = device

    def info(self):
        di = DeviceInfo()
        check(lib().snnhip_ctx_info(self.h, C.byref(di)))
        return {"name": di.name.decode(), "compute_units": di.compute_units, "lds_bytes_per_cu": di.lds_bytes_per_cu, "hbm_bytes": di.hbm_bytes}

    def sync(self):
        check(lib().snnhip_sync(self.h))

    def stream(self):
        return lib().snnhip_ctx_stream(self.h)

    def fork(self):
        """plans run from now on go to the context's side stream (it first waits for the main stream's work so far)"""
        check(lib().snnhip_ctx_fork(self.h))

    def main(self):
        check(lib().snnhip_ctx_main(self.h))

    def join(self):
        """back on the main stream, which waits for the side stream's work"""
        check(lib().snnhip_ctx_join(self.h))

    def group_begin(self):
        """groupable plans (Plan.groupable()) run from now on are launched by group_end -- two that fit one grid as ONE kernel launch"""
        check(lib().snnhip_ctx_group_begin(self.h))

    def group_end(self):
        check(lib().snnhip_ctx_group_end(self.h))

    def close(self):
        if self.h:
            lib().snnhip_ctx_destroy(self.h)
            self.h = None


F32, F16, U8 = 0, 1, 2  # SNNHIP_F32 / SNNHIP_F16 / SNNHIP_U8 (8-bit image input of image_u8_plan only)


class Tensor:
    """NHWC tensor in HBM: fp32, or fp16 storage (dtype=F16; upload / numpy() still speak float32 and convert at the edge)."""

    def __init__(self, ctx, n, h, w, c, device_ptr=None, keepalive=None, dtype=F32):
        self.ctx = ctx
        self.shape = (n, h, w, c)
        self.dtype = dtype
        hh = _P()
        if device_ptr is None:
            check(lib().snnhip_tensor_alloc(ctx.h, n, h, w, c, dtype, C.byref(hh)))
        else:
            check(lib().snnhip_tensor_wrap(ctx.h, _P(device_ptr), n, h, w, c, dtype, C.byref(hh)))
        self.h = hh
        self._keepalive = keepalive

    @staticmethod
    def from_numpy(ctx, a, dtype=F32):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.ndim == 4, "expected NHWC"
        t = Tensor(ctx, *a.shape, dtype=dtype)
        t.upload(a)
        return t

    @staticmethod
    def from_torch(ctx, tt):
        """Borrows the storage of a contiguous NHWC float32 torch tensor living on the context's device."""
        assert tt.is_contiguous() and tt.dim() == 4 and str(tt.dtype) == "torch.float32"
        n, h, w, c = tt.shape
        return Tensor(ctx, n, h, w, c, device_ptr=tt.data_ptr(), keepalive=tt)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == int(np.prod(self.shape)), (a.shape, self.shape)
        check(lib().snnhip_tensor_upload(self.h, _fptr(a)))

    def numpy(self):
        out = np.empty(self.shape, dtype=np.float32)
        check(lib().snnhip_tensor_download(self.h, _fptr(out)))
        return out

    def upload_u8(self, a):
        """Raw byte upload of an 8-bit image tensor (dtype=U8)."""
        a = np.ascontiguousarray(a, dtype=np.uint8)
        assert self.dtype == U8 and a.size == int(np.prod(self.shape)), (a.shape, self.shape)
        check(lib().snnhip_tensor_upload_raw(self.h, a.ctypes.data_as(_P), a.size))

    def argmax(self, n=0):
        """Index of the largest element of image n (first on ties) -- the reference reports this + 1 as classifierOutput (core.cpp:228-234)."""
        out = C.c_int(-1)
        check(lib().snnhip_tensor_argmax(self.h, n, C.byref(out)))
        return out.value

    def upload_c4hw4(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        check(lib().snnhip_tensor_upload_c4hw4(self.h, _fptr(a)))

    def numpy_c4hw4(self):
        n, h, w, c = self.shape
        out = np.empty((n, (c + 3) // 4, h, w, 4), dtype=np.float32)
        check(lib().snnhip_tensor_download_c4hw4(self.h, _fptr(out)))
        return out

    def fill(self, v):
        check(lib().snnhip_tensor_fill(self.h, float(v)))

    def data_ptr(self):
        return lib().snnhip_tensor_data(self.h)

    def free(self):
        if self.h:
            lib().snnhip_tensor_free(self.h)
            self.h = None


class Plan:
    def __init__(self, ctx, handle, keep=()):
        self.ctx = ctx
        self.h = handle
        self._keep = keep

    def out_shape(self):
        d = (C.c_int * 4)()
        check(lib().snnhip_plan_output_dims(self.h, C.byref(d)))
        return tuple(d)

    def describe(self):
        buf = C.create_string_buffer(2048)
        check(lib().snnhip_plan_describe(self.h, buf, 2048))
        return buf.value.decode()

    def cost(self):
        f, b = C.c_double(), C.c_double()
        check(lib().snnhip_plan_cost(self.h, C.byref(f), C.byref(b)))
        return f.value, b.value

    def run(self, x, y):
        if isinstance(x, (list, tuple)):
            arr = (_P * len(x))(*[t.h for t in x])
            check(lib().snnhip_plan_run_n(self.h, arr, len(x), y.h))
        else:
            check(lib().snnhip_plan_run(self.h, x.h, y.h))

    def __call__(self, x, y=None):
        if y is None:
            first = x[0] if isinstance(x, (list, tuple)) else x
            y = Tensor(self.ctx, *self.out_shape(), dtype=first.dtype)
        self.run(x, y)
        return y

    def groupable(self):
        return bool(lib().snnhip_plan_groupable(self.h))

    def num_steps(self):
        return lib().snnhip_plan_num_steps(self.h)

    def step_describe(self, i):
        buf = C.create_string_buffer(1024)
        check(lib().snnhip_plan_step_describe(self.h, i, buf, 1024))
        return buf.value.decode()

    def step_cost(self, i):
        f, b = C.c_double(), C.c_double()
        check(lib().snnhip_plan_step_cost(self.h, i, C.byref(f), C.byref(b)))
        return f.value, b.value

    def profile(self, enable=True):
        check(lib().snnhip_plan_profile_enable(self.h, int(enable)))

    def profile_read(self, i):
        """(total_ms, launches) of step i since the last read."""
        ms, n = C.c_double(), C.c_int()
        check(lib().snnhip_plan_profile_read(self.h, i, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def destroy(self):
        if self.h:
            lib().snnhip_plan_destroy(self.h)
            self.h = None


def _conv_desc(N, H, W, IC, OC, k, stride, pads, pad_mode, act, leaky, use_bias, use_bn, OH=0, OW=0):
    d = ConvDesc()
    d.N, d.H, d.W, d.IC, d.OC = N, H, W, IC, OC
    kh, kw = (k, k) if isinstance(k, int) else k
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    d.kh, d.kw, d.sh, d.sw = kh, kw, sh, sw
    d.padT, d.padB, d.padL, d.padR = pads
    d.padMode = PAD_MODE[pad_mode] if isinstance(pad_mode, str) else pad_mode
    d.act = ACT[act] if isinstance(act, str) else act
    d.leaky = leaky
    d.useBias, d.useBN, d.dtype, d.OH, d.OW = int(use_bias), int(use_bn), 0, OH, OW
    return d


def same_padding(k):
    """Conv2DLayer::getPaddingOffset for padding "same" (reference conv2d.cpp:57-66): returns (T, B, L, R)."""
    if k > 1:
        p = max(k // 2, 1)
        t = [p, p, p, p]
        if k % 2 == 0:
            t[0] -= 1
            t[2] -= 1
        return tuple(t)
    return (0, 0, 0, 0)


def conv2d_plan(ctx, N, H, W, w_oihw, bias=None, stride=1, pads=None, pad_mode="constant", act="", leaky=0.0, bn=None, depthwise=False, dtype=F32):
    """w_oihw: [OC][IC][kh][kw] (depthwise: [C][kh][kw]). bn: dict beta/gamma/mean/var or None. pads: (T,B,L,R) or None=same."""
    w = _f32(w_oihw)
    if depthwise:
        OC, kh, kw = w.shape
        IC = OC
    else:
        OC, IC, kh, kw = w.shape
    if pads is None:
        pads = same_padding(kh)
    b = _f32(bias)
    bnp = [None] * 4
    if bn is not None:
        bnp = [_f32(bn[k]) for k in ("beta", "gamma", "mean", "var")]
    d = _conv_desc(N, H, W, IC, OC, (kh, kw), stride, pads, pad_mode, act, leaky, b is not None, bn is not None)
    d.dtype = dtype
    h = _P()
    fn = lib().snnhip_depthwise_plan_create if depthwise else lib().snnhip_conv2d_plan_create
    check(fn(ctx.h, C.byref(d), _fptr(w), _fptr(b), _fptr(bnp[0]), _fptr(bnp[1]), _fptr(bnp[2]), _fptr(bnp[3]), C.byref(h)))
    return Plan(ctx, h)


def dense_plan(ctx, batch, w_flat, out_units, bias=None, act="", leaky=0.0):
    w = _f32(w_flat).reshape(-1)
    in_units = w.size // out_units
    d = DenseDesc(batch, in_units, out_units, DENSE_ACT[act] if isinstance(act, str) else act, leaky, int(bias is not None))
    b = _f32(bias)
    h = _P()
    check(lib().snnhip_dense_plan_create(ctx.h, C.byref(d), _fptr(w), _fptr(b), C.byref(h)))
    return Plan(ctx, h)


def subpixel_plan(ctx, N, H, W, Cc, factor=2, mode=0):
    d = SubpixelDesc(N, H, W, Cc, factor, mode)
    h = _P()
    check(lib().snnhip_subpixel_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def _act_id(act):
    ids = {"": 0, "linear": 0, "none": 0, "relu": 1, "relu6": 2, "tanh": 3, "sigmoid": 4, "leakyRelu": 5, "SiLU": 6}
    return ids[act]


def add_plan(ctx, N, H, W, Cc, act="", leaky=0.0):
    """y = act(a + b); run with plan([a, b]).  (AddLayerVulkan, vk_add.comp)"""
    d = EltwiseDesc(N, H, W, Cc, _act_id(act), leaky)
    h = _P()
    check(lib().snnhip_add_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def activation_plan(ctx, N, H, W, Cc, act, leaky=0.0):
    d = EltwiseDesc(N, H, W, Cc, _act_id(act), leaky)
    h = _P()
    check(lib().snnhip_activation_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def batchnorm_plan(ctx, N, H, W, Cc, bn, act="", leaky=0.0):
    d = EltwiseDesc(N, H, W, Cc, _act_id(act), leaky)
    arrs = [_f32(bn[k]) for k in ("beta", "gamma", "mean", "var")]
    h = _P()
    check(lib().snnhip_batchnorm_plan_create(ctx.h, C.byref(d), *[_fptr(a) for a in arrs], C.byref(h)))
    return Plan(ctx, h)


def pool2d_plan(ctx, N, H, W, Cc, k, stride, kind="max", same=True, pad_t=0, pad_l=0, OH=0, OW=0):
    """kind 'max' | 'avg'.  `same` only selects the reference's output-size rule; the window is always clipped to the image."""
    d = Pool2dDesc(N, H, W, Cc, k, k, stride, stride, pad_t, pad_l, OH, OW, int(bool(same)), 0 if kind == "max" else 1)
    h = _P()
    check(lib().snnhip_pool2d_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def global_avgpool_plan(ctx, N, H, W, Cc):
    """AdaptiveAvgPool2d with target 1: the mean over the whole image (adaptiveavgpool2dGL.cpp)."""
    d = Pool2dDesc(N, H, W, Cc, H, W, H, W, 0, 0, 1, 1, 0, 1)
    h = _P()
    check(lib().snnhip_pool2d_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def pad_plan(ctx, N, H, W, Cc, pads, mode="constant"):
    """pads = (T, B, L, R); mode constant | replicate | reflect."""
    d = PadDesc(N, H, W, Cc, pads[0], pads[1], pads[2], pads[3], {"constant": 0, "replicate": 1, "reflect": 2}[mode])
    h = _P()
    check(lib().snnhip_pad_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def upsample_plan(ctx, N, H, W, Cc, scale=2.0, mode="nearest"):
    d = UpsampleDesc(N, H, W, Cc, scale, {"nearest": 0, "bilinear": 1}[mode])
    h = _P()
    check(lib().snnhip_upsample_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def instancenorm_plan(ctx, N, H, W, Cc, beta, gamma, act="", leaky=0.0, eps=1e-5):
    d = InstanceNormDesc(N, H, W, Cc, _act_id(act), leaky, eps)
    b, g = _f32(beta), _f32(gamma)
    h = _P()
    check(lib().snnhip_instancenorm_plan_create(ctx.h, C.byref(d), _fptr(b), _fptr(g), C.byref(h)))
    return Plan(ctx, h)


def concat_plan(ctx, N, H, W, C0, C1, OC=None):
    d = ConcatDesc(N, H, W, C0, C1, C0 + C1 if OC is None else OC)
    h = _P()
    check(lib().snnhip_concat_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


UNARY_OPS = {"copy": 0, "fixed": 1, "neg": 2, "rcp": 3, "square": 4, "exp": 5, "abs": 6}


def unary_plan(ctx, N, H, W, Cc, op="copy", value=1.0):
    d = UnaryDesc(N, H, W, Cc, UNARY_OPS[op] if isinstance(op, str) else int(op), value)
    h = _P()
    check(lib().snnhip_unary_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def calculate_plan(ctx, N, H, W, Cc, OC):
    d = CalculateDesc(N, H, W, Cc, OC)
    h = _P()
    check(lib().snnhip_calculate_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def resize_plan(ctx, N, H, W, Cc, OH, OW, means=(0, 0, 0, 0), norms=(1, 1, 1, 1), linear=True):
    d = ResizeDesc(N, H, W, Cc, OH, OW, (C.c_float * 4)(*means), (C.c_float * 4)(*norms), 1 if linear else 0)
    h = _P()
    check(lib().snnhip_resize_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def image_u8_plan(ctx, N, H, W, src_channels, means=(0, 0, 0, 0), norms=(1, 1, 1, 1)):
    d = ImageU8Desc(N, H, W, src_channels, (C.c_float * 4)(*means), (C.c_float * 4)(*norms))
    h = _P()
    check(lib().snnhip_image_u8_plan_create(ctx.h, C.byref(d), C.byref(h)))
    return Plan(ctx, h)


def deconv2d_plan(ctx, N, H, W, w_oihw, bias=None, stride=2, same=True, act="", leaky=0.0, bn=None):
    """Conv2DTranspose: w_oihw [OC][IC][k][k]; output s*H ("same") or s*H + k - s (deconv2dGL.cpp:345-355)."""
    w = _f32(w_oihw)
    OC, IC, k, _ = w.shape
    p = (k - stride) // 2 if same else 0
    OH, OW = (stride * H, stride * W) if same else (stride * H + k - stride, stride * W + k - stride)
    d = _conv_desc(N, H, W, IC, OC, k, stride, (p, 0, 0, 0), "constant", act, leaky, bias is not None, bn is not None, OH, OW)
    b = _f32(bias) if bias is not None else None
    bnp = [_f32(bn[key]) for key in ("beta", "gamma", "mean", "var")] if bn is not None else [None] * 4
    h = _P()
    check(lib().snnhip_deconv2d_plan_create(ctx.h, C.byref(d), _fptr(w), _fptr(b), *[_fptr(a) for a in bnp], C.byref(h)))
    return Plan(ctx, h)


def chain_plan(ctx, plans):
    arr = (_P * len(plans))(*[p.h for p in plans])
    h = _P()
    check(lib().snnhip_chain_plan_create(ctx.h, arr, len(plans), C.byref(h)))
    return Plan(ctx, h, keep=tuple(plans))


def graph_fuse(ctx, nodes):
    """snnhip_graph_fuse: nodes = [(Plan, [producer index | -(k+1) for model input k], keep)] in execution order.
    Returns [(Plan | None, inputs)] per node: None = folded into a later node's plan, the same Plan object = unchanged, a new Plan = fused."""
    n = len(nodes)
    arr = (GraphNode * n)()
    for i, (plan, ins, keep) in enumerate(nodes):
        arr[i].plan = plan.h if plan is not None else None
        arr[i].n_inputs = len(ins)
        for k, v in enumerate(ins):
            arr[i].inputs[k] = v
        arr[i].keep = int(bool(keep))
    out = (FusedNode * n)()
    check(lib().snnhip_graph_fuse(ctx.h, arr, n, out))
    res = []
    for i, (plan, ins, _) in enumerate(nodes):
        o = out[i]
        if not o.plan:
            res.append((None, []))
        elif o.owned:
            res.append((Plan(ctx, _P(o.plan), keep=tuple(q for q, _, _ in nodes)), [o.inputs[k] for k in range(o.n_inputs)]))  # fused plans borrow their members
        else:
            res.append((plan, [o.inputs[k] for k in range(o.n_inputs)]))  # the node's own plan, possibly re-wired past a folded identity producer
    return res


class Graph:
    """A captured launch sequence (hipGraph): `with Graph.capture(ctx) as g: runner.run_device()`, then g.launch() replays it."""

    def __init__(self, ctx):
        self.ctx, self.h = ctx, _P()

    class _Cap:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            check(lib().snnhip_graph_begin_capture(self.g.ctx.h))
            return self.g

        def __exit__(self, et, ev, tb):
            rc = lib().snnhip_graph_end_capture(self.g.ctx.h, C.byref(self.g.h))
            if et is None:
                check(rc)
            return False

    @staticmethod
    def capture(ctx):
        return Graph._Cap(Graph(ctx))

    def launch(self):
        check(lib().snnhip_graph_launch(self.h))

    def num_nodes(self):
        return lib().snnhip_graph_num_nodes(self.h)

    def destroy(self):
        if self.h:
            lib().snnhip_graph_destroy(self.h)
            self.h = _P()


class Timer:
    def __init__(self, ctx):
        self.h = _P()
        check(lib().snnhip_timer_create(ctx.h, C.byref(self.h)))

    def start(self):
        check(lib().snnhip_timer_start(self.h))

    def stop(self):
        check(lib().snnhip_timer_stop(self.h))

    def elapsed_ms(self):
        ms = C.c_float()
        check(lib().snnhip_timer_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.h:
            lib().snnhip_timer_destroy(self.h)
            self.h = None
