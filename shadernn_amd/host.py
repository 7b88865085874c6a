"""ctypes binding of include/snn_c.h (libsnn_core.so): the C++ host mirror of the reference's MixedInferenceCore API."""
import ctypes as C
import os

import numpy as np

from . import capi

# SNN_CORE_LIB_PATH: another build of the host library (tools/sanitize.sh points it at lib/libsnn_core_asan.so)
LIB_PATH = os.environ.get("SNN_CORE_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsnn_core.so")
_lib = None
_P = C.c_void_p
_FP = C.POINTER(C.c_float)

SIGNATURES = {
    "snn_model_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snn_model_create2": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snn_model_create3": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snn_model_create4": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snn_model_batch": (C.c_int, [_P]),
    "snn_model_hip_ctx": (_P, [_P]),
    "snn_model_output_tensor": (_P, [_P]),
    "snn_pool_create": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "snn_pool_destroy": (C.c_int, [_P]),
    "snn_pool_replicas": (C.c_int, [_P]),
    "snn_pool_shard": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "snn_pool_output_dims": (C.c_int, [_P, C.POINTER(C.c_int * 3)]),
    "snn_pool_upload_input": (C.c_int, [_P, _FP]),
    "snn_pool_run": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "snn_pool_download_output": (C.c_int, [_P, _FP]),
    "snn_pool_allgather_output_rccl": (C.c_int, [_P, _FP]),
    "snn_model_stage_plan_steps": (C.c_int, [_P, C.c_int]),
    "snn_model_stage_plan_step": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "snn_model_profile_enable": (C.c_int, [_P, C.c_int]),
    "snn_model_profile_read": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "snn_model_suspend_replay": (C.c_int, [_P, C.c_int]),
    "snn_model_cost": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "snn_model_destroy": (C.c_int, [_P]),
    "snn_model_upload_input": (C.c_int, [_P, _FP]),
    "snn_model_run": (C.c_int, [_P]),
    "snn_model_run_async": (C.c_int, [_P]),
    "snn_model_sync": (C.c_int, [_P]),
    "snn_model_output_dims": (C.c_int, [_P, C.POINTER(C.c_int * 3)]),
    "snn_model_download_output": (C.c_int, [_P, _FP]),
    "snn_model_set_type": (C.c_int, [_P, C.c_int]),
    "snn_model_classifier_output": (C.c_int, [_P]),
    "snn_model_detections": (C.c_int, [_P, _FP, C.c_int]),
    "snn_model_upload_input_u8": (C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.c_int, _FP, _FP, _FP, _FP]),
    "snn_yolo_decode": (C.c_int, [_FP, _FP, C.c_int, _FP, C.c_int]),
    "snn_model_num_stages": (C.c_int, [_P]),
    "snn_model_stage_info": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int * 3), C.POINTER(C.c_int)]),
    "snn_model_download_stage": (C.c_int, [_P, C.c_int, _FP]),
    "snn_model_describe": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "snn_model_time_stats": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "snn_conv_test_with_layer": (C.c_int, [C.c_int, _FP, _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _FP, _FP, _FP, _FP,
                                           C.c_char_p, C.c_int]),
    "snn_graph_summary": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "snn_json_number": (C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "snn_dump_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int * 4), _FP, C.c_long]),
}


def lib():
    global _lib
    if _lib is None:
        capi.load_library()  # libsnnhip.so first (libsnn_core.so links against it)
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(_FP)


def graph_summary(json_path, w, h, c):
    buf = C.create_string_buffer(1 << 16)
    n = lib().snn_graph_summary(json_path.encode(), w, h, c, buf, len(buf))
    rows = []
    for line in buf.value.decode().strip().split("\n"):
        idx, name, loc, dims, ins = line.split("|")
        rows.append({"index": int(idx), "name": name, "loc": int(loc), "dims": tuple(int(v) for v in dims.split("x")),
                     "inputs": [int(v) for v in ins.split(",") if v]})
    assert len(rows) == n
    return rows


def json_number(text):
    """The model loader's JSON parser on one number literal (None if it refuses it)."""
    d = C.c_double()
    return d.value if lib().snn_json_number(text.encode(), C.byref(d)) == 0 else None


def read_dump(path):
    """Reference .dump -> (W, H, D, C, float32 array [D][H][W][4])."""
    d = (C.c_int * 4)()
    assert lib().snn_dump_read(path.encode(), C.byref(d), None, 0) == 0
    w, h, dd, c = d
    out = np.empty((dd, h, w, 4), dtype=np.float32)
    assert lib().snn_dump_read(path.encode(), C.byref(d), _fp(out), out.size) == 0
    return w, h, dd, c, out


def c4hw4_to_nhwc(a, channels):
    d, h, w, _ = a.shape
    return np.transpose(a, (1, 2, 0, 3)).reshape(h, w, d * 4)[:, :, :channels]


def yolo_decode(head_coarse, head_fine, net_size=416, max_rows=100):
    """YOLOLayer's CPU decode + NMS (host-only)."""
    a, b = np.ascontiguousarray(head_coarse, dtype=np.float32), np.ascontiguousarray(head_fine, dtype=np.float32)
    rows = np.zeros((max_rows, 6), np.float32)
    n = lib().snn_yolo_decode(_fp(a), _fp(b), net_size, _fp(rows), max_rows)
    return rows[:n].copy()


class Model:
    """MixedInferenceCore::create(context, jsonFile, options) + run(), one W x H x C input image."""

    def __init__(self, json_path, w, h, c, device=0, dump_outputs=False, fuse_chains=True, profiling=False, prefer_half=False, capture_graph=False,
                 batch=1):
        """batch > 1: every stage tensor carries `batch` images (snn_model_create4); upload() takes and output() returns a leading batch axis."""
        self.h = _P()
        assert lib().snn_model_create4(json_path.encode(), device, w, h, c, int(dump_outputs), int(fuse_chains), int(profiling), int(prefer_half),
                                       int(capture_graph), int(batch), C.byref(self.h)) == 0
        self.batch = batch
        self.in_shape = (h, w, c) if batch == 1 else (batch, h, w, c)

    def upload(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.in_shape)
        assert lib().snn_model_upload_input(self.h, _fp(x)) == 0

    def upload_u8(self, img, means=(0, 0, 0, 0), norms=(1, 1, 1, 1), resize_means=(0, 0, 0, 0), resize_norms=(1, 1, 1, 1)):
        """8-bit H x W x {1,3,4} image -> normalise -> bilinear resize to the model input, all on the device (modelInference.cpp:92-97)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        ih, iw, ic = img.shape
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (means, norms, resize_means, resize_norms)]
        rc = lib().snn_model_upload_input_u8(self.h, img.ctypes.data_as(C.c_void_p), iw, ih, ic, *[_fp(a) for a in arrs])
        assert rc == 0, rc

    MODEL_TYPES = {"classification": 0, "detection": 1, "segmentation": 2, "other": 3}

    def set_type(self, model_type):
        assert lib().snn_model_set_type(self.h, self.MODEL_TYPES[model_type]) == 0

    def classifier_output(self):
        """1-based class index of the last run (MixedInferenceCore::run, core.cpp:228-234); 0 = none."""
        return lib().snn_model_classifier_output(self.h)

    def detections(self, max_rows=100):
        rows = np.zeros((max_rows, 6), np.float32)
        n = lib().snn_model_detections(self.h, _fp(rows), max_rows)
        return rows[:n].copy()

    def run(self):
        assert lib().snn_model_run(self.h) == 0

    def run_async(self):
        """RunParameters::deferSync: enqueue one inference, no wait (pair with sync())."""
        assert lib().snn_model_run_async(self.h) == 0

    def sync(self):
        assert lib().snn_model_sync(self.h) == 0

    def output(self):
        d = (C.c_int * 3)()
        lib().snn_model_output_dims(self.h, C.byref(d))
        out = np.empty(tuple(d) if self.batch == 1 else (self.batch,) + tuple(d), dtype=np.float32)
        assert lib().snn_model_download_output(self.h, _fp(out)) == 0
        return out

    def __call__(self, x):
        self.upload(x)
        self.run()
        return self.output()

    def stages(self):
        out = []
        for i in range(lib().snn_model_num_stages(self.h)):
            name = C.create_string_buffer(512)
            d = (C.c_int * 3)()
            fused = C.c_int()
            lib().snn_model_stage_info(self.h, i, name, 512, C.byref(d), C.byref(fused))
            out.append({"name": name.value.decode(), "hwc": tuple(d), "fused_away": bool(fused.value & 1), "side": bool(fused.value & 2), "group": bool(fused.value & 4)})
        return out

    def stage_output(self, i):
        st = self.stages()[i]
        out = np.empty(st["hwc"] if self.batch == 1 else (self.batch,) + tuple(st["hwc"]), dtype=np.float32)
        if lib().snn_model_download_stage(self.h, i, _fp(out)) != 0:
            return None
        return out

    def plan_steps(self):
        """[(stage, step, kernel description, flops, bytes)] of every kernel launch of one inference, after fusion."""
        out = []
        for i in range(lib().snn_model_num_stages(self.h)):
            for k in range(lib().snn_model_stage_plan_steps(self.h, i)):
                desc = C.create_string_buffer(1024)
                f, b = C.c_double(), C.c_double()
                assert lib().snn_model_stage_plan_step(self.h, i, k, desc, 1024, C.byref(f), C.byref(b)) == 0
                out.append((i, k, desc.value.decode(), f.value, b.value))
        return out

    def profile(self, enable=True):
        assert lib().snn_model_profile_enable(self.h, int(enable)) == 0

    def suspend_replay(self, suspend=True):
        """launch by launch instead of replaying the recorded hipGraph (a launch trace needs the plans to run)"""
        assert lib().snn_model_suspend_replay(self.h, int(suspend)) == 0

    def profile_read(self, stage, step):
        ms, n = C.c_double(), C.c_int()
        assert lib().snn_model_profile_read(self.h, stage, step, C.byref(ms), C.byref(n)) == 0
        return ms.value, n.value

    def cost(self):
        f, b = C.c_double(), C.c_double()
        assert lib().snn_model_cost(self.h, C.byref(f), C.byref(b)) == 0
        return f.value, b.value

    def describe(self):
        buf = C.create_string_buffer(1 << 14)
        lib().snn_model_describe(self.h, buf, len(buf))
        return buf.value.decode()

    def time_stats(self):
        names = C.create_string_buffer(1 << 14)
        ms = (C.c_double * 256)()
        n = lib().snn_model_time_stats(self.h, names, len(names), ms, 256)
        return dict(zip(names.value.decode().strip().split("\n"), list(ms)[:n]))

    def close(self):
        if self.h:
            lib().snn_model_destroy(self.h)
            self.h = None


def conv_test_with_layer(x_hwc, w_oihw, bias, stride=1, pad=0, bn=None, device=0):
    """ShaderUnitTest::snnConvTestWithLayer; returns the dump path written by the layer."""
    x = np.ascontiguousarray(x_hwc, dtype=np.float32)
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    h, ww, ic = x.shape
    oc, _, k, _ = w.shape
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    arrs = [None] * 4
    if bn is not None:
        arrs = [np.ascontiguousarray(bn[q], dtype=np.float32) for q in ("gamma", "mean", "var", "beta")]
    path = C.create_string_buffer(1024)
    rc = lib().snn_conv_test_with_layer(device, _fp(x), _fp(w), _fp(b), ww, h, ic, oc, k, stride, pad, int(bn is not None), _fp(arrs[0]), _fp(arrs[1]),
                                        _fp(arrs[2]), _fp(arrs[3]), path, 1024)
    assert rc == 0
    return path.value.decode()


class Pool:
    """snn_pool (include/snn_c.h): one replica (host thread + HipContext + stream) per entry of `devices`, a global batch split [g*B/G, (g+1)*B/G)."""

    def __init__(self, json_path, w, h, c, devices, global_batch, micro_batch=0, prefer_half=False, capture_graph=True):
        self.h = _P()
        devs = (C.c_int * len(devices))(*devices)
        rc = lib().snn_pool_create(json_path.encode(), devs, len(devices), w, h, c, int(prefer_half), int(capture_graph), global_batch, micro_batch, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("snn_pool_create failed: %d" % rc)
        self.global_batch, self.in_shape = global_batch, (global_batch, h, w, c)
        hwc = (C.c_int * 3)()
        lib().snn_pool_output_dims(self.h, C.byref(hwc))
        self.out_shape = (global_batch,) + tuple(hwc)

    def replicas(self):
        return lib().snn_pool_replicas(self.h)

    def shard(self, g):
        a, b, s = C.c_int(), C.c_int(), C.c_int()
        assert lib().snn_pool_shard(self.h, g, C.byref(a), C.byref(b), C.byref(s)) == 0
        return a.value, b.value, s.value

    def upload(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape == self.in_shape, (x.shape, self.in_shape)
        assert lib().snn_pool_upload_input(self.h, _fp(x)) == 0

    def run(self, steps=1):
        sec = C.c_double()
        rc = lib().snn_pool_run(self.h, steps, C.byref(sec))
        if rc != 0:
            raise RuntimeError("snn_pool_run failed: %d" % rc)
        return sec.value

    def output(self, rccl=False):
        out = np.empty(self.out_shape, dtype=np.float32)
        rc = (lib().snn_pool_allgather_output_rccl if rccl else lib().snn_pool_download_output)(self.h, _fp(out))
        if rc != 0:
            raise RuntimeError("snn_pool output gather failed: %d" % rc)
        return out

    def close(self):
        if self.h:
            lib().snn_pool_destroy(self.h)
            self.h = None
