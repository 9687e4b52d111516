"""ctypes doorways used by the tests only: the CPU restatement (oracle/liboracle.so) and,
where it has been built, the reference compiled from its own sources (oracle/_ref/*.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def w8(w):
    return (w + 7) // 8 * 8


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_orc = None


def oracle():
    global _orc
    if _orc is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "yolo2_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build_oracle()
        lib = C.CDLL(so)
        lib.orc_conv_i16.argtypes = [_i16p, _i16p, _i16p, _i16p] + [C.c_int] * 14
        lib.orc_conv_f32.argtypes = [_f32p, _f32p, _f32p, _f32p] + [C.c_int] * 10
        lib.orc_maxpool_i16.argtypes = [_i16p, _i16p] + [C.c_int] * 7
        lib.orc_maxpool_f32.argtypes = [_f32p, _f32p] + [C.c_int] * 7
        lib.orc_reorg_i16.argtypes = [_i16p, _i16p, C.c_int]
        lib.orc_reorg_f32.argtypes = [_f32p, _f32p]
        lib.orc_quantize_input.argtypes = [_f32p, _i16p, C.c_size_t, C.c_int]
        lib.orc_leaky_i16.argtypes = [C.c_int16]
        lib.orc_leaky_i16.restype = C.c_int16
        lib.orc_region_forward.argtypes = [_f32p, _f32p]
        lib.orc_yolov2_forward_i16.restype = C.c_int
        lib.orc_yolov2_forward_f32.restype = C.c_int
        _orc = lib
    return _orc


class OrcWeightsI16(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("bias", C.c_void_p), ("weight_q", C.c_void_p),
                ("bias_q", C.c_void_p), ("act_q", C.c_void_p),
                ("n_weight_q", C.c_int), ("n_bias_q", C.c_int), ("n_act_q", C.c_int)]


class OrcWeightsF32(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("bias", C.c_void_p)]


def conv_i16(x, w_reorg, bias, C_, N, K, stride, W, H, pad, leaky, Qw, Qa_in, Qa_out, Qb, fill=0):
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    out = np.full((N, OH, w8(OW)), fill, dtype=np.int16)
    oracle().orc_conv_i16(np.ascontiguousarray(x), out, np.ascontiguousarray(w_reorg),
                          np.ascontiguousarray(bias), C_, N, K, stride, W, H, OW, OH, pad,
                          int(leaky), Qw, Qa_in, Qa_out, Qb)
    return out


def conv_f32(x, w_reorg, bias, C_, N, K, stride, W, H, pad, leaky):
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    out = np.zeros((N, OH, w8(OW)), dtype=np.float32)
    oracle().orc_conv_f32(np.ascontiguousarray(x), out, np.ascontiguousarray(w_reorg),
                          np.ascontiguousarray(bias), C_, N, K, stride, W, H, OW, OH, pad, int(leaky))
    return out


def maxpool(x, C_, W, H, K=2, stride=2):
    OW, OH = W // stride, H // stride
    out = np.zeros((C_, OH, w8(OW)), dtype=x.dtype)
    fn = oracle().orc_maxpool_i16 if x.dtype == np.int16 else oracle().orc_maxpool_f32
    fn(np.ascontiguousarray(x), out, C_, K, stride, W, H, OW, OH)
    return out


def forward_i16(model, frame, dump=False):
    """model: yolo2_amd.synth.SynthModel; frame float32 [3][416][416]."""
    w = model.weights_i16(); b = model.bias_i16()
    wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (model.weight_q, model.bias_q, model.act_q))
    wp = OrcWeightsI16(w.ctypes.data, b.ctypes.data, wq.ctypes.data, bq.ctypes.data, aq.ctypes.data,
                       len(wq), len(bq), len(aq))
    reg_i = np.zeros(425 * 169, dtype=np.int16)
    reg_f = np.zeros(425 * 169, dtype=np.float32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    dumps = (C.c_void_p * 32)() if dump else None
    q = oracle().orc_yolov2_forward_i16(C.byref(wp), frame.ctypes.data_as(C.c_void_p),
                                        reg_i.ctypes.data_as(C.c_void_p), reg_f.ctypes.data_as(C.c_void_p),
                                        dumps)
    assert q >= 0
    if dump:
        from yolo2_amd import net
        layers = {}
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        for l in net.LAYERS:
            if dumps[l.idx]:
                n = l.out_c * l.out_h * w8(l.out_w)
                arr = np.ctypeslib.as_array(C.cast(dumps[l.idx], C.POINTER(C.c_int16)), shape=(n,)).copy()
                layers[l.idx] = arr.reshape(l.out_c, l.out_h, w8(l.out_w))
                libc.free(dumps[l.idx])
        return reg_i, reg_f, q, layers
    return reg_i, reg_f, q


def forward_f32(model, frame):
    w = model.weights_f32(); b = model.bias_f32()
    wp = OrcWeightsF32(w.ctypes.data, b.ctypes.data)
    reg_f = np.zeros(425 * 169, dtype=np.float32)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    rc = oracle().orc_yolov2_forward_f32(C.byref(wp), frame.ctypes.data_as(C.c_void_p),
                                         reg_f.ctypes.data_as(C.c_void_p), None)
    assert rc == 0
    return reg_f


# ------------------------------------------------------------------ compiled reference

def have_ref():
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in ("libref_int16.so", "libref_fp32.so"))


_refs = {}


def ref(int16=True):
    key = "int16" if int16 else "fp32"
    if key not in _refs:
        lib = C.CDLL(os.path.join(REF_DIR, f"libref_{key}.so"))
        lib.ref_YOLO2_FPGA.argtypes = [C.c_void_p] * 4 + [C.c_int] * 23
        lib.ref_yolov2_hls_ps.argtypes = [C.c_char_p, _f32p, _f32p]
        lib.ref_yolov2_hls_ps.restype = C.c_int
        _refs[key] = lib
    return _refs[key]


def ref_conv(x, w_reorg, bias, C_, N, K, stride, W, H, pad, leaky, Qw=0, Qa_in=0, Qa_out=0, Qb=0, fill=0):
    """Calls the reference's YOLO2_FPGA exactly as yolov2_hls_ps does for a conv layer
    (hls/models/yolov2/yolo2_model.cpp:299-327)."""
    int16 = x.dtype == np.int16
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    TR = min((27 - K) // stride + 1, 13); TR = min(OH, TR)
    TC = min((27 - K) // stride + 1, 13); TC = min(OW, TC)
    TM = min(N, 32); TN = min(C_, 4)
    mLoops = -(-N // TM)
    # slack around the input like the reference arena (yolo2_model.cpp:243-244): halo loads
    # read (and discard) a few elements before/after the tensor
    slack = 2048
    xin = np.zeros(x.size + 2 * slack, dtype=x.dtype)
    xin[slack:slack + x.size] = x.reshape(-1)
    out = np.full(N * OH * w8(OW) + 64, fill, dtype=x.dtype)
    wr = np.ascontiguousarray(np.concatenate([w_reorg.reshape(-1), np.zeros(64, dtype=x.dtype)]))
    bb = np.ascontiguousarray(np.concatenate([bias.reshape(-1), np.zeros(1024, dtype=x.dtype)]))
    ref(int16).ref_YOLO2_FPGA(xin.ctypes.data + slack * x.itemsize, out.ctypes.data, wr.ctypes.data,
                              bb.ctypes.data, C_, N, K, stride, W, H, OW, OH, pad, int(leaky), 0,
                              TM, TN, TR, TC, (mLoops + 1) * TM, mLoops * TM, (mLoops + 1) * TM, 0,
                              Qw, Qa_in, Qa_out, Qb)
    return out[:N * OH * w8(OW)].reshape(N, OH, w8(OW))


def ref_maxpool(x, C_, W, H):
    """yolo2_model.cpp:341-357"""
    int16 = x.dtype == np.int16
    OW, OH = W // 2, H // 2
    TR = min((27 - 2) // 2 + 1, 13); TC = TR
    TR = min(OH, TR); TC = min(OW, TC)
    TM = min(min(32, 4), C_)
    mLoops = -(-C_ // TM)
    slack = 2048
    xin = np.zeros(x.size + 2 * slack, dtype=x.dtype)
    xin[slack:slack + x.size] = x.reshape(-1)
    out = np.zeros(C_ * OH * w8(OW) + 64, dtype=x.dtype)
    ref(int16).ref_YOLO2_FPGA(xin.ctypes.data + slack * x.itemsize, out.ctypes.data, None, None,
                              C_, C_, 2, 2, W, H, OW, OH, 1, 0, 0, TM, 0, TR, TC,
                              (mLoops + 2) * TM, mLoops * TM, (mLoops + 1) * TM, 1, 0, 0, 0, 0)
    return out[:C_ * OH * w8(OW)].reshape(C_, OH, w8(OW))


# ------------------------------------------------------------------ host logic (ours / reference)

HOST_LIB = os.path.join(ROOT, "yolo-fpga-accelerator_amd", "libyolo2_host.so")
_host = None


def host():
    global _host
    if _host is None:
        lib = C.CDLL(HOST_LIB)
        lib.y2h_last_error.restype = C.c_char_p
        lib.y2h_letterbox.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        lib.y2h_region_forward.argtypes = [_f32p, _f32p]
        lib.y2h_boxes_nms.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float, _f32p, C.c_int]
        lib.y2h_parse_cfg.argtypes = [C.c_char_p, _i32p, _i32p, C.c_int, _f32p, _i32p]
        lib.y2h_load_pnm.argtypes = [C.c_char_p, _i32p, _f32p, C.c_long]
        lib.y2h_postprocess_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, _i32p, C.c_float, C.c_float, C.c_int,
                                              _f32p, C.c_int, _i32p]
        _host = lib
    return _host


def ref_host():
    lib = ref(True)
    lib.ref_letterbox.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
    lib.ref_detect.argtypes = [C.c_char_p, _f32p, C.c_int, C.c_int, C.c_float, C.c_float, _f32p, _f32p, C.c_int]
    lib.ref_parse_cfg.argtypes = [C.c_char_p, _i32p, _i32p, C.c_int]
    return lib


def canon_rows(rows):
    """Detections with objectness > 0, in a canonical order (the reference's qsort is unstable)."""
    r = rows[rows[:, 4] > 0]
    return r[np.lexsort((r[:, 3], r[:, 2], r[:, 1], r[:, 0]))]
