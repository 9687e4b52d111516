#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE compiled from its own sources (oracle/_ref).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py            # per-layer KATs + full-network tensors

The reference ships no golden vectors, so these files pin parity:
each holds inputs and the outputs the reference's own YOLO2_FPGA / yolov2_hls_ps produced
for them.  Only data is stored here, never reference source.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))

import orclib  # noqa: E402
from yolo2_amd import synth  # noqa: E402

REF_CFG = "/root/reference/config/yolov2.cfg"


def conv_cases():
    """(name, C, N, K, stride, W, H, leaky, Qw, Qa_in, Qa_out, Qb, x_amp, w_amp, special)"""
    return [
        ("c3_first_layer_like", 3, 32, 3, 1, 26, 26, 1, 14, 14, 9, 12, 16384, 6000, None),
        ("partial_tm_tn_ragged", 7, 45, 3, 1, 17, 19, 1, 14, 9, 9, 12, 3000, 3000, None),
        ("k1_linear", 64, 40, 1, 1, 13, 13, 0, 14, 9, 9, 12, 3000, 3000, None),
        ("k1_leaky_425", 16, 425, 1, 1, 13, 13, 1, 13, 9, 10, 11, 3000, 3000, None),
        ("saturation_heavy", 8, 33, 3, 1, 14, 14, 1, 14, 9, 9, 12, 32768, 32768, None),
        ("left_shift_out", 5, 9, 3, 1, 20, 15, 1, 4, 9, 14, 12, 3000, 3000, None),
        ("left_shift_bias", 6, 6, 1, 1, 9, 9, 1, 14, 9, 9, 3, 3000, 3000, None),
        ("zero_shift", 4, 8, 3, 1, 13, 13, 1, 3, 6, 9, 9, 40, 40, None),
        ("all_min_overflow32", 4, 4, 3, 1, 5, 5, 0, 14, 9, 9, 12, 0, 0, "allmin"),
        ("mixed_extremes", 8, 8, 3, 1, 13, 13, 1, 15, 10, 9, 12, 0, 0, "extremes"),
        ("stride2", 6, 10, 3, 2, 16, 16, 1, 14, 9, 9, 12, 3000, 3000, None),
        ("mid_52", 16, 32, 3, 1, 52, 52, 1, 14, 9, 9, 12, 600, 1500, None),
        ("single_pixel", 12, 5, 1, 1, 1, 1, 1, 14, 9, 9, 12, 3000, 3000, None),
    ]


def make_conv_inputs(case, rng):
    name, C, N, K, stride, W, H, leaky, Qw, Qai, Qao, Qb, xa, wa, special = case
    W8 = orclib.w8(W)
    x = np.zeros((C, H, W8), dtype=np.int16)
    if special == "allmin":
        x[:, :, :W] = -32768
        w = np.full((N, C, K, K), -32768, dtype=np.int16)
        b = np.full(N, 32767, dtype=np.int16)
    elif special == "extremes":
        x[:, :, :W] = rng.choice(np.array([-32768, 32767, -1, 0, 1], dtype=np.int16), (C, H, W))
        w = rng.choice(np.array([-32768, 32767, -1, 0, 1], dtype=np.int16), (N, C, K, K))
        b = rng.choice(np.array([-32768, 32767, 0], dtype=np.int16), N)
    else:
        x[:, :, :W] = rng.integers(-xa, xa, (C, H, W)).clip(-32768, 32767)
        w = rng.integers(-wa, wa, (N, C, K, K)).clip(-32768, 32767).astype(np.int16)
        b = rng.integers(-20000, 20000, N).astype(np.int16)
    return x, w, b


def gen_kats():
    rng = np.random.default_rng(20260327)
    out = {}
    names = []
    for case in conv_cases():
        name, C, N, K, stride, W, H, leaky, Qw, Qai, Qao, Qb, xa, wa, special = case
        pad = 1 if K == 3 else 0
        x, w, b = make_conv_inputs(case, rng)
        wr = synth.reorg_weights(w, C, N, K)
        y = orclib.ref_conv(x, wr, b, C, N, K, stride, W, H, pad, leaky, Qw, Qai, Qao, Qb, fill=0)
        out[f"conv_i16/{name}/x"] = x
        out[f"conv_i16/{name}/w_reorg"] = wr
        out[f"conv_i16/{name}/bias"] = b
        out[f"conv_i16/{name}/params"] = np.array([C, N, K, stride, W, H, pad, leaky, Qw, Qai, Qao, Qb], dtype=np.int32)
        out[f"conv_i16/{name}/y"] = y
        names.append(name)
        print("conv_i16", name, y.shape, "saturated:", int((np.abs(y.astype(np.int32)) >= 32767).sum()))
    out["conv_i16/names"] = np.array(names)

    fnames = []
    for (name, C, N, K, W, H, leaky) in [("c3", 3, 32, 3, 26, 26, 1), ("ragged", 7, 45, 3, 17, 19, 1),
                                         ("k1_linear", 64, 40, 1, 13, 13, 0), ("mid_52", 16, 32, 3, 52, 52, 1)]:
        pad = 1 if K == 3 else 0
        x = np.zeros((C, H, orclib.w8(W)), dtype=np.float32)
        x[:, :, :W] = rng.standard_normal((C, H, W)).astype(np.float32)
        w = (rng.standard_normal((N, C, K, K)) * np.sqrt(2.0 / (C * K * K))).astype(np.float32)
        b = (rng.standard_normal(N) * 0.1).astype(np.float32)
        wr = synth.reorg_weights(w, C, N, K)
        y = orclib.ref_conv(x, wr, b, C, N, K, 1, W, H, pad, leaky)
        out[f"conv_f32/{name}/x"] = x
        out[f"conv_f32/{name}/w_reorg"] = wr
        out[f"conv_f32/{name}/bias"] = b
        out[f"conv_f32/{name}/params"] = np.array([C, N, K, 1, W, H, pad, leaky], dtype=np.int32)
        out[f"conv_f32/{name}/y"] = y
        fnames.append(name)
        print("conv_f32", name, y.shape)
    out["conv_f32/names"] = np.array(fnames)

    # maxpool, with -32768 inputs and ragged channel count
    for C, W, H in [(5, 14, 10), (32, 26, 26)]:
        x = np.zeros((C, H, orclib.w8(W)), dtype=np.int16)
        x[:, :, :W] = rng.integers(-32768, 32767, (C, H, W))
        x[0, :2, :2] = -32768
        out[f"pool_i16/{C}x{H}x{W}/x"] = x
        out[f"pool_i16/{C}x{H}x{W}/y"] = orclib.ref_maxpool(x, C, W, H)
        xf = np.zeros((C, H, orclib.w8(W)), dtype=np.float32)
        xf[:, :, :W] = rng.standard_normal((C, H, W)).astype(np.float32) * 5
        xf[0, :2, :2] = -2.0e6   # below the reference's -1024*1024 floor: the floor wins
        out[f"pool_f32/{C}x{H}x{W}/x"] = xf
        out[f"pool_f32/{C}x{H}x{W}/y"] = orclib.ref_maxpool(xf, C, W, H)
    np.savez_compressed(os.path.join(HERE, "kat_layers.npz"), **out)
    print("wrote kat_layers.npz")


Q_SETS = {
    # the benchmark configuration: Qw=14, Qb=12, act_q=[14, 9 x 23]
    "std": dict(),
    # varied tables: exercises conv-ordinal Q indexing, the route-28 alignment shift
    # (act_q[21] > act_q[20] -> reorg branch >>2, layer 29 Qa_in = 8) and Qb != 12
    "varq": dict(weight_q=[14, 14, 13, 14, 15, 14, 14, 13, 14, 14, 15, 14, 14, 14, 13, 14, 14, 14, 15, 14, 14, 14, 13],
                 bias_q=[12, 11, 12, 12, 13, 12, 12, 12, 10, 12, 12, 12, 12, 12, 12, 12, 12, 11, 12, 12, 12, 12, 9],
                 act_q=[14, 9, 10, 9, 9, 10, 9, 9, 9, 8, 9, 9, 10, 9, 9, 9, 9, 9, 9, 9, 8, 10, 9, 9]),
}


def run_ref_fullnet(model, frame, int16):
    """Runs the reference's own yolov2_hls_ps in a scratch cwd holding weights/*.bin."""
    work = tempfile.mkdtemp(prefix="ref_fullnet_")
    cwd = os.getcwd()
    try:
        model.write_files(os.path.join(work, "weights"), fp32=not int16, int16=int16)
        os.chdir(work)
        raw_path = os.path.join(work, "raw.txt")
        os.environ["YOLO2_DUMP_REGION_RAW_CPU"] = raw_path
        proc = np.zeros(425 * 169, dtype=np.float32)
        n = orclib.ref(int16).ref_yolov2_hls_ps(REF_CFG.encode(), np.ascontiguousarray(frame), proc)
        assert n == 425 * 169, n
        raw = np.loadtxt(raw_path, dtype=np.float64)
        assert raw.size == 425 * 169
        return raw, proc
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def gen_fullnet():
    out = {}
    for qname, kw in Q_SETS.items():
        model = synth.SynthModel(seed=1, **kw)
        frame = synth.frames(seed=7, count=1)[0]
        raw, proc = run_ref_fullnet(model, frame, int16=True)
        # the raw dump is int16 * 2^-Q printed with %.9g: recover the integers exactly
        # (final Q = last Qa_out; for both sets act_q[23])
        qf = int(model.act_q[23])
        ri = np.rint(raw * (1 << qf)).astype(np.int64)
        # %.9g keeps 9 significant digits: far finer than the 2^-Q grid, so rint() is exact
        assert np.all(np.abs(ri * 2.0 ** -qf - raw) <= 1e-7 * np.maximum(1.0, np.abs(raw))), "raw dump not on the int16 grid"
        assert ri.min() >= -32768 and ri.max() <= 32767
        out[f"i16/{qname}/region_raw_i16"] = ri.astype(np.int16)
        out[f"i16/{qname}/region_proc_f32"] = proc
        out[f"i16/{qname}/final_q"] = np.int32(qf)
        print("fullnet int16", qname, "raw min/max", ri.min(), ri.max(), "nonzero", int((ri != 0).sum()))
        if qname == "std":
            rawf, procf = run_ref_fullnet(model, frame, int16=False)
            out["f32/std/region_raw_f32"] = rawf.astype(np.float32)
            out["f32/std/region_proc_f32"] = procf
            print("fullnet fp32 raw min/max", rawf.min(), rawf.max())
    out["meta/model_seed"] = np.int32(1)
    out["meta/frame_seed"] = np.int32(7)
    np.savez_compressed(os.path.join(HERE, "fullnet.npz"), **out)
    print("wrote fullnet.npz")


def gen_host():
    """Host-logic fixtures from the reference's own src/core code: letterbox of a small synthetic
    image, region activations + boxes + NMS of the int16 full-network region tensor, and the layer
    table its parser builds from config/yolov2.cfg."""
    out = {}
    rng = np.random.default_rng(99)
    rh = orclib.ref_host()
    for (w, h, nw, nh) in [(57, 41, 96, 96), (33, 80, 96, 96), (200, 150, 64, 48), (96, 96, 96, 96)]:
        img = rng.random((3, h, w), dtype=np.float32)
        out[f"letterbox/{w}x{h}to{nw}x{nh}/in"] = (img * 255).round().astype(np.uint8)   # stored as bytes, /255 on use
        img8 = out[f"letterbox/{w}x{h}to{nw}x{nh}/in"].astype(np.float32) / np.float32(255)
        lb = np.zeros((3, nh, nw), dtype=np.float32)
        rh.ref_letterbox(np.ascontiguousarray(img8), w, h, 3, nw, nh, lb)
        out[f"letterbox/{w}x{h}to{nw}x{nh}/out"] = lb.copy()
    full = np.load(os.path.join(HERE, "fullnet.npz"))
    raw = full["i16/std/region_raw_i16"].astype(np.float32) * np.float32(2.0 ** -int(full["i16/std/final_q"]))
    # give a few cells a strong objectness + class so that boxes survive realistic thresholds
    raw = raw.reshape(5, 85, 13, 13).copy()
    for (a, y, x, cls) in [(0, 3, 4, 16), (2, 6, 6, 1), (2, 6, 7, 1), (4, 10, 2, 7), (1, 6, 6, 1)]:
        raw[a, 4, y, x] = 4.0
        raw[a, 5 + cls, y, x] = 9.0
    raw = raw.reshape(-1)
    for (name, imw, imh, thresh, nms) in [("low", 768, 576, 0.05, 0.45), ("std", 768, 576, 0.25, 0.45), ("tall", 300, 500, 0.25, 0.3)]:
        proc = np.zeros(425 * 169, dtype=np.float32)
        rows = np.zeros((845, 85), dtype=np.float32)
        n = rh.ref_detect(REF_CFG.encode(), np.ascontiguousarray(raw), imw, imh, thresh, nms, proc, rows, 845)
        assert n == 845
        out[f"detect/{name}/params"] = np.array([imw, imh, thresh, nms], dtype=np.float64)
        out[f"detect/{name}/rows"] = orclib.canon_rows(rows)
        out["detect/proc"] = proc
        print("detect", name, "kept", len(out[f"detect/{name}/rows"]), "with class prob", int((out[f"detect/{name}/rows"][:, 5:] > 0).any(axis=1).sum()))
    out["detect/raw"] = raw
    whc = np.zeros(3, dtype=np.int32)
    desc = np.zeros(40 * 12, dtype=np.int32)
    n = rh.ref_parse_cfg(REF_CFG.encode(), whc, desc, 40)
    out["cfg/net_whc"] = whc
    out["cfg/desc"] = desc[: n * 12].reshape(n, 12)
    np.savez_compressed(os.path.join(HERE, "host.npz"), **out)
    print("wrote host.npz")


def gen_dog():
    """configs[0] (C1) on the reference's own example image: examples/test_images/dog.jpg decoded by the
    compiled reference (load_image_stb: its vendored stb), letterboxed by its letterbox_image, run through its
    yolov2_hls_ps at both precisions, boxes by its get_network_boxes + do_nms_sort at the CLI defaults
    (NMS 0.45 like src/models/yolov2/yolov2_main.cpp:36-38; threshold 0.05 so that the synthetic weights leave rows to compare).  Stored: decoded RGB bytes, the
    letterboxed frame's checksum, both region tensors, both detection row sets.  Data only."""
    import ctypes
    import hashlib
    path = b"/root/reference/examples/test_images/dog.jpg"
    rh = orclib.ref_host()
    rh.ref_load_image_u8.restype = ctypes.c_long
    w, h = ctypes.c_int(0), ctypes.c_int(0)
    cap = 4096 * 4096 * 3
    rgb = np.zeros(cap, dtype=np.uint8)
    chw = np.zeros(cap, dtype=np.float32)
    n = rh.ref_load_image_u8(path, ctypes.byref(w), ctypes.byref(h), rgb.ctypes.data_as(ctypes.c_void_p),
                             chw.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap))
    assert n == w.value * h.value * 3, n
    W, H = w.value, h.value
    rgb = rgb[:n].reshape(H, W, 3).copy()
    chw = chw[:n].reshape(3, H, W).copy()
    # the stored bytes reproduce the reference's float image exactly
    assert np.array_equal((rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255)).view(np.uint32), chw.view(np.uint32))
    frame = np.zeros((3, 416, 416), dtype=np.float32)
    rh.ref_letterbox(np.ascontiguousarray(chw), W, H, 3, 416, 416, frame)
    out = {"rgb": rgb, "frame_sha256": np.frombuffer(hashlib.sha256(frame.tobytes()).digest(), dtype=np.uint8)}
    model = synth.SynthModel(seed=1)
    raw, proc = run_ref_fullnet(model, frame, int16=True)
    qf = int(model.act_q[23])
    ri = np.rint(raw * (1 << qf)).astype(np.int64)
    assert np.all(np.abs(ri * 2.0 ** -qf - raw) <= 1e-7 * np.maximum(1.0, np.abs(raw)))
    out["i16/region_raw_i16"] = ri.astype(np.int16)
    out["i16/final_q"] = np.int32(qf)
    rawf, procf = run_ref_fullnet(model, frame, int16=False)
    out["f32/region_raw_f32"] = rawf.astype(np.float32)
    for tag, rr in (("i16", ri.astype(np.float32) * np.float32(2.0 ** -qf)), ("f32", rawf.astype(np.float32))):
        p2 = np.zeros(425 * 169, dtype=np.float32)
        rows = np.zeros((845, 85), dtype=np.float32)
        nb = rh.ref_detect(REF_CFG.encode(), np.ascontiguousarray(rr), W, H, 0.05, 0.45, p2, rows, 845)
        assert nb == 845
        out[f"{tag}/detect_rows"] = orclib.canon_rows(rows)
        out[f"{tag}/detect_params"] = np.array([W, H, 0.05, 0.45], dtype=np.float64)
        print("dog", tag, "rows kept", len(out[f"{tag}/detect_rows"]))
    np.savez_compressed(os.path.join(HERE, "dog.npz"), **out)
    print("wrote dog.npz", rgb.shape)


def gen_refapp():
    """Expected region tensor for tests/test_gpu_ref_app.py: the reference's complete Linux application
    (oracle/_ref/yolo2_linux = linux_app/src/*.c linked against libyolo2_hip.so) is run there on a 416x416 PPM,
    for which its loader's letterbox is the identity and the frame is exactly bytes/255.f
    (linux_app/src/yolo2_image_loader.c:35-80,125-243); here the compiled reference's CPU path produces what it
    must print."""
    img = refapp_image()
    frame = (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    out = {}
    for qname, kw in Q_SETS.items():
        model = synth.SynthModel(seed=1, **kw)
        raw, _ = run_ref_fullnet(model, frame, int16=True)
        qf = int(model.act_q[23])
        ri = np.rint(raw * (1 << qf)).astype(np.int64)
        assert np.all(np.abs(ri * 2.0 ** -qf - raw) <= 1e-7 * np.maximum(1.0, np.abs(raw)))
        out[f"{qname}/region_raw_i16"] = ri.astype(np.int16)
        out[f"{qname}/final_q"] = np.int32(qf)
        print("refapp", qname, ri.min(), ri.max())
    np.savez_compressed(os.path.join(HERE, "refapp.npz"), **out)
    print("wrote refapp.npz")


def refapp_image():
    """416x416 RGB bytes from the seeded frame generator (smooth blocks so that a JPEG-free PPM is all we need)."""
    f = synth.frames(seed=31, count=1)[0]                       # float [3][416][416] in [0,1)
    return np.clip(np.floor(f * 256.0), 0, 255).astype(np.uint8).transpose(1, 2, 0).copy()


def _ref_decode(path):
    """stbi_load(path, 3) of the reference's vendored stb, through its load_image_stb (src/core/yolo_image.cpp:167-189)."""
    import ctypes
    rh = orclib.ref_host()
    rh.ref_load_image_u8.restype = ctypes.c_long
    w, h = ctypes.c_int(0), ctypes.c_int(0)
    cap = 2048 * 2048 * 3
    rgb = np.zeros(cap, dtype=np.uint8)
    chw = np.zeros(cap, dtype=np.float32)
    n = rh.ref_load_image_u8(path.encode(), ctypes.byref(w), ctypes.byref(h), rgb.ctypes.data_as(ctypes.c_void_p),
                             chw.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap))
    assert n == w.value * h.value * 3, (path, n)
    return rgb[:n].reshape(h.value, w.value, 3).copy()


def _png_bytes(pix, color, depth, interlace=False, filters=None, palette=None, trns=None, idat_split=0, level=6):
    """A minimal PNG writer for the variants PIL cannot produce (Adam7, forced filter types, 16-bit RGB, 1/2/4-bit grey).
    pix: [h][w][channels] integer samples (unscaled, < 2^depth)."""
    import struct
    import zlib
    h, w, ch = pix.shape

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    def pack_rows(sub):
        hh, ww, _ = sub.shape
        rows = []
        for y in range(hh):
            s = sub[y].reshape(-1).astype(np.int64)
            if depth == 16:
                b = np.stack([(s >> 8) & 255, s & 255], axis=1).reshape(-1).astype(np.uint8)
            elif depth == 8:
                b = s.astype(np.uint8)
            else:
                per = 8 // depth
                pad = (-len(s)) % per
                s = np.concatenate([s, np.zeros(pad, dtype=np.int64)]).reshape(-1, per)
                b = np.zeros(len(s), dtype=np.int64)
                for k in range(per):
                    b |= s[:, k] << (8 - depth * (k + 1))
                b = b.astype(np.uint8)
            rows.append(b)
        return rows

    def filt(rows, bpp, fsel):
        out = bytearray()
        prev = np.zeros(len(rows[0]) if rows else 0, dtype=np.int64)
        for y, r in enumerate(rows):
            cur = r.astype(np.int64)
            f = fsel[y % len(fsel)]
            left = np.concatenate([np.zeros(bpp, dtype=np.int64), cur[:-bpp]]) if len(cur) > bpp else np.zeros(len(cur), dtype=np.int64)
            if len(cur) > bpp:
                ul = np.concatenate([np.zeros(bpp, dtype=np.int64), prev[:-bpp]])
            else:
                ul = np.zeros(len(cur), dtype=np.int64)
            if f == 0:
                d = cur
            elif f == 1:
                d = cur - left
            elif f == 2:
                d = cur - prev
            elif f == 3:
                d = cur - ((left + prev) >> 1)
            else:
                pp = left + prev - ul
                pa, pb, pc = np.abs(pp - left), np.abs(pp - prev), np.abs(pp - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
                d = cur - pred
            out.append(f)
            out += (d & 255).astype(np.uint8).tobytes()
            prev = cur
        return bytes(out)

    bpp = max(1, ch * depth // 8)
    fsel = filters or [0]
    if not interlace:
        raw = filt(pack_rows(pix), bpp, fsel)
    else:
        raw = b""
        for xo, yo, xs, ys in zip((0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)):
            sub = pix[yo::ys, xo::xs]
            if sub.shape[0] and sub.shape[1]:
                raw += filt(pack_rows(sub), bpp, fsel)
    z = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    if idat_split:
        for k in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[k:k + idat_split])
    else:
        out += chunk(b"IDAT", z)
    return out + chunk(b"tEXt", b"Comment\0synthetic") + chunk(b"IEND", b"")


def gen_images():
    """Inputs and expected outputs for the host's own JPEG / PNG decoders (host/y2_codec.cpp): (1) the reference's nine example
    images (examples/test_images/*: the files' bytes are the input - data the reference ships for exactly this purpose), (2) small
    synthetic encodings of every variant the decoder supports, written here with PIL / a minimal PNG writer.  Expected output in
    both cases = the RGB bytes the compiled reference's stb (stbi_load(file, 3) inside load_image_stb) produces: the sha256 of all
    of them and, for the small ones, the bytes themselves.  Data only."""
    import hashlib
    import io
    from PIL import Image
    out = {}
    names = []

    def add(name, data, tmpdir, ext, keep_rgb):
        path = os.path.join(tmpdir, "img" + ext)
        open(path, "wb").write(data)
        rgb = _ref_decode(path)
        out[f"{name}/file"] = np.frombuffer(data, dtype=np.uint8)
        out[f"{name}/shape"] = np.array(rgb.shape[:2], dtype=np.int32)
        out[f"{name}/sha256"] = np.frombuffer(hashlib.sha256(rgb.tobytes()).digest(), dtype=np.uint8)
        if keep_rgb:
            out[f"{name}/rgb"] = rgb
        else:
            out[f"{name}/rgb_rows"] = rgb[::41].copy()     # a sample for diagnostics; equality is asserted on the hash
        names.append(name)
        print(f"  {name:28s} {len(data):8d} B -> {rgb.shape[1]}x{rgb.shape[0]}")

    tmp = tempfile.mkdtemp(prefix="y2img_")
    try:
        seen = {}
        for f in sorted(os.listdir("/root/reference/examples/test_images")):
            data = open(os.path.join("/root/reference/examples/test_images", f), "rb").read()
            key = hashlib.sha256(data).hexdigest()
            if key in seen:     # kite.jpg is a byte-identical copy of image2.jpg
                out[f"ref/{f}/same_as"] = np.frombuffer(seen[key].encode(), dtype=np.uint8)
                continue
            seen[key] = f"ref/{f}"
            add(f"ref/{f}", data, tmp, os.path.splitext(f)[1], keep_rgb=False)
        # ---- synthetic pictures: smooth gradients + hard edges + noise, odd sizes (partial MCUs, 1-pixel-wide chroma rows)
        rng = np.random.default_rng(20240607)

        def picture(h, w):
            yy, xx = np.mgrid[0:h, 0:w]
            base = np.stack([128 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), (xx * 3 + yy * 5) % 256,
                             255 * ((xx // 9 + yy // 6) % 2)], axis=2)
            return np.clip(base + rng.normal(0, 12, size=(h, w, 3)), 0, 255).astype(np.uint8)

        def jpeg(img, mode="RGB", **kw):
            b = io.BytesIO()
            Image.fromarray(img if mode != "L" else img[..., 1]).convert(mode).save(b, "JPEG", **kw)
            return b.getvalue()

        p1, p2, p3 = picture(61, 97), picture(33, 17), picture(1, 1)
        for name, data in [
            ("jpg/base_444", jpeg(p1, quality=90, subsampling=0)),
            ("jpg/base_422", jpeg(p1, quality=75, subsampling=1)),
            ("jpg/base_420", jpeg(p1, quality=60, subsampling=2)),
            ("jpg/base_420_q100", jpeg(p1, quality=100, subsampling=2)),
            ("jpg/base_420_q5", jpeg(p1, quality=5, subsampling=2)),
            ("jpg/base_420_optimize", jpeg(p1, quality=80, subsampling=2, optimize=True)),
            ("jpg/base_420_narrow", jpeg(p2, quality=85, subsampling=2)),
            ("jpg/base_422_narrow", jpeg(p2, quality=85, subsampling=1)),
            ("jpg/base_1x1", jpeg(p3, quality=85, subsampling=2)),
            ("jpg/grey", jpeg(p1, mode="L", quality=80)),
            ("jpg/grey_progressive", jpeg(p1, mode="L", quality=80, progressive=True)),
            ("jpg/prog_444", jpeg(p1, quality=85, subsampling=0, progressive=True)),
            ("jpg/prog_422", jpeg(p1, quality=70, subsampling=1, progressive=True)),
            ("jpg/prog_420", jpeg(p1, quality=92, subsampling=2, progressive=True)),
            ("jpg/prog_420_q20", jpeg(p1, quality=20, subsampling=2, progressive=True)),
            ("jpg/prog_420_narrow", jpeg(p2, quality=85, subsampling=2, progressive=True)),
            ("jpg/prog_1x1", jpeg(p3, quality=85, progressive=True)),
            ("jpg/restart_blocks", jpeg(p1, quality=80, subsampling=2, restart_marker_blocks=3)),
            ("jpg/restart_rows", jpeg(p1, quality=80, subsampling=0, restart_marker_rows=1)),
            ("jpg/prog_restart", jpeg(p1, quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)),
            ("jpg/cmyk", jpeg(p1, mode="CMYK", quality=85)),
            ("jpg/big_420", jpeg(picture(240, 323), quality=88, subsampling=2)),
            ("jpg/big_prog", jpeg(picture(240, 323), quality=88, subsampling=2, progressive=True)),
        ]:
            add(name, data, tmp, ".jpg", keep_rgb=True)

        def png(img, mode, **kw):
            b = io.BytesIO()
            im = Image.fromarray(img)
            if mode == "P":
                im = im.quantize(colors=kw.pop("colors", 200))
            elif mode == "1":
                im = im.convert("L").point(lambda v: 255 if v > 128 else 0).convert("1")
            else:
                im = im.convert(mode)
            im.save(b, "PNG", **kw)
            return b.getvalue()

        rgba = np.concatenate([p1, rng.integers(0, 256, size=p1.shape[:2] + (1,), dtype=np.uint8)], axis=2)
        g16 = (rng.integers(0, 65536, size=(23, 31, 1))).astype(np.int64)
        rgb16 = (rng.integers(0, 65536, size=(23, 31, 3))).astype(np.int64)
        pal = rng.integers(0, 256, size=(16, 3), dtype=np.uint8)
        for name, data in [
            ("png/rgb", png(p1, "RGB")),
            ("png/rgba", png(rgba, "RGBA")),
            ("png/grey", png(p1, "L")),
            ("png/grey_alpha", png(rgba, "LA")),
            ("png/palette", png(p1, "P")),
            ("png/palette_16", png(p1, "P", colors=16)),
            ("png/palette_2", png(p1, "P", colors=2)),
            ("png/bilevel", png(p1, "1")),
            ("png/rgb_level0", png(p1, "RGB", compress_level=0)),
            ("png/rgb_level9", png(picture(120, 160), "RGB", compress_level=9)),
            ("png/rgb16", _png_bytes(rgb16, 2, 16, filters=[0, 1, 2, 3, 4])),
            ("png/grey16_interlaced", _png_bytes(g16, 0, 16, interlace=True, filters=[4, 3])),
            ("png/rgb16_trns", _png_bytes(rgb16, 2, 16, trns=bytes([0, 1, 0, 2, 0, 3]))),
            ("png/grey4", _png_bytes(g16 % 16, 0, 4, filters=[1, 2])),
            ("png/grey2_interlaced", _png_bytes(g16 % 4, 0, 2, interlace=True)),
            ("png/grey1", _png_bytes(g16 % 2, 0, 1, filters=[0, 2])),
            ("png/grey8_trns", _png_bytes(g16 % 256, 0, 8, trns=bytes([0, 77]))),
            ("png/pal4_interlaced", _png_bytes(g16 % 16, 3, 4, interlace=True, palette=pal, trns=bytes(range(0, 160, 10)))),
            ("png/pal1", _png_bytes(g16 % 2, 3, 1, palette=pal[:2])),
            ("png/rgb_interlaced", _png_bytes(p1.astype(np.int64), 2, 8, interlace=True, filters=[0, 1, 2, 3, 4], idat_split=97)),
            ("png/rgba_interlaced_tiny", _png_bytes(rgba[:3, :5].astype(np.int64), 6, 8, interlace=True, filters=[4])),
            ("png/rgb_filters", _png_bytes(p1.astype(np.int64), 2, 8, filters=[4, 3, 2, 1, 0], idat_split=1000)),
            ("png/grey_alpha16", _png_bytes(rng.integers(0, 65536, size=(9, 14, 2)).astype(np.int64), 4, 16, filters=[3])),
            ("png/one_pixel", _png_bytes(np.array([[[200, 100, 50]]], dtype=np.int64), 2, 8)),
        ]:
            add(name, data, tmp, ".png", keep_rgb=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **out)
    print("wrote images.npz:", len(names), "images,", os.path.getsize(os.path.join(HERE, "images.npz")), "bytes")


if __name__ == "__main__":
    if not orclib.have_ref():
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` in the build container")
    what = sys.argv[1:] or ["kats", "fullnet", "host", "dog", "refapp", "images"]
    if "kats" in what:
        gen_kats()
    if "fullnet" in what:
        gen_fullnet()
    if "host" in what:
        gen_host()
    if "dog" in what:
        gen_dog()
    if "refapp" in what:
        gen_refapp()
    if "images" in what:
        gen_images()
