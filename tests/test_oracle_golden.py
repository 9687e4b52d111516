"""CPU tests: the restatement in oracle/ against the fixtures in tests/golden/ (generated from
the reference compiled from its own sources, tests/golden/make_golden.py).  Bit-exact."""
import numpy as np
import pytest

import orclib
from yolo2_amd import net, synth

KAT = np.load(orclib.os.path.join(orclib.ROOT, "tests", "golden", "kat_layers.npz"))
FULL = np.load(orclib.os.path.join(orclib.ROOT, "tests", "golden", "fullnet.npz"))


@pytest.mark.parametrize("name", [str(n) for n in KAT["conv_i16/names"]])
def test_conv_i16_kat(name):
    C, N, K, stride, W, H, pad, leaky, Qw, Qai, Qao, Qb = (int(v) for v in KAT[f"conv_i16/{name}/params"])
    x, wr, b, y = (KAT[f"conv_i16/{name}/{k}"] for k in ("x", "w_reorg", "bias", "y"))
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    out = np.zeros((N, OH, orclib.w8(OW)), dtype=np.int16)
    orclib.oracle().orc_conv_i16(x, out, wr, b, C, N, K, stride, W, H, OW, OH, pad, leaky, Qw, Qai, Qao, Qb)
    assert np.array_equal(out, y)


@pytest.mark.parametrize("name", [str(n) for n in KAT["conv_f32/names"]])
def test_conv_f32_kat(name):
    C, N, K, stride, W, H, pad, leaky = (int(v) for v in KAT[f"conv_f32/{name}/params"])
    x, wr, b, y = (KAT[f"conv_f32/{name}/{k}"] for k in ("x", "w_reorg", "bias", "y"))
    out = orclib.conv_f32(x, wr, b, C, N, K, stride, W, H, pad, leaky)
    assert np.array_equal(out.view(np.uint32), y.view(np.uint32))  # bit-exact incl. signed zeros


@pytest.mark.parametrize("shape", ["5x10x14", "32x26x26"])
def test_maxpool_kat(shape):
    C, H, W = (int(v) for v in shape.split("x"))
    for kind in ("i16", "f32"):
        x, y = KAT[f"pool_{kind}/{shape}/x"], KAT[f"pool_{kind}/{shape}/y"]
        assert np.array_equal(orclib.maxpool(x, C, W, H), y)
    # the fp32 floor quirk (core_compute.cpp:291): inputs below -1024*1024 lose to the init value
    assert KAT[f"pool_f32/{shape}/y"][0, 0, 0] == np.float32(-1048576.0)


def test_leaky_exhaustive():
    lib = orclib.oracle()
    xs = np.arange(-32768, 32768, dtype=np.int32)
    got = np.array([lib.orc_leaky_i16(int(v)) for v in xs], dtype=np.int32)
    exp = np.where(xs < 0, -((-xs) // 10), xs)   # truncation toward zero
    assert np.array_equal(got, exp)


def test_quantize_input_edges():
    lib = orclib.oracle()
    x = np.array([0.0, 0.5 / 16384, 1.5 / 16384, -0.5 / 16384, 0.99999, 1.0, 1.9999, 2.0, 5.0, -3.0,
                  2.5 / 16384, -2.5 / 16384], dtype=np.float32)
    out = np.zeros(x.size, dtype=np.int16)
    lib.orc_quantize_input(x, out, x.size, 14)
    #            0  .5->1  1.5->2  -.5->-1  ...                 sat      sat     sat    sat   half-away
    assert list(out) == [0, 1, 2, -1, 16384, 16384, 32766, 32767, 32767, -32768, 3, -3]


def test_reorg_shift():
    rng = np.random.default_rng(3)
    x = rng.integers(-32768, 32767, (64, 26, 32)).astype(np.int16)
    out0 = np.zeros((256, 13, 16), dtype=np.int16)
    out2 = np.zeros_like(out0)
    orclib.oracle().orc_reorg_i16(x, out0, 0)
    orclib.oracle().orc_reorg_i16(x, out2, 2)
    assert np.array_equal(out2, out0 >> 2)          # arithmetic shift, no rounding
    assert np.all(out0[:, :, 13:] == 0)
    # Darknet legacy index map (yolo2_model.cpp:112-129): flat views
    dense = x[:, :, :26].reshape(-1)
    k, j, i = 3, 100, 7
    assert out0.reshape(256 * 13, 16)[(i + 26 * (j + 416 * k)) // 13, (i + 26 * (j + 416 * k)) % 13] == \
        dense[(2 * i + k % 2) + 52 * (2 * j + k // 2)]


def test_int16_file_pad_strip():
    m = synth.SynthModel(seed=5)
    filed = synth.SynthModel._with_layer_pad(m.bias)
    assert filed.size == net.N_BIAS + 1      # only the 425-long last bias is odd (yolo2_model.cpp:216-223)
    dst = np.zeros(net.N_BIAS, dtype=np.int16)
    lens = (orclib.C.c_int * 23)(*net.BIAS_LEN)
    lib = orclib.oracle()
    lib.orc_strip_int16_layer_pad.restype = orclib.C.c_long
    n = lib.orc_strip_int16_layer_pad(filed.ctypes.data_as(orclib.C.c_void_p), orclib.C.c_size_t(filed.size),
                                      lens, 23, dst.ctypes.data_as(orclib.C.c_void_p))
    assert n == net.N_BIAS and np.array_equal(dst, m.bias_i16())


@pytest.mark.slow
@pytest.mark.parametrize("qset", ["std", "varq"])
def test_fullnet_int16_matches_reference(qset):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", orclib.os.path.join(orclib.ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    model = synth.SynthModel(seed=int(FULL["meta/model_seed"]), **mg.Q_SETS[qset])
    frame = synth.frames(int(FULL["meta/frame_seed"]), 1)[0]
    orclib.oracle().orc_set_threads(8)
    ri, rf, q = orclib.forward_i16(model, frame)
    assert q == int(FULL[f"i16/{qset}/final_q"])
    assert np.array_equal(ri, FULL[f"i16/{qset}/region_raw_i16"])
    proc = np.zeros_like(rf)
    orclib.oracle().orc_region_forward(rf, proc)
    assert np.array_equal(proc, FULL[f"i16/{qset}/region_proc_f32"])


@pytest.mark.slow
def test_fullnet_fp32_matches_reference():
    model = synth.SynthModel(seed=int(FULL["meta/model_seed"]))
    frame = synth.frames(int(FULL["meta/frame_seed"]), 1)[0]
    orclib.oracle().orc_set_threads(8)
    rf = orclib.forward_f32(model, frame)
    assert np.array_equal(rf.view(np.uint32), FULL["f32/std/region_raw_f32"].view(np.uint32))
    proc = np.zeros_like(rf)
    orclib.oracle().orc_region_forward(rf, proc)
    assert np.array_equal(proc, FULL["f32/std/region_proc_f32"])


# ------------------------------------------------------------------ configs[0] on the reference's example image

DOG = np.load(orclib.os.path.join(orclib.ROOT, "tests", "golden", "dog.npz"))
REFAPP = np.load(orclib.os.path.join(orclib.ROOT, "tests", "golden", "refapp.npz"))


def dog_frame():
    """dog.jpg as the compiled reference decoded it (bytes), through OUR host letterbox (pinned to the
    reference's letterbox_image by the frame checksum stored with the fixture)."""
    import hashlib
    rgb = DOG["rgb"]
    h, w, _ = rgb.shape
    chw = np.ascontiguousarray(rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    frame = np.zeros((3, 416, 416), dtype=np.float32)
    orclib.host().y2h_letterbox(chw, w, h, 3, 416, 416, frame)
    assert hashlib.sha256(frame.tobytes()).digest() == DOG["frame_sha256"].tobytes(), "host letterbox differs from the reference's"
    return frame


def test_c1_dog_int16_and_fp32_oracle_vs_reference():
    """configs[0]: the oracle reproduces the region tensors the compiled reference computed from dog.jpg,
    int16 bit-exact and fp32 bit-exact, and the host post-processing reproduces its detection rows."""
    frame = dog_frame()
    model = synth.SynthModel(seed=1)
    orclib.oracle().orc_set_threads(8)
    ri, rf, q = orclib.forward_i16(model, frame)
    assert q == int(DOG["i16/final_q"]) and np.array_equal(ri, DOG["i16/region_raw_i16"])
    f32 = orclib.forward_f32(model, frame)
    assert np.array_equal(f32.view(np.uint32), DOG["f32/region_raw_f32"].view(np.uint32))
    for tag, raw in (("i16", ri.astype(np.float32) * np.float32(2.0 ** -q)), ("f32", f32)):
        W, H, thresh, nms = DOG[f"{tag}/detect_params"]
        proc = np.zeros(425 * 169, dtype=np.float32)
        orclib.host().y2h_region_forward(np.ascontiguousarray(raw), proc)
        rows = np.zeros((845, 85), dtype=np.float32)
        orclib.host().y2h_boxes_nms(proc, int(W), int(H), float(thresh), float(nms), rows, 845)
        assert np.array_equal(orclib.canon_rows(rows), DOG[f"{tag}/detect_rows"]), tag


@pytest.mark.parametrize("qset", ["std", "varq"])
def test_refapp_fixture_oracle(qset):
    """The frame the reference's Linux app builds from the 416x416 test image is bytes/255.f; the oracle on it
    equals what the compiled reference computed (the fixture tests/test_gpu_ref_app.py checks the app against)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", orclib.os.path.join(orclib.ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    img = mg.refapp_image()
    frame = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    model = synth.SynthModel(seed=1, **mg.Q_SETS[qset])
    orclib.oracle().orc_set_threads(8)
    ri, _, q = orclib.forward_i16(model, frame)
    assert q == int(REFAPP[f"{qset}/final_q"]) and np.array_equal(ri, REFAPP[f"{qset}/region_raw_i16"])
