"""CPU tests of the repo's own C++ host (yolo-fpga-accelerator_amd/host) against fixtures
generated from the reference's src/core code (tests/golden/make_golden.py gen_host)."""
import os
import subprocess

import numpy as np
import pytest

import orclib
from yolo2_amd import net

HOSTG = np.load(os.path.join(orclib.ROOT, "tests", "golden", "host.npz"))
PKG = os.path.join(orclib.ROOT, "yolo-fpga-accelerator_amd")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-s", "-C", PKG, os.path.join(PKG, "libyolo2_host.so")], check=True)


def test_cfg_parser_matches_reference_parser_table():
    """Our parser on our generated cfg == the reference's parser on its config/yolov2.cfg."""
    whc = np.zeros(3, dtype=np.int32)
    desc = np.zeros(40 * 12, dtype=np.int32)
    anchors = np.zeros(10, dtype=np.float32)
    rp = np.zeros(4, dtype=np.int32)
    n = orclib.host().y2h_parse_cfg(os.path.join(PKG, "config", "yolov2.cfg").encode(), whc, desc, 40, anchors, rp)
    assert n == 32, orclib.host().y2h_last_error()
    want = HOSTG["cfg/desc"]
    got = desc[: n * 12].reshape(n, 12)
    assert list(whc) == list(HOSTG["cfg/net_whc"]) == [416, 416, 3]
    for i in range(32):
        t = want[i, 0]
        cols = list(range(12)) if t == 0 else ([0, 1, 2, 3, 4, 5, 6, 8, 9] if t == 1 else [0, 1, 2, 3, 4, 5, 6] if t in (2, 4) else [0, 4, 5, 6])
        assert list(got[i, cols]) == list(want[i, cols]), (i, got[i], want[i])
    assert np.allclose(anchors, net.ANCHORS) and list(rp) == [80, 4, 5, 1]
    # and the python layer table agrees with both
    for l in net.LAYERS:
        if l.type == net.CONV:
            assert list(got[l.idx, [1, 2, 3, 7, 8, 9, 10, 11]]) == [l.c, l.h, l.w, l.n, l.size, l.stride, l.pad, int(l.leaky)]


def test_cfg_errors():
    whc = np.zeros(3, dtype=np.int32); desc = np.zeros(12, dtype=np.int32)
    a = np.zeros(10, dtype=np.float32); rp = np.zeros(4, dtype=np.int32)
    assert orclib.host().y2h_parse_cfg(b"/nonexistent.cfg", whc, desc, 1, a, rp) == -1
    assert b"Couldn't open file" in orclib.host().y2h_last_error()


@pytest.mark.parametrize("key", sorted({k.split("/")[1] for k in HOSTG.files if k.startswith("letterbox/")}))
def test_letterbox_bit_exact(key):
    src, dst = key.split("to")
    w, h = (int(v) for v in src.split("x"))
    nw, nh = (int(v) for v in dst.split("x"))
    img = HOSTG[f"letterbox/{key}/in"].astype(np.float32) / np.float32(255)
    out = np.zeros((3, nh, nw), dtype=np.float32)
    orclib.host().y2h_letterbox(np.ascontiguousarray(img), w, h, 3, nw, nh, out)
    want = HOSTG[f"letterbox/{key}/out"]
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_region_forward_bit_exact():
    raw = np.ascontiguousarray(HOSTG["detect/raw"])
    proc = np.zeros_like(raw)
    orclib.host().y2h_region_forward(raw, proc)
    assert np.array_equal(proc, HOSTG["detect/proc"])
    p = proc.reshape(5, 85, 169)
    assert np.allclose(p[:, 5:, :].sum(axis=1), 1.0, atol=1e-5)      # softmax over classes
    assert np.all((p[:, [0, 1, 4], :] > 0) & (p[:, [0, 1, 4], :] < 1))   # logistic


@pytest.mark.parametrize("name", ["low", "std", "tall"])
def test_boxes_and_nms_match_reference(name):
    imw, imh, thresh, nms = HOSTG[f"detect/{name}/params"]
    rows = np.zeros((845, 85), dtype=np.float32)
    kept = orclib.host().y2h_boxes_nms(np.ascontiguousarray(HOSTG["detect/proc"]), int(imw), int(imh), float(thresh),
                                       float(nms), rows, 845)
    got = orclib.canon_rows(rows[:kept])
    want = HOSTG[f"detect/{name}/rows"]
    assert got.shape == want.shape
    assert np.array_equal(got[:, :5], want[:, :5])          # boxes + objectness: bit-exact
    assert np.array_equal(got[:, 5:] > 0, want[:, 5:] > 0)  # which class probabilities survive NMS
    assert np.array_equal(got[:, 5:], want[:, 5:])


def test_pnm_loader(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    p = tmp_path / "t.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n# comment\n5 7\n255\n" + img.tobytes())
    whc = np.zeros(3, dtype=np.int32)
    out = np.zeros(3 * 7 * 5, dtype=np.float32)
    assert orclib.host().y2h_load_pnm(str(p).encode(), whc, out, out.size) == 0
    assert list(whc) == [5, 7, 3]
    assert np.array_equal(out.reshape(3, 7, 5), img.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    assert orclib.host().y2h_load_pnm(b"/nonexistent.ppm", whc, out, out.size) == -1


def test_own_weight_gen_matches_reference_tool_and_python(tmp_path):
    """Our C++ yolov2_weight_gen == yolo2_amd.synth.reorg_weights == (where built) the reference's
    own yolov2_weight_gen, byte for byte, int16 (with the per-layer file pad) and fp32."""
    from yolo2_amd import synth
    subprocess.run(["make", "-s", "-C", PKG, os.path.join(PKG, "yolov2_weight_gen")], check=True)
    m = synth.SynthModel(seed=6)
    m.write_files(str(tmp_path / "weights"), natural=True)
    cfg = os.path.join(PKG, "config", "yolov2.cfg")
    for prec, src, ours in (("int16", "weight_int16.bin", "weights_reorg_int16.bin"), ("fp32", "weights.bin", "weights_reorg.bin")):
        out = tmp_path / f"mine_{prec}.bin"
        r = subprocess.run([os.path.join(PKG, "yolov2_weight_gen"), "--cfg", cfg, "--weights", str(tmp_path / "weights" / src),
                            "--out", str(out), "--precision", prec], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == open(tmp_path / "weights" / ours, "rb").read(), prec
        ref_tool = os.path.join(orclib.REF_DIR, "yolov2_weight_gen")
        if os.path.exists(ref_tool) and os.path.exists("/root/reference/config/yolov2.cfg"):
            ref_out = tmp_path / f"ref_{prec}.bin"
            subprocess.run([ref_tool, "--cfg", "/root/reference/config/yolov2.cfg", "--weights", str(tmp_path / "weights" / src),
                            "--out", str(ref_out), "--precision", prec], check=True, capture_output=True, cwd=str(tmp_path))
            assert open(out, "rb").read() == open(ref_out, "rb").read(), prec
    r = subprocess.run([os.path.join(PKG, "yolov2_weight_gen"), "--weights", "/nonexistent.bin", "--cfg", cfg], capture_output=True, text=True)
    assert r.returncode == 1 and "Fatal error" in r.stderr


@pytest.mark.parametrize("threads", [1, 4])
def test_postprocess_batch_equals_single_frame_calls(threads):
    """The threaded batch tail (int16 region tensors -> detections) gives exactly what the
    single-frame functions (pinned to the reference above) give frame by frame."""
    H = orclib.host()
    full = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullnet.npz"))
    base = full["i16/std/region_raw_i16"].reshape(-1)
    q = int(full["i16/std/final_q"])
    rng = np.random.default_rng(3)
    B = 6
    region = np.stack([base if f == 0 else np.roll(base, 1009 * f) + rng.integers(-3, 4, base.size) for f in range(B)]).astype(np.int16)
    ws = np.array([768, 416, 500, 640, 1, 1920], dtype=np.int32)
    hs = np.array([576, 416, 375, 480, 1, 1080], dtype=np.int32)
    thresh, nms = 0.3, 0.45
    rows = np.zeros((B, 845, 85), dtype=np.float32)
    totals = np.zeros(B, dtype=np.int32)
    assert H.y2h_postprocess_batch(region.ctypes.data, B, q, ws, hs, thresh, nms, threads, rows, 845, totals) == 0
    for f in range(B):
        raw = np.ascontiguousarray(region[f].astype(np.float32) * np.float32(2.0 ** -q))
        proc = np.empty_like(raw)
        H.y2h_region_forward(raw, proc)
        one = np.zeros((845, 85), dtype=np.float32)
        n = H.y2h_boxes_nms(proc, int(ws[f]), int(hs[f]), thresh, nms, one, 845)
        assert n == totals[f] and n > 0
        assert np.array_equal(rows[f, :n].view(np.uint32), one[:n].view(np.uint32))
