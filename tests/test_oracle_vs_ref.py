"""CPU tests against the reference compiled from its own sources (oracle/_ref).  Skipped where
oracle/_ref has not been built (it is built by `make -C oracle ref` / __graft_entry__.build() in
the build container and shipped as prebuilt files)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import orclib
from yolo2_amd import net, synth

pytestmark = pytest.mark.skipif(not orclib.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("seed", range(12))
def test_random_conv_i16_vs_reference(seed):
    """Random shapes incl. partial Tn/Tm tiles, ragged sizes, all shift directions."""
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 1, 2]))
    C, N = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    W, H = int(rng.integers(K, 40)), int(rng.integers(K, 40))
    pad = int(rng.integers(0, 2)) if K == 3 else 0
    Qw, Qai, Qao, Qb = (int(rng.integers(lo, hi)) for lo, hi in ((8, 16), (6, 15), (6, 15), (4, 15)))
    if seed % 4 == 3:
        Qw = 2   # left shifts
    amp = int(rng.choice([500, 5000, 32768]))
    x = np.zeros((C, H, orclib.w8(W)), dtype=np.int16)
    x[:, :, :W] = rng.integers(-amp, amp, (C, H, W)).clip(-32768, 32767)
    w = rng.integers(-amp, amp, (N, C, K, K)).clip(-32768, 32767).astype(np.int16)
    b = rng.integers(-32768, 32767, N).astype(np.int16)
    wr = synth.reorg_weights(w, C, N, K)
    leaky = int(rng.integers(0, 2))
    a = orclib.conv_i16(x, wr, b, C, N, K, stride, W, H, pad, leaky, Qw, Qai, Qao, Qb, fill=123)
    r = orclib.ref_conv(x, wr, b, C, N, K, stride, W, H, pad, leaky, Qw, Qai, Qao, Qb, fill=123)
    assert np.array_equal(a, r)


@pytest.mark.parametrize("seed", range(4))
def test_random_conv_f32_vs_reference(seed):
    rng = np.random.default_rng(2000 + seed)
    K = int(rng.choice([1, 3]))
    C, N = int(rng.integers(1, 30)), int(rng.integers(1, 50))
    W, H = int(rng.integers(K, 30)), int(rng.integers(K, 30))
    pad = 1 if K == 3 else 0
    x = np.zeros((C, H, orclib.w8(W)), dtype=np.float32)
    x[:, :, :W] = rng.standard_normal((C, H, W))
    w = rng.standard_normal((N, C, K, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    wr = synth.reorg_weights(w, C, N, K)
    a = orclib.conv_f32(x, wr, b, C, N, K, 1, W, H, pad, 1)
    r = orclib.ref_conv(x, wr, b, C, N, K, 1, W, H, pad, 1)
    assert np.array_equal(a.view(np.uint32), r.view(np.uint32))


def test_maxpool_vs_reference():
    rng = np.random.default_rng(5)
    for C, W, H in [(3, 8, 6), (17, 26, 26), (4, 52, 52)]:
        x = np.zeros((C, H, orclib.w8(W)), dtype=np.int16)
        x[:, :, :W] = rng.integers(-32768, 32767, (C, H, W))
        assert np.array_equal(orclib.maxpool(x, C, W, H), orclib.ref_maxpool(x, C, W, H))


def test_reorg_stream_matches_reference_weight_gen():
    """yolo2_amd.synth.reorg_weights against the reference's own yolov2_weight_gen binary
    (src/models/yolov2/yolov2_weight_gen.cpp), int16 and fp32, whole network."""
    tool = os.path.join(orclib.REF_DIR, "yolov2_weight_gen")
    cfg = "/root/reference/config/yolov2.cfg"
    if not (os.path.exists(tool) and os.path.exists(cfg)):
        pytest.skip("weight_gen tool or reference cfg not present")
    m = synth.SynthModel(seed=3)
    with tempfile.TemporaryDirectory() as d:
        m.write_files(os.path.join(d, "weights"), natural=True)
        for prec, src, dst, dt in (("int16", "weight_int16.bin", "o16.bin", np.int16),
                                   ("fp32", "weights.bin", "o32.bin", np.float32)):
            subprocess.run([tool, "--cfg", cfg, "--weights", os.path.join(d, "weights", src),
                            "--out", os.path.join(d, "weights", dst), "--precision", prec],
                           check=True, cwd=d, capture_output=True)
            got = np.fromfile(os.path.join(d, "weights", dst), dtype=dt)
            mine = np.fromfile(os.path.join(d, "weights",
                                            "weights_reorg_int16.bin" if prec == "int16" else "weights_reorg.bin"), dtype=dt)
            assert got.size == mine.size and np.array_equal(got, mine), prec
