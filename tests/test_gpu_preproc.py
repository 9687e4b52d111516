"""GPU pre-processing (bytes -> letterboxed float frame) against the reference's own letterbox_image
output (tests/golden/host.npz, generated from the compiled reference) and against the host C++ code,
bit for bit; and the camera-style entry (bytes in, region tensors out) against the float-frame entry."""
import numpy as np
import pytest

import orclib
from yolo2_amd import hipdrv, synth

pytestmark = pytest.mark.gpu


def _host_letterbox(img_hwc, nw, nh):
    h, w = img_hwc.shape[:2]
    chw = np.ascontiguousarray(np.moveaxis(img_hwc if img_hwc.ndim == 3 else np.repeat(img_hwc[:, :, None], 3, 2), 2, 0)
                               .astype(np.float32) / np.float32(255))
    out = np.zeros((3, nh, nw), dtype=np.float32)
    orclib.host().y2h_letterbox(chw, w, h, 3, nw, nh, out)
    return out


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "host.npz"))


@pytest.mark.parametrize("key", ["57x41to96x96", "33x80to96x96", "200x150to64x48", "96x96to96x96"])
def test_letterbox_matches_reference_fixture(key):
    g = _golden()
    nw, nh = (int(v) for v in key.split("to")[1].split("x"))
    img = np.ascontiguousarray(np.moveaxis(g[f"letterbox/{key}/in"], 0, 2))      # CHW bytes -> HWC
    got = hipdrv.letterbox_u8(img, nw, nh)
    want = g[f"letterbox/{key}/out"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("w,h,ch", [(768, 576, 3), (500, 375, 3), (416, 416, 3), (200, 640, 3), (1, 1, 3), (5, 2, 3),
                                    (2, 7, 3), (37, 23, 1), (1920, 1080, 3), (417, 415, 3)])
def test_letterbox_ragged_sizes_match_host_code(w, h, ch):
    rng = np.random.default_rng(w * 10007 + h)
    img = rng.integers(0, 256, size=(h, w, 3) if ch == 3 else (h, w), dtype=np.uint8)
    got = hipdrv.letterbox_u8(img, 416, 416)
    want = _host_letterbox(img, 416, 416)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_letterbox_rejects_bad_geometry():
    L = hipdrv.lib()
    assert L.yolo2_hip_letterbox_u8(1, 0, 5, 3, 1, 416, 416, None) == hipdrv.YOLO2_ERROR
    assert L.yolo2_hip_letterbox_u8(1, 5, 5, 2, 1, 416, 416, None) == hipdrv.YOLO2_ERROR
    assert L.yolo2_hip_letterbox_u8(0, 5, 5, 3, 1, 416, 416, None) == hipdrv.YOLO2_ERROR
    assert L.yolo2_hip_letterbox_u8(1, 100000, 1, 3, 1, 416, 416, None) == hipdrv.YOLO2_ERROR   # fitted height 0


def test_images_entry_equals_float_frame_entry():
    """Bytes of three differently sized images through the camera-style entry == host letterbox +
    float-frame entry (itself pinned to the reference fixtures)."""
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in ((576, 768, 3), (375, 500, 3), (416, 416, 3))]
    model = synth.SynthModel(seed=1)
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    got, q = ctx.run_images_host(imgs)
    frames = np.stack([_host_letterbox(im, 416, 416) for im in imgs])
    want, q2 = ctx.run_batch_host(frames)
    assert q == q2 and np.array_equal(got, want)
    assert not np.array_equal(got[0], got[1])
    # chunked pipeline with a ragged last chunk: 7 images in chunks of 2
    seven = [imgs[i % 3] for i in range(7)]
    got7, _ = ctx.run_images_host(seven, batch=2)
    assert all(np.array_equal(got7[i], want[i % 3]) for i in range(7))
    ctx.close()
