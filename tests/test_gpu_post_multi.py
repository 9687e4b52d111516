"""GPU tests of the rows SURVEY.md 8(f).1 / 8(e) widened this round: region + boxes + NMS on the GPU
(yolo2_hip_postprocess_*), and more than one device behind the C ABI (yolo2_hip_multi_*, the RCCL rank API).
Checkers: the reference-generated fixtures (tests/golden/host.npz, dog.npz) and the host-threaded restatement
(libyolo2_host.so), itself pinned to the reference on CPU (tests/test_host_side.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import orclib
from yolo2_amd import hipdrv, net, synth

pytestmark = pytest.mark.gpu
ROOT = orclib.ROOT
HOSTFX = np.load(os.path.join(ROOT, "tests", "golden", "host.npz"))
FULL = np.load(os.path.join(ROOT, "tests", "golden", "fullnet.npz"))
DOG = np.load(os.path.join(ROOT, "tests", "golden", "dog.npz"))


def _host_rows(region_i16, q, ws, hs, thresh, nms, threads=8):
    B = region_i16.shape[0]
    rows = np.zeros((B, 845, 85), dtype=np.float32)
    totals = np.zeros(B, dtype=np.int32)
    r = np.ascontiguousarray(region_i16.reshape(B, -1))
    assert orclib.host().y2h_postprocess_batch(r.ctypes.data, B, q, np.ascontiguousarray(ws, dtype=np.int32),
                                               np.ascontiguousarray(hs, dtype=np.int32), thresh, nms, threads, rows, 845, totals) == 0
    return rows, totals


@pytest.mark.parametrize("name", ["low", "std", "tall"])
def test_gpu_postprocess_reproduces_reference_detection_rows(name):
    """The region tensor behind tests/golden/host.npz detect/* (the int16 full-network tensor with a few strong cells
    injected: every value is still int16 x 2^-9) through yolo2_hip_postprocess_int16: l.output of forward_region_layer
    and the detection rows after do_nms_sort equal the compiled reference's, bit for bit."""
    raw = HOSTFX["detect/raw"]
    q = 9
    ri = np.rint(raw * (1 << q)).astype(np.int64)
    assert np.array_equal(ri.astype(np.float32) * np.float32(2.0 ** -q), raw) and ri.min() >= -32768 and ri.max() <= 32767
    imw, imh, thresh, nms = HOSTFX[f"detect/{name}/params"]
    ctx = hipdrv.Yolo2Hip(0)
    buf = hipdrv.DevBuf(ri.astype(np.int16))
    out = hipdrv.postprocess(ctx, buf.addr, 1, [int(imw)], [int(imh)], float(thresh), float(nms), final_q=q, cap=2048,
                             want_rows=True, want_proc=True)
    assert np.array_equal(out["proc"][0].view(np.uint32), HOSTFX["detect/proc"].view(np.uint32))
    want = HOSTFX[f"detect/{name}/rows"]
    got = orclib.canon_rows(out["rows"][0])
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the compact records are exactly the (row, class) pairs with prob > 0, in array order
    rows = out["rows"][0]
    pairs = [(i, j) for i in range(int(out["totals"][0])) for j in range(80) if rows[i, 5 + j] > 0]
    recs = out["dets"][0]
    assert int(out["counts"][0]) == len(pairs) == len(recs)
    assert [(int(r["det"]), int(r["cls"])) for r in recs] == pairs
    assert all(r["prob"] == rows[r["det"], 5 + r["cls"]] and r["x"] == rows[r["det"], 0] and r["h"] == rows[r["det"], 3] for r in recs)
    # float entry on the same tensor (device exp instead of the host-filled tables): same candidates, values within 1e-6
    fbuf = hipdrv.DevBuf(np.ascontiguousarray(raw, dtype=np.float32))
    outf = hipdrv.postprocess(ctx, fbuf.addr, 1, [int(imw)], [int(imh)], float(thresh), float(nms), cap=2048, want_rows=True, want_proc=True)
    assert np.allclose(outf["proc"][0], HOSTFX["detect/proc"], rtol=2e-6, atol=1e-9)
    gf = orclib.canon_rows(outf["rows"][0])
    assert gf.shape == want.shape and np.allclose(gf, want, rtol=1e-5, atol=1e-7)
    buf.free(); fbuf.free(); ctx.close()


def test_gpu_postprocess_batch_equals_host_rows_in_order():
    """A batch straight from the int16 network (region tensor stays in HBM) against the host-threaded tail
    (postprocess_batch, pinned to the reference on CPU): the whole dets[] array of every frame, element for element and
    in the same ORDER - the 80 successive stable sorts included - at a low threshold that keeps hundreds of candidates
    and many classes busy, ragged image sizes, with and without NMS."""
    model = synth.SynthModel(seed=1)
    B = 12
    frames = np.concatenate([synth.frames(7, 1), synth.frames(321, B - 1)])
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    fd = hipdrv.DevBuf(frames)                                   # (device memory through the library: no torch in this process)
    rd = hipdrv.DevBuf(nbytes=B * 425 * 169 * 2)
    q = ctx.run_batch_ptr(fd.addr, B, rd.addr, 0)
    region = rd.get(np.int16, (B, 425, 13, 13))                 # (a blocking copy on the null stream: orders behind the pass)
    assert np.array_equal(region[0].reshape(-1), FULL["i16/std/region_raw_i16"])
    ws = [768, 416, 500, 640, 1, 1920, 333, 416, 100, 4000, 640, 77]
    hs = [576, 416, 375, 480, 1, 1080, 999, 415, 100, 3000, 360, 78]
    for thresh, nms in ((0.004, 0.45), (0.02, 0.3), (0.24, 0.45), (0.02, 0.0)):
        out = hipdrv.postprocess(ctx, rd.addr, B, ws, hs, thresh, nms, final_q=q, cap=4096, want_rows=True)
        rows, totals = _host_rows(region, q, ws, hs, thresh, nms)
        if nms <= 0:      # (without NMS the host keeps all 845 slots; the candidates in front are what counts)
            totals = out["totals"]
        assert np.array_equal(out["totals"], totals), (thresh, nms)
        assert totals.max() > (300 if thresh < 0.01 else 0)
        for f in range(B):
            n = int(totals[f])
            assert np.array_equal(out["rows"][f, :n].view(np.uint32), rows[f, :n].view(np.uint32)), (thresh, nms, f)
            assert not out["rows"][f, n:].any()
    ctx.close()


def test_gpu_postprocess_dog_and_throughput():
    """configs[0]'s boxes from the GPU tail: dog.jpg's int16 region tensor -> the reference's detection rows
    (tests/golden/dog.npz); and the rate on a batch-256 tensor resident in HBM (the fp16 path produces ~28 k frames/s:
    the tail must not be the bottleneck)."""
    import time
    q = int(DOG["i16/final_q"])
    W, H, thresh, nms = DOG["i16/detect_params"]
    ctx = hipdrv.Yolo2Hip(0)
    one = hipdrv.DevBuf(DOG["i16/region_raw_i16"].reshape(1, -1).copy())
    out = hipdrv.postprocess(ctx, one.addr, 1, [int(W)], [int(H)], float(thresh), float(nms), final_q=q, cap=4096, want_rows=True)
    assert np.array_equal(orclib.canon_rows(out["rows"][0]).view(np.uint32), DOG["i16/detect_rows"].view(np.uint32))
    B = 256
    rng = np.random.default_rng(0)
    base = DOG["i16/region_raw_i16"].astype(np.int32)
    batch = np.stack([np.clip(np.roll(base, 173 * f) + rng.integers(-2, 3, base.size), -32768, 32767) for f in range(B)]).astype(np.int16)
    rd = hipdrv.DevBuf(batch)
    ws, hs = [768] * B, [576] * B
    hipdrv.postprocess(ctx, rd.addr, B, ws, hs, 0.24, 0.45, final_q=q, cap=128)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        res = hipdrv.postprocess(ctx, rd.addr, B, ws, hs, 0.24, 0.45, final_q=q, cap=128)
    dt = (time.perf_counter() - t0) / reps
    fps = B / dt
    print(f"GPU region+boxes+NMS: {fps:.0f} frames/s at batch {B} (thresh 0.24)")
    rows, totals = _host_rows(batch[:8], q, ws[:8], hs[:8], 0.24, 0.45)
    full = hipdrv.postprocess(ctx, rd.addr, 8, ws[:8], hs[:8], 0.24, 0.45, final_q=q, cap=128, want_rows=True)
    for f in range(8):
        assert np.array_equal(full["rows"][f, :totals[f]].view(np.uint32), rows[f, :totals[f]].view(np.uint32))
    assert fps >= 30000, fps
    ctx.close()


# ------------------------------------------------------------------ more than one device behind the C ABI

def test_multi_context_sharding_on_one_gpu():
    """yolo2_hip_multi_* with the device list [0] and [0, 0, 0] (one GPU listed three times: RCCL refuses duplicate
    devices, so the blobs travel by device-to-device copies; the sharding, the per-device host threads and the streaming
    entry are the real ones): 13 frames in shards of 5 + 4 + 4, chunks of 2 -> identical to one context."""
    model = synth.SynthModel(seed=1)
    frames = np.concatenate([synth.frames(7, 1), synth.frames(55, 12)])
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    want, q0 = ctx.run_batch_host(frames)
    ctx.close()
    assert np.array_equal(want[0].reshape(-1), FULL["i16/std/region_raw_i16"])
    for devs in ([0], [0, 0, 0]):
        m = hipdrv.Yolo2HipMulti(devs)
        assert not m.uses_rccl()
        m.load_model(model)
        got, q = m.run_frames(frames, batch_per_device=2)
        assert q == q0 and np.array_equal(got, want), devs
        m.close()
    with pytest.raises(hipdrv.Yolo2HipError, match="no HIP device"):
        hipdrv.Yolo2HipMulti([0, 99])
    assert [hipdrv.shard_range(13, r, 3) for r in range(3)] == [(0, 5), (5, 9), (9, 13)]


def test_multi_images_entry_and_rank_api_with_rccl():
    """(a) byte images through the multi entry == the single-context entry; (b) the one-process-per-device API on a
    1-rank communicator: ncclGetUniqueId, ncclCommInitRank, ncclBroadcast from librccl.so (the degenerate world a 1-GPU
    box allows), then the int16 and fp32 weight sets arrive through the broadcast loader and give the fixture tensors."""
    model = synth.SynthModel(seed=1)
    rgb = DOG["rgb"]
    imgs = [rgb, rgb[::2, ::2].copy(), rgb[:, ::-1].copy(), rgb[100:400, 50:700].copy(), rgb.transpose(1, 0, 2).copy()]
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    want, _ = ctx.run_images_host(imgs, batch=2)
    assert np.array_equal(want[0].reshape(-1), DOG["i16/region_raw_i16"])
    ctx.close()
    m = hipdrv.Yolo2HipMulti([0, 0])
    m.load_model(model)
    got, _ = m.run_images(imgs, batch_per_device=2)
    assert np.array_equal(got, want)
    m.close()
    # rank API, world size 1
    ctx = hipdrv.Yolo2Hip(0)
    with pytest.raises(hipdrv.Yolo2HipError, match="rccl_init_rank"):
        ctx.load_model_bcast(model, root=0)
    uid = hipdrv.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.rccl_init_rank(uid, 1, 0)
    ctx.load_model_bcast(model, root=0)
    frame = synth.frames(7, 1)
    region, q = ctx.run_batch_host(frame)
    assert q == 9 and np.array_equal(region[0].reshape(-1), FULL["i16/std/region_raw_i16"])
    info = ctx.rccl_info()                      # what the communicator itself reports (bench.py's "rccl" object)
    assert (info["nranks"], info["rank"], info["device"], info["bcasts"]) == (1, 0, 0, 1)
    assert info["bytes"] == 2 * (net.N_WEIGHTS + net.N_BIAS) + 4 * (3 + 3 * 64) and info["bcast_ms"] > 0
    assert "rccl" in info["lib_path"] and info["version"] >= 20000
    ctx.load_model_fp32_bcast(model, root=0)
    f32 = ctx.run_frame_fp32_host(frame[0])
    assert np.array_equal(f32.reshape(-1).view(np.uint32), FULL["f32/std/region_raw_f32"].view(np.uint32))
    assert ctx.rccl_info()["bcasts"] == 2
    # a root-side failure is carried INTO the collective (status agreement), not returned in front of it: the call fails with
    # the root's own message, the communicator stays usable, the model loaded before is still in force
    w = np.ascontiguousarray(model.weights_i16()[:1000])
    b = np.ascontiguousarray(model.bias_i16())
    import ctypes as C
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (model.weight_q, model.bias_q, model.act_q))
    rc = hipdrv.lib().yolo2_hip_load_weights_int16_bcast(ctx._h, vp(w), w.size, vp(b), b.size, vp(wq), wq.size, vp(bq), bq.size, vp(aq), aq.size, 0)
    assert rc == hipdrv.YOLO2_ERROR and b"too small" in hipdrv.lib().yolo2_hip_last_error()
    assert ctx.rccl_info()["bcasts"] == 2       # no broadcast ran
    region2, _ = ctx.run_batch_host(frame)
    assert np.array_equal(region2, region)
    ctx.close()                                 # destroy leaves the communicator (no explicit rccl_finalize) ...
    ctx = hipdrv.Yolo2Hip(0)                    # ... so a new context, wherever it is allocated, can join a new one
    ctx.rccl_init_rank(hipdrv.rccl_unique_id(), 1, 0)
    assert ctx.rccl_info()["bcasts"] == 0
    ctx.close()


def test_postprocess_rejects_a_region_tensor_it_cannot_reach():
    """yolo2_hip_postprocess_* runs on the context's device; an address that is not device-accessible memory is refused
    (on a multi-GPU box: also a tensor that lives in another GPU's HBM - allocate with yolo2_hip_alloc_on)."""
    ctx = hipdrv.Yolo2Hip(0)
    host = np.zeros(net.REGION_ELEMS if hasattr(net, "REGION_ELEMS") else 425 * 169, dtype=np.int16)
    with pytest.raises(hipdrv.Yolo2HipError, match="not device-accessible"):
        hipdrv.postprocess(ctx, host.ctypes.data, 1, [640], [480], 0.25, 0.45, final_q=9)
    import ctypes as C
    a = C.c_uint64(0)
    hipdrv.check(hipdrv.lib().yolo2_hip_alloc_on(ctx._h, host.nbytes, C.byref(a)), "alloc_on")
    hipdrv.check(hipdrv.lib().yolo2_hip_memset(a, 0, host.nbytes), "memset")
    out = hipdrv.postprocess(ctx, a.value, 1, [640], [480], 0.6, 0.45, final_q=9)
    assert int(out["counts"][0]) == 0           # an all-zero tensor: objectness 0.5 everywhere, below the threshold
    hipdrv.lib().yolo2_hip_free(a)
    assert hipdrv.lib().yolo2_hip_ctx_device(ctx._h) == 0
    ctx.close()


def test_c5_batch2048_sharded_over_eight_contexts_on_one_gpu():
    """configs[4] rehearsed on the one GPU of this box: 2048 frames frame-sharded over EIGHT contexts (device list
    [0]*8, so RCCL is replaced by device-to-device copies; the sharding, the per-device host threads, the 256-frame
    shards with their two 128-frame lanes and the streaming entry are the real ones).  Frames of period 8: every frame of
    every shard equals the period's reference result (frame 0 = the compiled reference's fixture), i.e. the result does
    not depend on which shard or which position a frame lands in."""
    model = synth.SynthModel(seed=1)
    base = np.concatenate([synth.frames(7, 1), synth.frames(4000, 7)])
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    want, q0 = ctx.run_batch_host(base)
    ctx.close()
    assert np.array_equal(want[0].reshape(-1), FULL["i16/std/region_raw_i16"])
    frames = np.broadcast_to(base[None], (256, 8, 3, 416, 416)).reshape(2048, 3, 416, 416)      # 4.25 GB of floats
    frames = np.ascontiguousarray(frames)
    m = hipdrv.Yolo2HipMulti([0] * 8)
    m.load_model(model)
    assert [hipdrv.shard_range(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    got, q = m.run_frames(frames, batch_per_device=256)
    m.close()
    assert q == q0 and got.shape == (2048, 425, 13, 13)
    g = got.reshape(256, 8, -1)
    assert np.array_equal(g, np.broadcast_to(want.reshape(1, 8, -1), g.shape))


def test_images_to_detections_entry_keeps_the_region_tensor_on_the_device():
    """yolo2_hip_run_images_u8_dets (round 3): bytes in -> letterbox + network + region / boxes / NMS + record compaction on the device,
    chunks overlapped on three streams; its records must be exactly those of the two-step route (region tensors to the host, uploaded
    again, yolo2_hip_postprocess_int16) - all classes, and the best-class mode the streaming CLI uses; through one context (chunks
    of 2 and 3 with a ragged last chunk) and through the multi entry (two shards on device 0)."""
    model = synth.SynthModel(seed=1)
    rgb = DOG["rgb"]
    imgs = [rgb, rgb[::2, ::2].copy(), rgb[:, ::-1].copy(), rgb[100:400, 50:700].copy(), rgb.transpose(1, 0, 2).copy(), rgb[::3, ::2].copy(),
            rgb[:300].copy()]
    thresh, nms = 0.05, 0.45
    ctx = hipdrv.Yolo2Hip(0)
    ctx.load_model(model)
    region, q = ctx.run_images_host(imgs, batch=3)
    buf = hipdrv.DevBuf(region)
    ws, hs = [im.shape[1] for im in imgs], [im.shape[0] for im in imgs]
    want = hipdrv.postprocess(ctx, buf.addr, len(imgs), ws, hs, thresh, nms, final_q=q, cap=4096)
    buf.free()
    assert min(int(c) for c in want["counts"]) > 5
    for batch in (2, 3):
        got = hipdrv.run_images_dets(ctx._h, imgs, batch, thresh, nms, cap=4096, best_class=False)
        assert got["final_q"] == q and np.array_equal(got["counts"], want["counts"])
        for f in range(len(imgs)):
            assert np.array_equal(got["dets"][f], want["dets"][f]), (batch, f)
    # best-class mode: one record per detection, its best class, first among equals
    best = hipdrv.run_images_dets(ctx._h, imgs, 3, thresh, nms, cap=845, best_class=True)
    for f in range(len(imgs)):
        w = want["dets"][f]
        exp = []
        for det in np.unique(w["det"]):       # records are grouped by detection, classes ascending
            rows = w[w["det"] == det]
            exp.append(rows[np.argmax(rows["prob"])])
        exp = np.array(exp, dtype=w.dtype)
        assert int(best["counts"][f]) == len(exp) <= 845
        assert np.array_equal(best["dets"][f], exp), f
    ctx.close()
    m = hipdrv.Yolo2HipMulti([0, 0])
    m.load_model(model)
    gm = hipdrv.run_images_dets(m._m, imgs, 2, thresh, nms, cap=845, best_class=True, multi=True)
    m.close()
    assert np.array_equal(gm["counts"], best["counts"])
    for f in range(len(imgs)):
        assert np.array_equal(gm["dets"][f], best["dets"][f]), f
