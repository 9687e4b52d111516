#!/usr/bin/env python3
"""Single-frame / small-batch latency of the int16 path and the PCIe-inclusive rate (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, synth
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model)
for B in (1, 2, 4, 8, 16, 64):
    ctx.set_batch(B)
    frames_h = synth.frames(7, B)
    frames = torch.from_numpy(frames_h).cuda()
    region = torch.empty((B, 425, 13, 13), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
    torch.cuda.synchronize()
    n = 20 if B <= 16 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
        torch.cuda.synchronize()          # one sync per batch, like a latency-bound caller
    dt = (time.perf_counter() - t0) / n
    # PCIe-inclusive: pageable host floats in, region tensor out, synchronous (yolo2_hip_run_batch_int16_host)
    t0 = time.perf_counter()
    for _ in range(3): ctx.run_batch_host(frames_h)
    dth = (time.perf_counter() - t0) / 3
    print(f"batch {B:3d}: device-resident {dt*1e3:8.3f} ms/batch  {B/dt:8.1f} FPS   |  host buffers in/out (H2D + run + D2H, sync) {dth*1e3:8.3f} ms  {B/dth:8.1f} FPS")

# streaming entry: 512 host frames through chunks of 64 with overlapped copies
ctx.set_batch(64)
many = np.tile(synth.frames(7, 64), (8, 1, 1, 1))
ctx.run_frames(many[:128], 64)
t0 = time.perf_counter(); ctx.run_frames(many, 64); dts = time.perf_counter() - t0
print(f"streaming 512 host frames, chunks of 64, overlapped H2D/compute/D2H: {512/dts:8.1f} FPS (PCIe-inclusive)")
