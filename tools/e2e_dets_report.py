#!/usr/bin/env python3
"""Rate of the images -> detections entry (yolo2_hip_run_images_u8_dets: bytes in, letterbox + network + region / boxes / NMS on the
device, records out) by images per CALL, next to the device-resident rate of the network alone (GPU box).
usage: python3 tools/e2e_dets_report.py [batch=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np
from yolo2_amd import hipdrv, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = [np.clip(np.floor(f * 256.0), 0, 255).astype(np.uint8).transpose(1, 2, 0).copy() for f in synth.frames(31, 16)]
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model); ctx.set_batch(B)
imgs = [base[i % 16] for i in range(4096)]
t0 = time.perf_counter()
hipdrv.run_images_dets(ctx._h, imgs[:2 * B], B, 0.25, 0.45)      # buffers, plan, tables
print(f"first call ({2 * B} images: staging buffers, streams, tail tables): {(time.perf_counter() - t0) * 1e3:.1f} ms")
for n in (B, 4 * B, 16 * B, 64 * B):
    t0 = time.perf_counter()
    reps = max(1, 2048 // n)
    for _ in range(reps):
        out = hipdrv.run_images_dets(ctx._h, imgs[:n], B, 0.25, 0.45)
    dt = (time.perf_counter() - t0) / reps
    print(f"{n:5d} images of 416x416x3 bytes per call (chunks of {B}): {dt * 1e3:8.2f} ms per call = {n / dt:7.0f} frames/s, {np.mean(out['counts']):.1f} records/frame")
ctx.close()
