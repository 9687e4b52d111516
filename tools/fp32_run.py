#!/usr/bin/env python3
"""Runs the tiled exact fp32 pass a few times (for rocprofv3 --kernel-trace --stats). usage: fp32_run.py [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np
from yolo2_amd import hipdrv, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
frames = hipdrv.DevBuf(np.concatenate([synth.frames(7, 4)] * (B // 4)))
region = hipdrv.DevBuf(nbytes=B * 425 * 169 * 4)
ctx.run_batch_fp32_ptr(frames.addr, B, region.addr, 0)
region.get(np.float32, (1,))
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run_batch_fp32_ptr(frames.addr, B, region.addr, 0)
region.get(np.float32, (1,))
dt = (time.perf_counter() - t0) / steps
print(f"fp32 exact tiled: batch {B}: {dt*1e3:.2f} ms/step = {B/dt:.1f} frames/s")
