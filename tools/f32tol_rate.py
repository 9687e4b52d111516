#!/usr/bin/env python3
"""fp32-tolerance pass: frames/s at a batch size, device-resident, as bench.py's fp32tol_b128 times it (GPU box).  YOLO2_F16_LANES /
YOLO2_F16_NO_LANES select the lane count (latched when the weights are loaded).  usage: python3 tools/f32tol_rate.py [batch = 128] [steps = 8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import torch
from yolo2_amd import hipdrv, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
frames = torch.from_numpy(synth.frames(7, B)).to(dev)
region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"batch {B}, {ctx.num_lanes_f32tol()} lane(s), options '{ctx.options()}': {dt * 1e3:.3f} ms per pass, {B / dt:.0f} frames/s")
