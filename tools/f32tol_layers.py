#!/usr/bin/env python3
"""Split-fp16 ("fp32 fast") path, per layer: hipEvent time of one launch with the chip to itself (batch below the lane threshold) and the
kernel the launch table holds (GPU box).  usage: python tools/f32tol_layers.py [frames = 60] [steps = 5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, net, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
frames = torch.from_numpy(synth.frames(7, B)).to(dev)
region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize()
ctx.set_profiling(True)
for _ in range(steps):
    ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize()
ms = ctx.layer_times_ms()
tot = 0.0
for l in net.LAYERS:
    if ms[l.idx] <= 0:
        continue
    tot += ms[l.idx]
    print(f"L{l.idx:2d} {l.type:8s} {l.c:4d}->{l.n:4d} @{l.h:3d}  {ms[l.idx]:7.4f} ms  {ctx.f32tol_layer_kernel(l.idx)}")
print(f"sum of layer times {tot:.3f} ms per {B} frames ({ctx.num_lanes_f32tol()} lane(s)): {B / tot * 1e3:.0f} frames/s one launch at a time")
