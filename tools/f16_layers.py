#!/usr/bin/env python3
"""fp16 path, per layer: hipEvent time of one launch with the chip to itself (lanes off), the kernel the launch table holds,
TFLOP/s, and the whole-pass rate with and without lanes (GPU box).  YOLO2_HIP_LIB selects another build (tools/build_variant.sh).
usage: python tools/f16_layers.py [frames per launch = 128] [steps = 5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, net, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
frames = torch.from_numpy(synth.frames(7, 2 * B)).to(dev)
region = torch.empty((2 * B, 425, 13, 13), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def rate(nb, lanes):
    ctx.set_fp16_lanes(lanes)
    for _ in range(2):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), nb, region.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), nb, region.data_ptr(), st)
    torch.cuda.synchronize()
    return nb * steps / (time.perf_counter() - t0)


solo = rate(B, 1)
kern = ctx.fp16_layer_kernels()
ctx.set_profiling(True)
torch.cuda.synchronize()
for _ in range(steps):
    ctx.run_batch_fp16_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize()
ms = ctx.layer_times_ms()
ctx.set_profiling(False)
tot = 0.0
for l in net.LAYERS:
    if ms[l.idx] <= 0:
        continue
    fl = 2.0 * l.size * l.size * l.c * l.n * l.out_h * l.out_w * B if l.type == net.CONV else 0.0
    tot += ms[l.idx]
    print(f"L{l.idx:2d} {l.type:8s} {l.c:4d}->{l.n:4d} @{l.h:3d}  {ms[l.idx]:7.4f} ms  {fl / ms[l.idx] / 1e9 if fl else 0:7.1f} TFLOP/s  {kern.get(l.idx, '')}")
print(f"sum of layer times {tot:.3f} ms per {B} frames; one launch at a time: {solo:.0f} frames/s; two lanes of {B}: {rate(2 * B, 2):.0f} frames/s")
