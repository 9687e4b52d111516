#!/usr/bin/env python3
"""Per-layer device time of the batched int16 path + issue-cycle efficiency (GPU box).
usage: python tools/layer_report.py [batch] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, net, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CYC = {0: 20, 1: 16, 3: 14, 4: 12}
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model); ctx.set_batch(B)
frames = torch.from_numpy(synth.frames(7, B)).cuda()
region = torch.empty((B, 425, 13, 13), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
ctx.set_profiling(True)
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(steps): ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
ms = ctx.layer_times_ms(); paths = ctx.layer_paths()
print(f"batch {B}: {dt*1e3:.3f} ms/step  {B/dt:.1f} FPS   sum(layer_ms)={ms.sum():.3f}")
tot_ideal = 0
for l in net.LAYERS:
    if l.type == net.CONV:
        info = ctx.conv_launch_info(l.ord)
        st_ = l.size * l.size * ((l.c + 3) // 4) * l.n * l.out_h * l.out_w * B
        cyc = CYC.get(paths[l.ord], 16)
        ideal = st_ / 64 * cyc / 1024 / 2.4e9 * 1e3
        tot_ideal += ideal
        print(f"L{l.idx:2d} conv{l.size} {l.c:4d}->{l.n:4d} @{l.h:3d}  P={info['pixels_per_lane']} blk={info['block']} path={paths[l.ord]} [{ctx.conv_plan(l.ord)}] grid=({info['grid_x']},{info['grid_y']}) lds={info['lds_bytes']:6d}  {ms[l.idx]:7.3f} ms  ideal@2.4GHz {ideal:6.3f}  eff {ideal/ms[l.idx]*100:5.1f}%")
    elif l.type in (net.MAXPOOL, net.REORG, net.REGION):
        print(f"L{l.idx:2d} {l.type:6s}                                                       {ms[l.idx]:7.3f} ms")
print(f"ideal conv total {tot_ideal:.3f} ms -> {B/(tot_ideal*1e-3):.0f} FPS ceiling at 2.4 GHz for these forms")
