#!/usr/bin/env python3
"""Small-batch latency with and without the split-K kernel, and the layers set_batch picked it for (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, synth, net
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model)
convs = net.CONVS
for B in (1, 2, 4, 8):
    frames = torch.from_numpy(synth.frames(7, B)).cuda()
    region = torch.empty((B, 425, 13, 13), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ("0", None):
        if mode is None: os.environ.pop("YOLO2_SPLITK", None)
        else: os.environ["YOLO2_SPLITK"] = mode
        ctx.set_batch(B)
        for _ in range(5): ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), st)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        split = [l.idx for l in convs if ctx.conv_launch_info(l.ord)["pixels_per_lane"] == 0]
        res[mode] = (dt, split, region.cpu().numpy().copy())
    assert np.array_equal(res["0"][2], res[None][2])
    print(f"batch {B}: without split-K {res['0'][0]*1e3:7.3f} ms   tuned {res[None][0]*1e3:7.3f} ms  ({res['0'][0]/res[None][0]:.2f}x)  split-K layers {res[None][1]}", flush=True)

# per-layer view at batch 1
import subprocess
for mode in ("0", None):
    if mode is None: os.environ.pop("YOLO2_SPLITK", None)
    else: os.environ["YOLO2_SPLITK"] = mode
    ctx.set_batch(1)
    frames = torch.from_numpy(synth.frames(7, 1)).cuda()
    region = torch.empty((1, 425, 13, 13), dtype=torch.int16, device="cuda")
    ctx.set_profiling(True)
    for _ in range(20): ctx.run_batch_ptr(frames.data_ptr(), 1, region.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms = ctx.layer_times_ms()
    ctx.set_profiling(False)
    print(f"--- batch 1 per-layer ms, YOLO2_SPLITK={mode}: total {sum(ms):.3f}")
    for l in net.LAYERS:
        info = ctx.conv_launch_info(l.ord) if l.type == net.CONV else None
        tag = "" if info is None else (f"split-K grid=({info['grid_x']},{info['grid_y']})" if info["pixels_per_lane"] == 0 else f"P={info['pixels_per_lane']} grid=({info['grid_x']},{info['grid_y']})")
        print(f"  L{l.idx:2d} {l.type:14s} {ms[l.idx]*1e3:8.1f} us  {tag}")
