#!/usr/bin/env python3
"""Camera bytes -> detections on one MI355X box: stage rates of the widened path (GPU box).
  pre  : uint8 images (640x480) cross PCIe, GPU letterbox      (yolo2_hip_run_images_u8_host)
  path : int16 network, bit-exact                              (same call)
  post : dequantise + region + boxes + NMS on host threads     (y2h_postprocess_batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orclib
from yolo2_amd import hipdrv, synth

B = 64
N = 512
rng = np.random.default_rng(1)
base = [rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for _ in range(B)]
imgs = [base[i % B] for i in range(N)]
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model)
ctx.set_batch(B)
region, q = ctx.run_images_host(imgs[:128], B)
t0 = time.perf_counter()
region, q = ctx.run_images_host(imgs, B)
t_all = time.perf_counter() - t0
t_gpu = t_all / (N / B)
print(f"{N} images as bytes in -> region tensors out (H2D of {imgs[0].nbytes/1e6:.2f} MB/image, GPU letterbox, int16 network, D2H; chunks of {B}, 3 streams): {t_all*1e3:.1f} ms = {N/t_all:.0f} FPS")
region = region[:B]

H = orclib.host()
ws = np.full(B, 640, dtype=np.int32); hs = np.full(B, 480, dtype=np.int32)
rows = np.zeros((B, 845, 85), dtype=np.float32); totals = np.zeros(B, dtype=np.int32)
cores = min(len(os.sched_getaffinity(0)), 16)
for th in (1, cores):
    t0 = time.perf_counter()
    for _ in range(3): H.y2h_postprocess_batch(region.ctypes.data, B, q, ws, hs, 0.5, 0.45, th, rows, 845, totals)
    t_post = (time.perf_counter() - t0) / 3
    print(f"region tensors -> detections (thresh 0.5, nms 0.45; synthetic weights: {totals.mean():.0f} boxes/frame kept), {th:2d} host threads: {t_post*1e3:.2f} ms = {B/t_post:.0f} FPS")
print(f"serial sum (no overlap between GPU batch k+1 and host tail of batch k): {B/(t_gpu+t_post):.0f} FPS")
