#!/usr/bin/env python3
"""End-to-end rate of the streaming frontend (GPU box): N synthetic 416x416 PPM files on disk -> yolov2_detect --input-list ->
JSONL, i.e. file I/O + decode + PCIe + letterbox + network + region/boxes/NMS + output, against the device-resident rate of
bench.py.  usage: python3 tools/e2e_stream.py [n_files=2048] [batch=64] [chunk_batches=4] [post=gpu]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "yolo-fpga-accelerator_amd")
sys.path.insert(0, PKG)
import numpy as np
from yolo2_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cb = sys.argv[3] if len(sys.argv) > 3 else "4"
post = sys.argv[4] if len(sys.argv) > 4 else "gpu"
tmp = tempfile.mkdtemp(prefix="y2e2e_")
model = synth.SynthModel(seed=1)
model.write_files(os.path.join(tmp, "weights"), fp32=False, int16=True)
base = [np.clip(np.floor(f * 256.0), 0, 255).astype(np.uint8).transpose(1, 2, 0).copy() for f in synth.frames(31, 16)]
paths = []
for k in range(n):
    p = os.path.join(tmp, f"f{k:05d}.ppm")
    with open(p, "wb") as f:
        f.write(b"P6\n416 416\n255\n" + base[k % 16].tobytes())
    paths.append(p)
lst = os.path.join(tmp, "list.txt")
open(lst, "w").write("\n".join(paths) + "\n")
cmd = [os.path.join(PKG, "yolov2_detect"), "--cfg", os.path.join(PKG, "config", "yolov2.cfg"), "--names", os.path.join(PKG, "config", "coco.names"),
       "--weights", os.path.join(tmp, "weights"), "--input-list", lst, "--batch", str(batch), "--chunk-batches", cb, "--post", post,
       "--thresh", "0.25", "--jsonl", os.path.join(tmp, "out.jsonl")]
for rep in range(2):      # the second run finds the files in the page cache and the plan in place
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, YOLO2_NO_DUMP="1"))
    dt = time.time() - t0
    if r.returncode != 0:
        sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
    tail = [l for l in r.stdout.splitlines() if "Streaming inference completed" in l][-1]
    print(f"run {rep}: n={n} batch={batch} chunk_batches={cb} post={post}: process wall {dt:.2f} s ({n / dt:.0f} frames/s incl. start-up, weight load and planning)")
    print("   ", tail)
    import re
    per = [float(m.group(1)) for m in re.finditer(r"inference time: ([0-9.]+) ms", r.stdout)]
    chunks = [per[0]] + [b for a, b in zip(per, per[1:]) if b != a]
    print("    per-frame share of each accelerator call, ms:", " ".join(f"{x:.3f}" for x in chunks))
nrec = sum(1 for _ in open(os.path.join(tmp, "out.jsonl")))
assert nrec == n, (nrec, n)
import shutil; shutil.rmtree(tmp, ignore_errors=True)
