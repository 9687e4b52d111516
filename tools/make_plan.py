#!/usr/bin/env python3
"""Builds yolo-fpga-accelerator_amd/config/plan_gfx950.txt (GPU box): for every batch the bench, the full-size tests and the CLI
defaults use, the autotuner times its candidates (YOLO2_AUTOTUNE=1 ignores an existing table) and the chosen plan is appended through
YOLO2_PLAN_WRITE.  usage: python3 tools/make_plan.py out.txt [batch ...]
The batches are those of the CONTEXTS that run: a batch of 64 runs as lanes of 21 + 21 + 22 frames, 256 as 2 x 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
batches = [int(b) for b in sys.argv[2:]] or [1, 2, 3, 4, 5, 7, 8, 12, 16, 21, 22, 32, 64, 128, 256]
os.environ["YOLO2_AUTOTUNE"] = "1"
os.environ["YOLO2_PLAN_WRITE"] = out
os.environ["YOLO2_NO_LANES"] = "1"      # plan each size as ONE context; laned batches use the entries of their lane sizes
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
from yolo2_amd import hipdrv, synth
open(out, "w").write("# B L S path P pad splitk pp w16 fuse   (tools/make_plan.py on MI355X; csrc/yolo2_int16.hip: the plan table)\n")
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_model(model)
for b in batches:
    ctx.set_batch(b)
    print("planned batch", b, flush=True)
ctx.close()
