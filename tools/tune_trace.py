#!/usr/bin/env python3
"""Prints set_batch's autotune timings (YOLO2_VERBOSE) for batch 1: every tile shape / split-K shape per layer (GPU box)."""
import os, sys
sys.path.insert(0, "yolo-fpga-accelerator_amd")
from yolo2_amd import hipdrv, synth
model = synth.SynthModel(seed=1)
ctx = hipdrv.Yolo2Hip(0); ctx.load_model(model)
os.environ["YOLO2_VERBOSE"] = "1"
ctx.set_batch(1)
