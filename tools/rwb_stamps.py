#!/usr/bin/env python3
"""Group timeline of one tile of k_conv_f16_rwb (diagnostic build: tools/build_variant.sh stamps yolo2_fp16 -DY2_STAMPS): shader cycles
per group of twelve MFMA slots (ideal 192) of wavefront 0 of workgroup 0, fourth tile.
Usage (GPU box): YOLO2_HIP_LIB=<lib_stamps.so> YOLO2_STAMP_LAYER=<4|6> YOLO2_F16_NO_LANES=1 python3 tools/rwb_stamps.py [batch]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-fpga-accelerator_amd"))
from yolo2_amd import hipdrv, synth

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = hipdrv.Yolo2Hip(0)
model = synth.SynthModel(seed=3)
net.load_weights_fp32(model.weights_f32(), model.bias_f32())
frames = np.random.default_rng(0).random((batch, 3, 416, 416), dtype=np.float32)
fd = hipdrv.DevBuf(frames)
rd = hipdrv.DevBuf(nbytes=batch * 425 * 169 * 4)
for _ in range(3):
    net.run_batch_fp16_ptr(fd.addr, batch, rd.addr)
    rd.get(np.float32, (4,))
lib = hipdrv.lib()
buf = np.zeros((8, 8), dtype=np.uint64)
lib.yolo2_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.yolo2_hip_debug_stamps(buf.ctypes.data, 8) == 0
t = buf.reshape(-1)[:40].astype(np.int64)
d = (t[1:] - t[:-1]) & 0xffffffff
print("layer", os.environ.get("YOLO2_STAMP_LAYER"), "kernel", net.fp16_layer_kernels().get(int(os.environ.get("YOLO2_STAMP_LAYER", "6"))))
print("cycles per group (39 groups, ideal 192 each = 12 MFMAs of 16 cycles):")
print(" ".join(str(int(x)) for x in d))
print("tile (group 0 start -> end of group 38):", int((t[39] - t[0]) & 0xffffffff), "cycles; sum of ideal 7488")
