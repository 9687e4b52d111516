#!/usr/bin/env python3
"""Workgroup timeline of one halo-kernel launch (diagnostic build: tools/build_variant.sh stamps -DY2_STAMPS).

Usage (GPU box): YOLO2_HIP_LIB=<lib_stamps.so> YOLO2_STAMP_LAYER=<i> YOLO2_F16_NO_LANES=1 python3 tools/stamps.py [batch]
Prints, for the stamped layer, the phases of a workgroup's life in shader cycles (median / p90 over workgroups) and
how the workgroups follow one another on a CU.
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-fpga-accelerator_amd"))
from yolo2_amd import hipdrv, synth

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    layer = int(os.environ["YOLO2_STAMP_LAYER"])
    net = hipdrv.Yolo2Hip(0)
    model = synth.SynthModel(seed=3)
    net.load_weights_fp32(model.weights_f32(), model.bias_f32())
    frames = np.random.default_rng(0).random((batch, 3, 416, 416), dtype=np.float32)
    fd = hipdrv.DevBuf(frames)
    rd = hipdrv.DevBuf(nbytes=batch * 425 * 169 * 4)
    for _ in range(3):
        net.run_batch_fp16_ptr(fd.addr, batch, rd.addr)
        rd.get(np.float32, (4,))   # blocking copy = synchronisation
    lib = hipdrv.lib()
    n = 16384
    buf = np.zeros((n, 8), dtype=np.uint64)
    lib.yolo2_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.yolo2_hip_debug_stamps(buf.ctypes.data, n) == 0
    used = buf[:, 1] != 0
    b = buf[used].astype(np.int64)
    print(f"layer {layer}: {len(b)} workgroups stamped")
    persistent = (b[:, 6] < 4096).all()       # the persistent kernel stores its tile count there, k_conv_f16_halo the HW_ID
    if persistent:
        nt = np.maximum(b[:, 6], 1)
        print(f"  persistent workgroups: {int(nt.min())}..{int(nt.max())} tiles each")
        ph = {"set-up + first fill": b[:, 2] - b[:, 1], "first tile loop": b[:, 3] - b[:, 2], "first tile epilogue": b[:, 4] - b[:, 3],
              "total": b[:, 5] - b[:, 1], "total / tiles": (b[:, 5] - b[:, 1]) // nt}
    else:
        ph = {"set-up": b[:, 2] - b[:, 1], "prologue fill": b[:, 3] - b[:, 2], "main loop": b[:, 4] - b[:, 3], "epilogue": b[:, 5] - b[:, 4],
              "total": b[:, 5] - b[:, 1]}
    for k, v in ph.items():
        print(f"  {k:14s} median {int(np.median(v)):8d}  p10 {int(np.percentile(v, 10)):8d}  p90 {int(np.percentile(v, 90)):8d} cycles")
    real = (b[:, 7] - b[:, 0])   # 100 MHz ticks
    clk = np.median((b[:, 5] - b[:, 1]) / np.maximum(real, 1)) * 100.0
    print(f"  in-kernel clock ~ {clk:.0f} MHz")
    if persistent:
        span = (b[:, 7].max() - b[:, 0].min()) * 10.0 / 1000.0
        print(f"  kernel span {span:.1f} us")
        return
    # succession on a CU: key = (xcc, se, sh, cu)
    hw = b[:, 6] & 0xFFFFFFFF; xcc = (b[:, 6] >> 32) & 0xF
    key = (xcc << 16) | (hw & 0xFF00)
    t0 = b[:, 0].min()
    gaps = []
    for k in np.unique(key):
        rows = b[key == k]
        rows = rows[np.argsort(rows[:, 0])]
        for i in range(1, len(rows)):
            gaps.append(rows[i, 0] - rows[i - 1, 7])
    if gaps:
        g = np.array(gaps) * 10.0   # ns
        print(f"  gap between a workgroup's exit and the next one's entry on the same CU: median {np.median(g):.0f} ns  p90 {np.percentile(g, 90):.0f} ns  ({len(g)} successions on {len(np.unique(key))} CUs)")
    span = (b[:, 7].max() - t0) * 10.0 / 1000.0
    busy = real.sum() * 10.0 / 1000.0 / len(np.unique(key))
    print(f"  kernel span {span:.1f} us; mean per-CU time inside workgroups {busy:.1f} us")

if __name__ == "__main__":
    main()
