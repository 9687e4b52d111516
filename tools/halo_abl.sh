#!/bin/bash
# Usage (GPU box): bash tools/halo_abl.sh -- k_conv_f16_halo (52x52 and smaller 3x3 layers) with parts compiled out
# (tools/build_variant.sh haloabl<n> yolo2_fp16 -DY2_ABL=<n>; results wrong by construction, only time matters):
# 1 = one fragment read per tap, 2 = no weight-tile fills, 4 = a quarter of the MFMAs, 8 = fills from one cache-hot tile,
# 16 = no per-tap address arithmetic, 32 = one barrier per nine taps, 34 = 32 + 2
P=$PWD/yolo-fpga-accelerator_amd/build
for v in "" 1 2 4 8 16 32 34 ""; do
  L=${v:+$P/lib_haloabl$v.so}
  echo "== Y2_ABL=${v:-0}"; YOLO2_HIP_LIB=$L python3 tools/f16_layers.py 128 10 2>/dev/null | grep "^L10\|^L12\|^L18\|^L23"
done
