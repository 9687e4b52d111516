#!/bin/bash
# Usage (GPU box): bash tools/rw_abl.sh -- k_conv_f16_rw (layers 4 + 5, 6) with parts compiled out (tools/build_variant.sh rwabl<n> yolo2_fp16 -DY2_RW_ABL=<n>):
# 1 = halo swizzle key, 2 = no fragment reads, 4 = no epilogue, 8 = no input staging (results wrong by construction, only time matters)
P=$PWD/yolo-fpga-accelerator_amd/build
for v in "" 4 6 14 ""; do
  L=${v:+$P/lib_rwabl$v.so}
  echo "== Y2_RW_ABL=${v:-0}"; YOLO2_HIP_LIB=$L python3 tools/f16_layers.py 128 10 2>/dev/null | grep "^L 4\|^L 6\|^L 8"
done
