#!/bin/bash
# Usage (GPU box): bash tools/rwb_abl.sh -- k_conv_f16_rwb (layers 4 + 5, 6) with parts compiled out
# (tools/build_variant.sh rwbabl<n> yolo2_fp16 -DY2_RWB_ABL=<n>; results wrong by construction, only time matters):
# 1 = no per-group LDS waits, 2 = no staging, 4 = no epilogue work between the MFMAs, 8 = no fragment reads
P=$PWD/yolo-fpga-accelerator_amd/build
for v in "" 1 2 4 8 ""; do
  L=${v:+$P/lib_rwbabl$v.so}
  echo "== Y2_RWB_ABL=${v:-0}"; YOLO2_HIP_LIB=$L python3 tools/f16_layers.py 128 10 2>/dev/null | grep "^L 4\|^L 6"
done
