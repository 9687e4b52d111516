#!/bin/bash
# Usage (GPU box): bash tools/c0_abl.sh  -- layer 0 (k_conv0_pool_mfma) with parts compiled out (tools/build_variant.sh c0abl<n> yolo2_fp16 -DY2_C0_ABL=<n>)
P=$PWD/yolo-fpga-accelerator_amd/build
for v in "" 1 2 4 7; do
  L=${v:+$P/lib_c0abl$v.so}
  echo "== Y2_C0_ABL=${v:-0}"; YOLO2_HIP_LIB=$L python3 tools/f16_layers.py 128 10 2>/dev/null | grep "^L 0"
done
