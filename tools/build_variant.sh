#!/bin/bash
# Usage (here, no GPU needed): bash tools/build_variant.sh <name> <tu> [-DMACRO=V ...]
# Compiles ONE translation unit (csrc/<tu>.hip: yolo2_int16, yolo2_fp16, yolo2_fp32, ...) with extra flags and links
# yolo-fpga-accelerator_amd/build/lib_<name>.so from it and the other (unchanged) objects: an alternative build of the
# library for tools/ab.sh / tools/abn.sh (YOLO2_HIP_LIB).  The variant finds the package's plan table (config/plan_gfx950.txt) one
# directory above build/ (csrc/yolo2_plan.hip default_plan_path), so both arms of an A/B run the same kernel selection; ab.sh /
# abn.sh print each arm's conv_plan_source and fail when they differ.
set -e
NAME=$1; TU=$2; shift; shift
P=$(cd "$(dirname "$0")/.." && pwd)/yolo-fpga-accelerator_amd
make -s -C "$P" -j6 "$P/libyolo2_hip.so" >/dev/null
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -ffp-contract=off -fno-slp-vectorize"
[ "$TU" = yolo2_fp16 ] && FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form"     # (the Makefile's TUFLAGS_yolo2_fp16)
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "$P/build/${TU}_$NAME.o" "$P/csrc/$TU.hip"
OBJS=""
for t in yolo2_hip yolo2_driver yolo2_plan yolo2_int16 yolo2_fp16 yolo2_fp32 yolo2_multi yolo2_post; do
  if [ "$t" = "$TU" ]; then OBJS="$OBJS $P/build/${TU}_$NAME.o"; else OBJS="$OBJS $P/build/$t.o"; fi
done
/opt/rocm/bin/hipcc $FLAGS -shared -o "$P/build/lib_$NAME.so" $OBJS -ldl
echo "$P/build/lib_$NAME.so"
