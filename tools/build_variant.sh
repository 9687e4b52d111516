#!/bin/bash
# Usage (here, no GPU needed): bash tools/build_variant.sh <name> [-DMACRO=V ...]
# Compiles csrc/yolo2_hip.hip with extra flags and links yolo-fpga-accelerator_amd/build/lib_<name>.so from it and the
# other (unchanged) objects: an alternative build of the library for tools/abn.sh (YOLO2_HIP_LIB).
set -e
NAME=$1; shift
P=$(cd "$(dirname "$0")/.." && pwd)/yolo-fpga-accelerator_amd
make -s -C "$P" -j3 "$P/build/yolo2_multi.o" "$P/build/yolo2_post.o" >/dev/null
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "$P/build/yolo2_hip_$NAME.o" "$P/csrc/yolo2_hip.hip"
/opt/rocm/bin/hipcc $FLAGS -shared -o "$P/build/lib_$NAME.so" "$P/build/yolo2_hip_$NAME.o" "$P/build/yolo2_multi.o" "$P/build/yolo2_post.o" -ldl
echo "$P/build/lib_$NAME.so"
