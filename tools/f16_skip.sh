#!/bin/bash
# Usage (GPU box): bash tools/f16_skip.sh [batch=256]
# Prices groups of fp16 launches INSIDE the overlapped two-lane step: the pass is timed with a group left out of the launch table
# (option f16_skip, results wrong by construction) - what a group's removal or fusion could gain at most, as opposed to its solo time.
B=${1:-256}
OUT=$PWD/gpurun_out/f16_skip; mkdir -p "$OUT"
for m in 0 1 2 4 8 16 32 0; do
  YOLO2_F16_SKIP=$m python3 bench.py --precision fp16 --batch $B --steps 20 --no-cpu-baseline > "$OUT/skip_$m.json" 2> "$OUT/skip_$m.err" || exit 1
  python3 -c "import json;d=json.load(open('$OUT/skip_$m.json'));print('f16_skip=%-3s  %8.1f frames/s  %.3f ms/step' % ('$m', d['value'], d['ms_per_step']))"
done | tee "$OUT/summary.txt"
