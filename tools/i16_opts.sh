#!/bin/bash
# Usage (GPU box): bash tools/i16_opts.sh <rounds> "<ENV=VAL ...>" [...]   -- the int16 headline (batch 64) under option sets, alternating runs on one box
N=$1; shift
OUT=$PWD/gpurun_out/i16_opts; mkdir -p "$OUT"
for r in $(seq 1 $N); do
  i=0
  for o in "" "$@"; do
    i=$((i+1))
    env $o python3 bench.py --steps 20 --no-cpu-baseline --no-sub-records > "$OUT/o_${i}_$r.json" 2> "$OUT/o_${i}_$r.err" || { tail -3 "$OUT/o_${i}_$r.err"; exit 1; }
    python3 -c "import json;d=json.load(open('$OUT/o_${i}_$r.json'));print('round $r  %-32s %8.1f frames/s  valu %.4f  %s' % ('${o:-default}', d['value'], d['valu_roofline']['frac'], d['config']['conv_plan_source']))"
  done
done | tee "$OUT/summary.txt"
