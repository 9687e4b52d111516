#!/bin/bash
# Usage (GPU box): bash tools/f16_opts.sh "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]   -- the fp16 pass at batch 256 under option sets (one bench run each, same box)
OUT=$PWD/gpurun_out/f16_opts; mkdir -p "$OUT"
i=0
for o in "" "$@" ""; do
  i=$((i+1))
  env $o python3 bench.py --precision fp16 --batch 256 --steps 20 --no-cpu-baseline > "$OUT/o_$i.json" 2> "$OUT/o_$i.err" || { tail -3 "$OUT/o_$i.err"; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/o_$i.json'));print('%-40s %8.1f frames/s  %.3f ms/step  whole_pass %.4f' % ('${o:-default}', d['value'], d['ms_per_step'], d['whole_pass']['frac']))"
done | tee "$OUT/summary.txt"
