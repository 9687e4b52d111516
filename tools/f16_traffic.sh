#!/bin/bash
# Usage (GPU box): bash tools/f16_traffic.sh <tag>
# HBM-side traffic per launch of every kernel of the fp16 MFMA pass (batch 256, two lanes of 128 frames), from the PMC counters as
# MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, kernel-trace only.
# The fp16 pass has no autotune and no per-process kernel choice (launch table), so one pass per counter is enough: every launch of a
# layer has the same shape, bytes per launch = counter sum / launches.  tools/f16_traffic.py applies the gfx950 correction (x2 on
# FETCH_SIZE for 16-byte-per-lane reads) and checks it on k_maxpool2_f16, whose byte count is known.
set -e
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/f16_traffic_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for CNT in FETCH_SIZE WRITE_SIZE; do
  D="$OUT/$CNT"; mkdir -p "$D"
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$D" -- python3 bench.py --no-cpu-baseline --no-sub-records --precision fp16 --batch 256 --steps 3 --warmup 1 > "$D/bench.json" 2> "$D/bench.err" || { tail -20 "$D/bench.err"; exit 1; }
  echo "pass $CNT done"
done
python3 tools/f16_traffic.py "$OUT" "$PWD/gpurun_out/f16_traffic_$TAG.json" | tee "$PWD/gpurun_out/f16_traffic_${TAG}_summary.txt"
