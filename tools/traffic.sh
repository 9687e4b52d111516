#!/bin/bash
# Usage (GPU box): bash tools/traffic.sh <tag> [bench args]
# HBM-side traffic per kernel launch from the PMC counters, as MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only).
# Each counter is collected at --steps 2 and --steps 6 (no warm-up); the difference / 4 is one
# step's traffic with set_batch's autotune launches cancelled out.  gfx950 correction: FETCH_SIZE
# tallies 128-B requests at 64 B -> x2; checked here on k_maxpool2, whose byte count is known
# exactly (same 8-byte-per-lane access pattern as the conv kernel) - in a calibration pass pair with
# YOLO2_NO_POOLFUSE=1, because the default run fuses most pools into the conv before them.  WRITE_SIZE is exact.
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/traffic_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for CNT in FETCH_SIZE WRITE_SIZE; do
  for N in 2 6; do
    D="$OUT/${CNT}_$N"; mkdir -p "$D"
    rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$D" -- python3 bench.py --no-cpu-baseline --no-sub-records --warmup 0 --steps $N "$@" > "$D/bench.json" 2> "$D/bench.err" || { tail -20 "$D/bench.err"; exit 1; }
    echo "pass $CNT steps=$N done"
  done
done
export YOLO2_NO_POOLFUSE=1
for N in 2 6; do
  D="$OUT/CAL_FETCH_SIZE_$N"; mkdir -p "$D"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$D" -- python3 bench.py --no-cpu-baseline --no-sub-records --warmup 0 --steps $N "$@" > "$D/bench.json" 2> "$D/bench.err" || { tail -20 "$D/bench.err"; exit 1; }
  echo "calibration pass steps=$N done"
done
unset YOLO2_NO_POOLFUSE
python3 tools/traffic_report.py "$OUT" "$@" | tee "$PWD/gpurun_out/traffic_${TAG}_summary.txt"
cp "$OUT/traffic.json" "$PWD/gpurun_out/traffic_${TAG}.json"
