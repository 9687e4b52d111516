#!/bin/bash
# Usage (on the GPU box, via gpurun):  bash tools/profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats of bench.py under gpurun_out/prof_<tag>/ and a compact
# summary to gpurun_out/prof_<tag>_summary.txt (copy that into profiles/ to have it judged).
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
STATS=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-sub-records $*"
  echo "# bench line:"; tail -1 "$OUT/bench.json"
  echo "# kernel stats of the WHOLE run (weight load, set_batch - no autotune launches when the plan table covers the batch -, warm-up and timed steps):"
  python3 - "$STATS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    name = r["Name"]
    name = name.replace("HIP_vector_type<int, 2u>", "int2")[:110]
    print(f'{name:110s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:10.3f} avg_us={float(r["AverageNs"])/1e3:10.2f} pct={float(r["Percentage"]):6.2f}')
PY
  TRACE=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
  STEPS=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['steps'])")
  python3 tools/steady_stats.py "$TRACE" "$STEPS" || true
} > "$PWD/gpurun_out/prof_${TAG}_summary.txt"
cat "$PWD/gpurun_out/prof_${TAG}_summary.txt"
