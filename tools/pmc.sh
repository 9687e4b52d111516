#!/bin/bash
# Usage (GPU box): bash tools/pmc.sh <tag> "<counter list>" [bench args]
# One rocprofv3 --pmc pass (no tracing flags besides kernel-trace) over a short bench run;
# prints per-kernel-name sums of each counter.
set -e
TAG=$1; CNT=$2; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
CSV=$(find "$OUT" -name "*counter_collection.csv" | head -1)
python3 - "$CSV" <<'PY' | tee "$PWD/gpurun_out/pmc_${TAG}_summary.txt"
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("HIP_vector_type<int, 2u>", "int2")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(60), "calls", *[n[:22].rjust(22) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print(k.ljust(60), str(calls[k]).rjust(5), *[f"{v.get(n,0):22.4g}" for n in names])
PY
