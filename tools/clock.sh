#!/bin/bash
# Usage (GPU box): bash tools/clock.sh <tag> [bench args]
# The clock the chip holds under each kernel: one rocprofv3 --pmc GRBM_GUI_ACTIVE pass (kernel-trace only) over a short
# bench run; per dispatch, clock = GRBM_GUI_ACTIVE / 8 XCDs / (end - start)  (MI355X_MICROARCH.md, "DVFS give-back": the
# quotient reads high on dispatches shorter than ~0.3 ms, so only longer ones are reported).  Counter collection
# serialises the dispatches: every kernel has the chip to itself (no lanes side by side).
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/clock_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -- python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
CSV=$(find "$OUT" -name "*counter_collection.csv" | head -1)
{
echo "# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -- python3 bench.py --no-cpu-baseline --no-sub-records $*"
echo "# effective clock per dispatch = GRBM_GUI_ACTIVE / 8 / duration; dispatches of >= 0.3 ms only; dispatches are serialised under counter collection"
python3 - "$CSV" <<'PY'
import csv, sys, collections, statistics
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])   # ns
    if dur < 0.3e6: continue
    k = r["Kernel_Name"].replace("HIP_vector_type<int, 2u>", "int2")[:84]
    per[k].append((float(r["Counter_Value"]) / 8.0 / dur * 1e3, dur / 1e3))   # MHz, us
print(f'{"kernel":84s} {"n":>5s} {"median MHz":>10s} {"min":>7s} {"max":>7s} {"median us":>10s}')
for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    mhz = [a for a, _ in v]
    print(f'{k:84s} {len(v):5d} {statistics.median(mhz):10.0f} {min(mhz):7.0f} {max(mhz):7.0f} {statistics.median([d for _, d in v]):10.1f}')
PY
} | tee "$PWD/gpurun_out/clock_${TAG}_summary.txt"
