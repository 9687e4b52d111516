#!/bin/bash
# Usage (GPU box): bash tools/ab.sh <tag> <alt-lib> [rounds] [bench args]
# A/B of two builds of libyolo2_hip.so (e.g. one compiled with another -D kernel variant into
# yolo-fpga-accelerator_amd/build/): alternating bench runs in ONE call, same box, same clocks; prints frames/s of each.
TAG=$1; ALT=$2; N=${3:-3}; shift; shift; shift
OUT=$PWD/gpurun_out/ab_$TAG; mkdir -p "$OUT"
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/a_$i.json" 2> "$OUT/a_$i.err" || exit 1
  YOLO2_HIP_LIB=$ALT python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/b_$i.json" 2> "$OUT/b_$i.err" || exit 1
done
python3 - "$OUT" $N <<'PY' | tee "$OUT/summary.txt"
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
src = {}
for k in "ab":
    recs = [json.load(open(f"{out}/{k}_{i}.json")) for i in range(1, n + 1)]
    v = [r["value"] for r in recs]
    src[k] = sorted({r["config"].get("conv_plan_source", "?") for r in recs})
    print(k, "default" if k == "a" else "alt    ", " ".join(f"{x:8.1f}" for x in v), f"  median {sorted(v)[len(v)//2]:.1f}   plan source: {src[k]}")
# both arms must run the SAME kernel selection (a variant that silently autotuned was compared with a baseline on the plan table)
if src["a"] != src["b"]:
    sys.exit(f"A/B invalid: the arms planned differently ({src})")
PY
