#!/bin/bash
# Usage (GPU box): bash tools/abn.sh <tag> <rounds> <lib1> [lib2 ...] -- [bench args]
# Like tools/ab.sh for any number of alternative builds (tools/build_variant.sh): per round one bench run of the
# default library and one of each alternative, interleaved on the same box; prints frames/s and the median of each.
TAG=$1; N=$2; shift; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
OUT=$PWD/gpurun_out/abn_$TAG; mkdir -p "$OUT"
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/default_$i.json" 2> "$OUT/default_$i.err" || exit 1
  for L in "${LIBS[@]}"; do
    K=$(basename "$L" .so)
    YOLO2_HIP_LIB=$L python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/${K}_$i.json" 2> "$OUT/${K}_$i.err" || exit 1
  done
  echo "round $i done"
done
python3 - "$OUT" $N default "${LIBS[@]}" <<'PY' | tee "$OUT/summary.txt"
import json, os, sys
out, n = sys.argv[1], int(sys.argv[2])
src = {}
for lib in sys.argv[3:]:
    k = os.path.basename(lib).replace(".so", "")
    recs = [json.load(open(f"{out}/{k}_{i}.json")) for i in range(1, n + 1)]
    v = [r["value"] for r in recs]
    src[k] = sorted({r["config"].get("conv_plan_source", "?") for r in recs})
    print(f"{k:24s}", " ".join(f"{x:8.1f}" for x in v), f"  median {sorted(v)[len(v)//2]:.1f}   plan source: {src[k]}")
# every arm must run the SAME kernel selection (variants that silently autotuned were compared with a baseline on the plan table)
if len({tuple(s) for s in src.values()}) > 1:
    sys.exit(f"A/B invalid: the arms planned differently ({src})")
PY
