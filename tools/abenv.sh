#!/bin/bash
# Usage (GPU box): bash tools/abenv.sh <tag> <rounds> "<VAR=1 ...>" ["<VAR2=1 ...>" ...] -- [bench args]
# Like tools/abn.sh, but the alternatives are environment settings of ONE library build (kernel variants the library
# selects through a diagnostic environment variable).
TAG=$1; N=$2; shift; shift
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; shift
OUT=$PWD/gpurun_out/abenv_$TAG; mkdir -p "$OUT"
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/v0_$i.json" 2> "$OUT/v0_$i.err" || exit 1
  k=1
  for E in "${ENVS[@]}"; do
    env $E python3 bench.py --no-cpu-baseline --no-sub-records "$@" > "$OUT/v${k}_$i.json" 2> "$OUT/v${k}_$i.err" || exit 1
    k=$((k+1))
  done
done
python3 - "$OUT" $N default "${ENVS[@]}" <<'PY' | tee "$OUT/summary.txt"
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for k, name in enumerate(sys.argv[3:]):
    v = [json.load(open(f"{out}/v{k}_{i}.json"))["value"] for i in range(1, n + 1)]
    print(f"{name:32s}", " ".join(f"{x:8.1f}" for x in v), f"  median {sorted(v)[len(v)//2]:.1f}")
PY
