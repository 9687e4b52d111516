#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 kernel trace of bench.py.
The whole-run --stats table also counts set_batch's autotune launches (every tile shape tried on every
layer), which shifts the per-kernel averages; the timed steps are the periodic tail of the dispatch
sequence, so this finds the period D (dispatches per step) and aggregates the last steps*D dispatches.
usage: steady_stats.py <kernel_trace.csv> <steps>"""
import collections, csv, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
steps = int(sys.argv[2])
sig = [(r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"]) for r in rows]
D = None
for d in range(8, len(sig) // max(steps, 2)):
    if all(sig[len(sig) - (k + 1) * d: len(sig) - k * d] == sig[len(sig) - d:] for k in range(1, steps)):
        D = d
        break
if D is None:
    sys.exit("no periodic tail found")
tail = rows[len(rows) - steps * D:]
agg = collections.defaultdict(list)
for r in tail:
    name = r["Kernel_Name"].replace("HIP_vector_type<int, 2u>", "int2").split("(")[0].replace("void ", "")
    agg[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print(f"# timed steps only: the last {steps} x {D} dispatches (D = shortest period of the dispatch sequence: one pass, or one lane's "
      f"pass where the lanes issue identical sequences); sum of kernel durations {tot/1e6/steps:.3f} ms per period, "
      f"first start to last end {span/1e6/steps:.3f} ms per period (the lanes' launches overlap)")
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:70]:70s} launches/period={len(v)//steps:3d} avg_us={sum(v)/len(v)/1e3:9.2f} min_us={min(v)/1e3:9.2f} max_us={max(v)/1e3:9.2f} pct={100*sum(v)/tot:6.2f}")
