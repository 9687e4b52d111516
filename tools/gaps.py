#!/usr/bin/env python3
"""Device idle time in a rocprofv3 kernel trace: union of all kernel intervals over the last `frac` of the run, the gaps between them
(count, total, the largest with the kernels on either side) and busy time by kernel.  usage: gaps.py <kernel_trace.csv> [frac=0.5]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = max(int(r["End_Timestamp"]) for r in rows)
cut = t_hi - (t_hi - t_lo) * frac
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
cur_end, cur_name, busy, gaps = None, None, 0, []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); name = r["Kernel_Name"].split("(")[0][-40:]
    if cur_end is None: cur_start, cur_end, cur_name = s, e, name; continue
    if s > cur_end:
        busy += cur_end - cur_start; gaps.append((s - cur_end, cur_name, name, cur_end)); cur_start, cur_end, cur_name = s, e, name
    elif e > cur_end: cur_end, cur_name = e, name
busy += cur_end - cur_start
span = cur_end - int(rows[0]["Start_Timestamp"])
print(f"window {span / 1e6:.2f} ms: some kernel running {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), {len(gaps)} gaps totalling {sum(g[0] for g in gaps) / 1e6:.2f} ms")
hist = collections.Counter()
for g in gaps: hist[(g[1], g[2])] += g[0]
for (a, b), t in hist.most_common(12): print(f"  {t / 1e6:8.3f} ms idle between {a} -> {b}  ({sum(1 for g in gaps if (g[1], g[2]) == (a, b))} times)")
