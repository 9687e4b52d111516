#!/usr/bin/env python3
"""Where a laned int16 step loses time: lane timeline of the TIMED steps from a rocprofv3 kernel trace of bench.py.
For every timed step: when each lane (HIP queue) starts and ends, and how long 3 / 2 / 1 / 0 lanes have a kernel running;
plus, per layer position in a lane's pass, the mean kernel duration.
usage: timeline.py <kernel_trace.csv> <steps> [lanes=3]"""
import collections, csv, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
steps = int(sys.argv[2])
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
byq = collections.defaultdict(list)
for r in rows:
    byq[r[qkey]].append(r)
# the lanes = the queues with the most dispatches; each lane's pass is the shortest period of ITS OWN dispatch sequence
recent = collections.Counter(r[qkey] for r in rows[-steps * 60:])     # (the null-stream queue holds set_batch's launches: not a lane)
qs = sorted(recent, key=lambda q: -recent[q])[:lanes]
def period(v):
    sig = [(r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"]) for r in v]
    for d in range(8, len(sig) // max(steps, 2) + 1):
        if all(sig[len(sig) - (k + 1) * d: len(sig) - k * d] == sig[len(sig) - d:] for k in range(1, steps)):
            return d
    return None
per = {q: period(byq[q]) for q in qs}
if any(v is None for v in per.values()):
    sys.exit(f"no periodic tail found: {per}")
n_per_lane_step = min(per.values())
print(f"# queues {qs}; dispatches per lane and step {per}; times in ms relative to the step's first kernel start")
T = lambda r, k: int(r[k])
gaps = []
for s in range(steps):
    seg = {q: byq[q][len(byq[q]) - (steps - s) * per[q]: len(byq[q]) - (steps - s - 1) * per[q]] for q in qs}
    t0 = min(T(v[0], "Start_Timestamp") for v in seg.values())
    t1 = max(T(v[-1], "End_Timestamp") for v in seg.values())
    ev = []
    for q, v in seg.items():
        for r in v:
            ev.append((T(r, "Start_Timestamp"), 1)); ev.append((T(r, "End_Timestamp"), -1))
    ev.sort()
    active, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(active, lanes)] += t - last
        last = t
        active += d
    line = f"step {s}: {(t1 - t0) / 1e6:7.3f} ms | " + " ".join(f"lane {q}: {(T(v[0], 'Start_Timestamp') - t0) / 1e6:6.3f}..{(T(v[-1], 'End_Timestamp') - t0) / 1e6:7.3f}" for q, v in seg.items())
    line += " | kernels running: " + " ".join(f"{k}:{hist[k] / 1e6:6.3f}" for k in range(lanes, -1, -1))
    print(line)
    gaps.append({k: hist[k] / 1e6 for k in range(lanes + 1)})
print("# mean over steps, ms with k kernels running concurrently:", {k: round(sum(g[k] for g in gaps) / steps, 3) for k in range(lanes, -1, -1)})
# per position in the lane's pass: mean duration and the mean number of other kernels running meanwhile is not known; print durations
pos = collections.defaultdict(list)
for q in qs:
    v = byq[q][len(byq[q]) - steps * per[q]:]
    for i, r in enumerate(v):
        pos[i % per[q]].append((T(r, "End_Timestamp") - T(r, "Start_Timestamp"), r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"], r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else ""))
print("# per position in a lane's pass: mean / min / max kernel duration (us) over lanes and steps")
for i in sorted(pos):
    d = [x[0] for x in pos[i]]
    print(f"{i:3d} {pos[i][0][1]:60s} grid {pos[i][0][2]:>9s} {sum(d) / len(d) / 1e3:9.1f} {min(d) / 1e3:9.1f} {max(d) / 1e3:9.1f}")
