#!/usr/bin/env python3
"""One period of a kernel trace by hardware queue: for the n-th last occurrence of an anchor kernel, every dispatch until the next
occurrence - queue id, start (us), duration (us), kernel, grid.  usage: queue_timeline.py <kernel_trace.csv> <anchor substring> [skip=3] [group=1]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]; skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3; group = int(sys.argv[4]) if len(sys.argv) > 4 else 1
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]][::group]
s, e = idx[-skip - 1], idx[-skip]
t0 = int(rows[s]["Start_Timestamp"])
print(collections.Counter(r["Queue_Id"] for r in rows[s:e]))
last = None
for r in rows[s:e]:
    n = r["Kernel_Name"].split("(")[0][-42:]
    if n == last and "letterbox" in n: continue
    last = n
    print(r["Queue_Id"], f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f}', n, r["Grid_Size_X"])
