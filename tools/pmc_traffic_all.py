#!/usr/bin/env python3
"""HBM-side bytes per launch of EVERY kernel of the default bench line (int16 headline + all sub-records) from two rocprofv3 --pmc passes:
  bash tools/pmc.sh allF FETCH_SIZE ; bash tools/pmc.sh allW WRITE_SIZE ; python3 tools/pmc_traffic_all.py gpurun_out/pmc_allF gpurun_out/pmc_allW
bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB = 1024 B; the x2 is calibrated for 16-byte-per-lane reads - tools/f16_traffic.py - and NOT for
byte gathers such as k_letterbox_u8_batch), mean over a kernel name's launches (a name may cover several layers), launches serialised;
durations from the same runs' kernel traces.  A triage table: where a kernel moves several times its tensors, look at its XCD order."""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import demangle


def load(d, cnt):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    v = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cnt:
            k = demangle(r["Kernel_Name"]).replace("HIP_vector_type<int, 2u>", "int2")
            v[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])):
        t[demangle(r["Kernel_Name"]).replace("HIP_vector_type<int, 2u>", "int2")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return v, {k: len(x) for k, x in n.items()}, t


(fv, fc, ft), (wv, wc, _) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fv:
    if fc[k] != wc.get(k, -1):
        continue
    rd, wr = 2 * fv[k] * 1024 / fc[k], wv[k] * 1024 / fc[k]
    ms = sum(ft[k]) / len(ft[k])
    rows.append(((rd + wr) * fc[k], k, fc[k], rd, wr, ms))
print(f"{'kernel':76s} {'launches':>8s} {'read MB':>9s} {'write MB':>9s} {'ms':>8s} {'TB/s':>6s}")
for _, k, n, rd, wr, ms in sorted(rows, reverse=True)[:48]:
    print(f"{k[:76]:76s} {n:8d} {rd / 1e6:9.2f} {wr / 1e6:9.2f} {ms:8.4f} {(rd + wr) / ms / 1e9:6.2f}")
