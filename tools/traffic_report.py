#!/usr/bin/env python3
"""Aggregates the four passes of tools/traffic.sh into per-kernel HBM bytes per launch."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
from yolo2_amd import net

out = sys.argv[1]
batch = 64
if "--batch" in sys.argv:
    batch = int(sys.argv[sys.argv.index("--batch") + 1])


def short(name):
    """Kernel family: set_batch's autotune may pick another pixels-per-lane instantiation from one
    pass to the next, so the conv kernel is keyed by its kernel size only (launch count per step is
    fixed), everything else by name."""
    n = name.replace("HIP_vector_type<int, 2u>", "int2").split("(")[0].replace("void ", "").strip()
    if n.startswith("y2::k_conv_i16_splitk<"):
        return "y2::k_conv_i16_splitk<KS=" + n.split("<")[1].split(",")[0] + ">"
    if n.startswith("y2::k_conv_i16_pool<"):
        return "y2::k_conv_i16_pool<...>"
    if n.startswith("y2::k_conv_i16_w16<"):      # the same layers with 16 channels per wavefront: one family with k_conv_i16
        return "y2::k_conv_i16<KS=" + n.split("<")[1].split(",")[0] + ",...>"
    if n.startswith("y2::k_conv_i16<"):
        return "y2::k_conv_i16<KS=" + n.split("<")[1].split(",")[0] + ",...>"
    return n


def load(cnt, n, prefix=""):
    f = glob.glob(os.path.join(out, f"{prefix}{cnt}_{n}", "**", "*counter_collection.csv"), recursive=True)[0]
    val = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cnt:
            continue
        k = short(r["Kernel_Name"])
        val[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return val, {k: len(v) for k, v in disp.items()}


res = {}
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    (v2, c2), (v6, c6) = load(cnt, 2), load(cnt, 6)
    for k in v6:
        dc = c6.get(k, 0) - c2.get(k, 0)
        if dc <= 0 or dc % 4:
            continue   # launched at load / autotune only, or the autotune pick differed between the passes
        e = res.setdefault(k, {"launches_per_step": dc // 4})
        e[cnt + "_KB_per_launch"] = (v6[k] - v2.get(k, 0.0)) / dc
# known bytes of the five maxpool layers (logical, unpadded): calibration of the x2 FETCH correction, from the
# calibration pass pair (YOLO2_NO_POOLFUSE=1: all five pools run as k_maxpool2)
pool_in = sum(l.c * l.h * l.w for l in net.LAYERS if l.type == net.MAXPOOL) * 2 * batch
cal = None
try:
    (v2, c2), (v6, c6) = load("FETCH_SIZE", 2, "CAL_"), load("FETCH_SIZE", 6, "CAL_")
    k = "y2::k_maxpool2"
    dc = c6.get(k, 0) - c2.get(k, 0)
    if dc > 0 and dc % 4 == 0:
        raw = (v6[k] - v2.get(k, 0.0)) * 1024 / 4
        cal = {"known_read_bytes_per_step": pool_in, "FETCH_SIZE_bytes_per_step_raw": raw, "launches_per_step": dc // 4,
               "fetch_raw_over_known": raw / pool_in, "pass": "YOLO2_NO_POOLFUSE=1 (all five pools as k_maxpool2)"}
except (IndexError, KeyError):
    pass
for k, e in res.items():
    e["hbm_bytes_per_launch"] = 2.0 * e.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024 + e.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
sys.path.insert(0, ROOT)
import bench   # the hash of the int16 device sources, one definition
doc = {"batch": batch, "kernel_source_hash": bench.kernel_source_hash(), "hashed_sources": list(bench.INT16_DEVICE_SOURCES), "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, (steps=6 - steps=2)/4; "
                                   "bytes = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KB = 1024 B",
       "calibration_on_k_maxpool2": cal, "kernels": res}
try:
    b = json.load(open(os.path.join(out, "FETCH_SIZE_6", "bench.json")))
    doc["lanes"] = b["config"]["lanes"]
    doc["conv_pool_fused_layers"] = b["config"].get("conv_pool_fused_layers")
except (OSError, KeyError, ValueError):
    pass
json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(cal, indent=1))
print(f"{'kernel':58s} {'launch/step':>11s} {'fetch MB/launch (x2)':>22s} {'write MB/launch':>16s}")
tot = 0.0
for k, e in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"]):
    tot += e["hbm_bytes_per_launch"] * e["launches_per_step"]
    print(f"{k[:58]:58s} {e['launches_per_step']:11d} {2 * e.get('FETCH_SIZE_KB_per_launch', 0) * 1024 / 1e6:22.2f} {e.get('WRITE_SIZE_KB_per_launch', 0) * 1024 / 1e6:16.2f}")
print(f"total HBM-side bytes per step: {tot / 1e6:.1f} MB  ({tot / batch / 1e6:.2f} MB/frame; algorithmic {77.326626 + 101.905106 / batch:.2f} MB/frame)")
