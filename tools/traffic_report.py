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
    if n.startswith("y2::k_conv_i16<"):
        return "y2::k_conv_i16<KS=" + n.split("<")[1].split(",")[0] + ",...>"
    return n


def load(cnt, n):
    f = glob.glob(os.path.join(out, f"{cnt}_{n}", "**", "*counter_collection.csv"), recursive=True)[0]
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
# known bytes of the five maxpool layers (logical, unpadded): calibration of the x2 FETCH correction
pool_in = sum(l.c * l.h * l.w for l in net.LAYERS if l.type == net.MAXPOOL) * 2 * batch
pool_out = sum(l.c * l.out_h * l.out_w for l in net.LAYERS if l.type == net.MAXPOOL) * 2 * batch
cal = None
if "y2::k_maxpool2" in res:
    p = res["y2::k_maxpool2"]
    cal = {"known_read_bytes_per_step": pool_in, "FETCH_SIZE_bytes_per_step_raw": p["FETCH_SIZE_KB_per_launch"] * 1024 * p["launches_per_step"],
           "known_write_bytes_per_step": pool_out, "WRITE_SIZE_bytes_per_step_raw": p["WRITE_SIZE_KB_per_launch"] * 1024 * p["launches_per_step"]}
    cal["fetch_raw_over_known"] = cal["FETCH_SIZE_bytes_per_step_raw"] / pool_in
    cal["write_raw_over_known"] = cal["WRITE_SIZE_bytes_per_step_raw"] / pool_out
for k, e in res.items():
    e["hbm_bytes_per_launch"] = 2.0 * e.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024 + e.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
doc = {"batch": batch, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, (steps=6 - steps=2)/4; "
                                   "bytes = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KB = 1024 B",
       "calibration_on_k_maxpool2": cal, "kernels": res}
json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(cal, indent=1))
print(f"{'kernel':58s} {'launch/step':>11s} {'fetch MB/launch (x2)':>22s} {'write MB/launch':>16s}")
tot = 0.0
for k, e in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"]):
    tot += e["hbm_bytes_per_launch"] * e["launches_per_step"]
    print(f"{k[:58]:58s} {e['launches_per_step']:11d} {2 * e.get('FETCH_SIZE_KB_per_launch', 0) * 1024 / 1e6:22.2f} {e.get('WRITE_SIZE_KB_per_launch', 0) * 1024 / 1e6:16.2f}")
print(f"total HBM-side bytes per step: {tot / 1e6:.1f} MB  ({tot / batch / 1e6:.2f} MB/frame; algorithmic {77.326626 + 101.905106 / batch:.2f} MB/frame)")
