#!/usr/bin/env python3
"""Per-kernel MFMA-busy of the fp16 pass from one rocprofv3 --pmc pass (tools/pmc.sh <tag> "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --no-sub-records --precision fp16 --batch 256 --steps 3 --warmup 1), as a JSON file bench.py attaches to the
fp16 record: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = the share of the SIMD-cycles of the
kernel's own run time in which the matrix pipe is occupied (dispatches are serialised under counter collection: every launch has the
chip to itself).  The file carries the hash of csrc/kernels_f16.hpp; bench.py ignores it when the kernels have changed since.
usage: python3 tools/mfma_busy.py gpurun_out/pmc_<tag> profiles/<name>.json"""
import collections, csv, glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, out = sys.argv[1], sys.argv[2]
path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); calls[k] += 1
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_names import demangle


rows = {}
for k, v in agg.items():
    if v.get("GRBM_GUI_ACTIVE", 0) <= 0 or "y2" not in k:
        continue
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
    rows[demangle(k)] = {"launches": calls[k], "mfma_busy": round(busy, 4), "gui_active_cycles_sum": v["GRBM_GUI_ACTIVE"]}
src = open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "kernels_f16.hpp"), "rb").read()
doc = {"what": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024) per kernel name, one rocprofv3 --pmc pass, launches serialised",
       "command": "tools/pmc.sh + tools/mfma_busy.py", "kernels_f16_hash": hashlib.sha256(src).hexdigest()[:16],
       "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["gui_active_cycles_sum"]))}
json.dump(doc, open(out, "w"), indent=1)
for k, v in doc["kernels"].items():
    print(f"{k[:70]:70s} launches {v['launches']:4d}  mfma_busy {v['mfma_busy'] * 100:5.1f} %")
