#!/usr/bin/env python3
"""Per-kernel HBM-side bytes per launch of the fp16 pass from the two rocprofv3 --pmc passes of tools/f16_traffic.sh, as a JSON file
bench.py attaches to the fp16 record's roofline object (`traffic`).  bytes = 2 x FETCH_SIZE (gfx950: 128-byte requests of
16-byte-per-lane reads are tallied at 64 bytes, MI355X_MICROARCH.md) + WRITE_SIZE, counter unit KB = 1024 bytes; the x2 is checked on
k_maxpool2_f16 (layers 11 and 17: 16-byte reads of a tensor whose size is known; its padded layout adds the zero row/column items the
kernel never reads).  L2-to-fabric bytes: Infinity-Cache hits are included, so a tensor the previous launch has just written counts as
read.  The file carries the hash of csrc/kernels_f16.hpp; bench.py ignores it when the kernels have changed since.
usage: python3 tools/f16_traffic.py gpurun_out/f16_traffic_<tag> profiles/<name>.json"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yolo2_amd import net
from kernel_names import demangle

d, out = sys.argv[1], sys.argv[2]


def load(cnt):
    path = glob.glob(os.path.join(d, cnt, "**", "*counter_collection.csv"), recursive=True)[0]
    val = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cnt:
            continue
        k = demangle(r["Kernel_Name"])
        val[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    return val, {k: len(v) for k, v in disp.items()}


(fv, fc), (wv, wc) = load("FETCH_SIZE"), load("WRITE_SIZE")
bench_line = json.loads(open(os.path.join(d, "FETCH_SIZE", "bench.json")).read().strip().splitlines()[-1])
Bl = bench_line["config"]["frames_per_launch"]
kernels = {int(i): k for i, k in bench_line["config"]["kernels"].items()}
rows = {}
for k in fv:
    if "y2::" not in k or fc.get(k, 0) != wc.get(k, 0):
        continue          # not ours, or launched a different number of times in the two passes
    rd, wr = 2.0 * fv[k] * 1024 / fc[k], wv.get(k, 0.0) * 1024 / wc[k]
    rows[k] = {"launches": fc[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}


def layer_bytes(l):
    """Algorithmic bytes of one launch of layer l on Bl frames, fp16 tensors (SURVEY.md 8d's layer-at-a-time model at 2 bytes)."""
    if l.type == net.CONV:
        return (l.c * l.h * l.w + l.n * l.out_h * l.out_w) * 2 * Bl + (l.n * l.c * l.size * l.size) * 2 + l.n * 4
    return (l.c * l.h * l.w + l.out_c * l.out_h * l.out_w) * 2 * Bl


# algorithmic bytes per launch, mean over the layers a kernel name runs (every layer is launched equally often)
alg = collections.defaultdict(list)
for i, name in kernels.items():
    l = net.LAYERS[i]
    b = layer_bytes(l)
    if "+1x1" in name or "<+1x1>" in name:            # the fused 1x1 after it: its weights and ITS output instead of the 3x3's
        nx = net.LAYERS[i + 1]
        b = l.c * l.h * l.w * 2 * Bl + (l.n * l.c * 9 + nx.n * nx.c) * 2 + nx.n * nx.out_h * nx.out_w * 2 * Bl
    elif "<pool>" in name or name in ("k_conv_f16_rwc", "k_conv0_pool_mfma"):   # conv + 2x2 pool: a quarter of the output
        b = l.c * l.h * l.w * (4 if i == 0 else 2) * Bl + l.n * l.c * l.size * l.size * 2 + l.n * (l.out_h // 2) * (l.out_w // 2) * 2 * Bl
    alg[name].append(b)
# launch-table name (fp16_b256.config.kernels) -> the profiler's name of the instantiation
PROF = {"k_conv_f16_halo<256,2,16>": "y2::k_conv_f16_halo<256,2,16,32,false,false>", "k_conv_f16_halo<256,2,16>+1x1": "y2::k_conv_f16_halo<256,2,16,32,false,true>",
        "k_conv_f16_rwb<+1x1>": "y2::k_conv_f16_rwb<2>", "k_conv_f16_rwb<pool>": "y2::k_conv_f16_rwb<1>", "k_conv_f16_glds<128>": "y2::k_conv_f16_glds<128,false>",
        "k_conv_f16_glds<64>": "y2::k_conv_f16_glds<64,false>", "k_conv0_pool_mfma": "y2::k_conv0_pool_mfma<false>"}
algo = {}
for name, v in alg.items():
    pk = PROF.get(name, "y2::" + name)
    algo[name] = {"layers": [i for i, n in kernels.items() if n == name], "algorithmic_bytes_per_launch": sum(v) / len(v), "profiler_name": pk,
                  "hbm_bytes_per_launch": rows[pk]["hbm_bytes_per_launch"] if pk in rows else None}
    if pk in rows:
        algo[name]["traffic_over_algorithmic"] = rows[pk]["hbm_bytes_per_launch"] / algo[name]["algorithmic_bytes_per_launch"]
pool = rows.get("y2::k_maxpool2_f16")
cal = None
if pool:
    pl = [net.LAYERS[i] for i, n in kernels.items() if n == "k_maxpool2_f16"]
    known_r = sum(l.c * l.h * l.w * 2 * Bl for l in pl) / len(pl)
    known_w = sum(l.c * l.out_h * l.out_w * 2 * Bl for l in pl) / len(pl)
    cal = {"layers": [l.idx for l in pl], "known_read_bytes_per_launch": known_r, "FETCH_SIZE_x2_bytes_per_launch": pool["read_bytes_per_launch"],
           "read_ratio": pool["read_bytes_per_launch"] / known_r, "known_write_bytes_per_launch": known_w,
           "WRITE_SIZE_bytes_per_launch": pool["write_bytes_per_launch"], "write_ratio": pool["write_bytes_per_launch"] / known_w}
src = open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "kernels_f16.hpp"), "rb").read()
doc = {"what": "HBM-side (L2-to-fabric) bytes per launch per kernel name of the fp16 pass: 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc "
               "passes, launches serialised; Infinity-Cache hits included",
       "command": "tools/f16_traffic.sh + tools/f16_traffic.py", "kernels_f16_hash": hashlib.sha256(src).hexdigest()[:16],
       "frames_per_launch": Bl, "calibration_on_k_maxpool2_f16": cal,
       "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])),
       "algorithmic": algo}
json.dump(doc, open(out, "w"), indent=1)
print("calibration (k_maxpool2_f16):", json.dumps(cal))
for k, v in doc["kernels"].items():
    print(f"{k[:64]:64s} launches {v['launches']:4d}  read {v['read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")
for k, v in algo.items():
    print(f"algorithmic {k[:40]:40s} layers {v['layers']}  {v['algorithmic_bytes_per_launch'] / 1e6:9.2f} MB  measured / algorithmic {v.get('traffic_over_algorithmic', float('nan')):5.2f}")
