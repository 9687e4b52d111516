#!/usr/bin/env python3
"""bench.py -- YOLOv2 INT16 416x416 frames/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the accelerator path (input quantise -> 23 conv + 5 maxpool + reorg ->
region gather) over one batch of synthetic frames already resident in HBM.  One process per
GPU; frames shard across ranks with no data-path collective (weak scaling: --batch frames per
GPU); the only collective is the broadcast of the weight blobs at init (RCCL over xGMI).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# torch (device memory, streams, torch.distributed) and the ctypes binding are imported by load_runtime(), i.e. only in a
# process that is going to run a rank: the launcher form of `python bench.py --gpus N` (spawn_ranks) starts N fresh children
# and must neither pay for the import nor come anywhere near the GPU itself.
torch = dist = ydist = hipdrv = net = synth = None


def load_runtime():
    global torch, dist, ydist, hipdrv, net, synth
    import torch as _torch
    import torch.distributed as _dist
    from yolo2_amd import dist as _ydist
    from yolo2_amd import hipdrv as _hipdrv, net as _net, synth as _synth
    torch, dist, ydist, hipdrv, net, synth = _torch, _dist, _ydist, _hipdrv, _net, _synth


HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue roofline: 256 CU x 4 SIMD issue one wave64 instruction per 2 or 4 cycles depending on
# the opcode (measured: profiles/r01_ubench_valu_issue_cost.txt).  SIMD issue cycles needed by one
# requant step (4 channels x tap x output, per wave of 64 outputs):
#   form A (MODE 0): v_mov 2 + 2 x v_dot2c 4 + v_ashrrev 4 + v_add 2 + v_med3 4 = 20
#   form B (MODE 1): 2 x v_dot2c 4 + v_and_or 4 + v_med3 4                       = 16
#   form C (MODE 3): 2 x v_dot2 4 + v_ashrrev(_sdwa) 4 + half a v_pk_add_i16 2   = 14
VALU_PEAK_TCYCLES = 256 * 4 * 2.4e9 / 1e12          # SIMD issue cycles per second at the 2.4 GHz max clock
CYCLES_PER_STEP = {0: 20, 1: 16, 2: None, 3: 14, 4: 12}


def conv_layer_bytes(l, batch):
    """Algorithmic HBM bytes of one conv launch (layer-at-a-time model)."""
    act = (l.c * l.h * l.w + l.n * l.out_h * l.out_w) * 2 * batch
    wts = (l.n * l.c * l.size * l.size + l.n) * 2
    return act + wts


def _ref_forward_one(orclib, model, frame):
    """One frame through the reference's own YOLO2_FPGA (oracle/_ref), layer by layer exactly as
    yolov2_hls_ps drives it (yolo2_model.cpp:294-425), std Q set."""
    x = np.zeros((3, 416, 416), dtype=np.int16)
    orclib.oracle().orc_quantize_input(np.ascontiguousarray(frame), x.reshape(-1), x.size, int(model.act_q[0]))
    wq, bq, aq = model.weight_q, model.bias_q, model.act_q
    outs = {}
    cur = x
    for l in net.LAYERS:
        if l.type == net.CONV:
            src = outs[16] if l.idx == 26 else (np.concatenate([outs[27], outs[24]]) if l.idx == 29 else cur)
            cur = orclib.ref_conv(src, model.w_reorg[l.ord], model.bias[l.ord], l.c, l.n, l.size, 1, l.w, l.h,
                                  l.pad, l.leaky, int(wq[l.ord]), int(aq[l.ord]), int(aq[l.ord + 1]), int(bq[l.ord]))
            outs[l.idx] = cur
        elif l.type == net.MAXPOOL:
            cur = orclib.ref_maxpool(cur, l.c, l.w, l.h)
            outs[l.idx] = cur
        elif l.type == net.REORG:
            o = np.zeros((256, 13, 16), dtype=np.int16)
            orclib.oracle().orc_reorg_i16(np.ascontiguousarray(cur), o, 0)   # std Q set: no alignment shift
            cur = o
            outs[l.idx] = cur
    return cur[:, :, :13].reshape(-1)


def cpu_baseline(model, frames, gpu_regions):
    """Times the CPU side on a bounded sample (~7 s) and checks the GPU result against it.
    Preferred: the reference itself, compiled from its own sources (oracle/_ref), driving every
    conv and maxpool layer through its YOLO2_FPGA exactly as yolov2_hls_ps does, single thread
    (the reference is not re-entrant).  Fallback: our C restatement (oracle/liboracle.so)."""
    import orclib
    n = len(frames)
    if orclib.have_ref():
        t0 = time.perf_counter()
        regions = [_ref_forward_one(orclib, model, frames[0])]      # ONE frame (about 6.5 s): keeps the default run short
        dt = time.perf_counter() - t0
        out = {"value": 1 / dt, "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "1 frame, 1 sample: 23 conv + 5 maxpool through the reference's own YOLO2_FPGA (oracle/_ref), 1 thread",
               "seconds_per_frame": dt}
    else:
        orclib.oracle().orc_set_threads(1)
        t0 = time.perf_counter()
        regions = [orclib.forward_i16(model, frames[0])[0]]
        dt = time.perf_counter() - t0
        out = {"value": 1 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "1 frame through oracle/yolo2_oracle.c (bit-exact C restatement), single thread",
               "seconds_per_frame": dt}
    out["gpu_matches_cpu_bit_exact"] = bool(np.array_equal(np.asarray(regions[0]).reshape(-1), gpu_regions[0].reshape(-1)))
    # the same work on every host core this process may use.  The reference is not
    # re-entrant (function-local statics), so this leg is the re-entrant C restatement, OpenMP over
    # output channels + AVX2 rows; bit-compared with the single-thread result above.
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)      # a one-GPU box's CPU share
    orclib.oracle().orc_set_threads(cores)
    t0 = time.perf_counter()
    mt = [orclib.forward_i16(model, f)[0] for f in frames]
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                        "sample": f"{n} frames, oracle/yolo2_oracle.c, {cores} OpenMP threads",
                        "matches_single_thread_reference": bool(np.array_equal(np.asarray(mt[0]).reshape(-1), np.asarray(regions[0]).reshape(-1))),
                        "gpu_matches_bit_exact": bool(all(np.array_equal(np.asarray(a).reshape(-1), g.reshape(-1))
                                                          for a, g in zip(mt, gpu_regions)))}
    return out


INT16_DEVICE_SOURCES = ("kernels_int16.hpp", "conv_common.hpp", "kernels_pre.hpp", "layout.hpp")


def kernel_source_hash():
    """sha256 over the device sources of the int16 path (kernels, input packing, layouts): ties the committed PMC traffic
    measurement of that path to the kernels it was taken from (the fp16 / fp32 kernel files do not enter)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc")
    for f in INT16_DEVICE_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_FILE = os.path.join("profiles", "r03_hbm_traffic.json")


def hbm_traffic_per_launch(ks, batch):
    """PMC counters cannot be collected from inside this process: tools/traffic.sh runs this same
    command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 correction on
    FETCH_SIZE, checked on a kernel with a known byte count) and commits the result under profiles/ together
    with the hash of the kernel sources it was measured on.  A file measured on other sources, another batch or
    another lane count is NOT reported (traffic: null + the reason) - a stale number is worse than none.
    Scope: mean over all launches of the conv kernel with this kernel size."""
    try:
        doc = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
    except (OSError, ValueError):
        return None, f"{TRAFFIC_FILE} not present"
    if doc.get("batch") != batch:
        return None, f"{TRAFFIC_FILE} was measured at batch {doc.get('batch')}"
    if doc.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{TRAFFIC_FILE} was measured on other kernel sources ({doc.get('kernel_source_hash')}): re-run tools/traffic.sh"
    try:
        return doc["kernels"][f"y2::k_conv_i16<KS={ks},...>"]["hbm_bytes_per_launch"], f"{TRAFFIC_FILE} (tools/traffic.sh)"
    except KeyError:
        return None, f"{TRAFFIC_FILE} holds no entry for KS={ks}"


def leave_job(rank, msg, ctx=None, code=1):
    """Every rank calls this together (after ydist.all_ok said that some rank failed): say why, tear down, exit non-zero."""
    print(f"bench.py[rank {rank}]: {msg}", file=sys.stderr, flush=True)
    try:
        if ctx is not None:
            ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
    finally:
        sys.exit(code)


def init_context(precision, rank, world, local_rank, dev):
    """Phase 1 (local): the synthetic model on rank 0, a context on this rank's GPU.  Phase 2 (collective, only with a process
    group): the 128-byte communicator id travels over torch.distributed, every rank joins the LIBRARY's RCCL communicator and
    calls the library's _bcast loader - whose failure is collective (include/yolo2_hip.h): every rank raises or none does.
    After each phase the ranks agree (ydist.all_ok) and leave together if any of them failed, so no rank is ever left alone
    in a barrier; init_process_group carries a timeout for the case of a rank that died outright."""
    model, ctx, err = None, None, None
    try:
        model = synth.SynthModel(seed=1) if rank == 0 else None
        ctx = hipdrv.Yolo2Hip(local_rank)
    except Exception as e:      # noqa: BLE001 - reported below, on every rank
        err = e
    if not ydist.all_ok(err is None, dev):
        leave_job(rank, f"context creation failed: {err}" if err else "another rank failed to create its context", ctx)
    try:
        if dist.is_initialized():   # the library's own RCCL broadcast (shared with the C host's --devices path)
            ctx.rccl_init_rank(ydist.exchange_unique_id(hipdrv.rccl_unique_id, dev), world, rank)
            if precision == "fp16":
                ctx.load_model_fp32_bcast(model, root=0)
            else:
                ctx.load_model_bcast(model, root=0)
        elif precision == "fp16":
            ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
        else:
            ctx.load_model(model)
    except Exception as e:      # noqa: BLE001
        err = e
    if not ydist.all_ok(err is None, dev):
        leave_job(rank, f"weight load failed: {err}" if err else "another rank failed to load the weights", ctx)
    return model, ctx


def rccl_record(ctx, B, steps, dt_rank, dev):
    """The bench line's "rccl" object: what the library's communicator reports on rank 0 (nranks is ncclCommCount, not WORLD_SIZE),
    the broadcast it carried, the file RCCL was resolved from - and one row per rank (its own view of the communicator, its own
    time for the timed region), so that "did RCCL see N ranks" and "did every rank do its share" are answerable from the record."""
    if not dist.is_initialized():
        return None, None
    info = ctx.rccl_info()
    rows = ydist.gather_row([info["rank"], info["device"], info["nranks"], info["bcast_ms"], dt_rank], dev)
    per_rank = [{"rank": int(r[0]), "device": int(r[1]), "nranks_seen": int(r[2]), "bcast_ms": float(r[3]),
                 "frames_per_s": B * steps / float(r[4]), "timed_region_s": float(r[4])} for r in rows]
    rec = {"nranks": info["nranks"], "rank0_bcast_ms": info["bcast_ms"], "bytes": info["bytes"], "lib_path": info["lib_path"],
           "version": info["version_str"], "version_code": info["version"], "bcasts": info["bcasts"],
           "all_ranks_agree_on_nranks": bool(all(p["nranks_seen"] == info["nranks"] for p in per_rank)),
           "note": "from the library's own communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice, dladdr of ncclBroadcast); "
                   "torch.distributed carries only the 128-byte id, the barriers and the timing reductions"}
    return rec, per_rank


print_record = None                # set by main(): writes the one JSON line to the original stdout
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


def fp16_solo_layer_ms(ctx, frames, region, Bl, stream, dev, steps=5):
    """Per-layer hipEvent times of ONE lane's launches with the chip to themselves: lanes off, batch = one lane's share.  This is
    what the per-kernel roofline object is computed from (the x lanes extrapolation of an overlapped launch's duration
    assumed perfect co-scheduling)."""
    ctx.set_fp16_lanes(1)
    for _ in range(2):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), Bl, region.data_ptr(), stream.cuda_stream)
    ctx.set_profiling(True)
    torch.cuda.synchronize(dev)
    for _ in range(steps):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), Bl, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    ms = ctx.layer_times_ms()
    ctx.set_profiling(False)
    return ms


def box_iou_vs_reference(gpu_region, ref_region, obj_thresh=0.3):
    """BASELINE.json's accuracy figure for the floating-point paths: IoU of every box the region layer decodes (all 845 cell/anchor
    slots, no threshold, no NMS: the same slot in both tensors) between the GPU tensor and the fp32 oracle's, through the host's own
    region + box code (libyolo2_host.so).  Also restricted to the slots whose reference objectness exceeds obj_thresh."""
    import orclib
    H = orclib.host()

    def rows(r):
        proc = np.zeros(425 * 169, dtype=np.float32)
        H.y2h_region_forward(np.ascontiguousarray(r, dtype=np.float32).reshape(-1), proc)
        out = np.zeros((845, 85), dtype=np.float32)
        n = H.y2h_boxes_nms(proc, 640, 480, 0.0, 0.0, out, 845)
        assert n == 845
        return out
    a, b = rows(gpu_region), rows(ref_region)
    l = np.maximum(a[:, 0] - a[:, 2] / 2, b[:, 0] - b[:, 2] / 2); r = np.minimum(a[:, 0] + a[:, 2] / 2, b[:, 0] + b[:, 2] / 2)
    tp = np.maximum(a[:, 1] - a[:, 3] / 2, b[:, 1] - b[:, 3] / 2); d = np.minimum(a[:, 1] + a[:, 3] / 2, b[:, 1] + b[:, 3] / 2)
    inter = np.clip(r - l, 0, None) * np.clip(d - tp, 0, None)
    iou = inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)
    conf = b[:, 4] > obj_thresh
    return {"boxes": 845, "box_iou_min": float(iou.min()), "box_iou_mean": float(iou.mean()),
            "confident_boxes": int(conf.sum()), "confident_box_iou_min": float(iou[conf].min()) if conf.any() else None,
            "max_abs_coord_err": float(np.abs(a[:, :4] - b[:, :4]).max()), "max_abs_objectness_err": float(np.abs(a[:, 4] - b[:, 4]).max()),
            "note": f"all 845 cell/anchor slots of frame 0 (640x480 image geometry), same slot in both tensors; confident = reference objectness > {obj_thresh}"}


MFMA_BUSY_FILE = os.path.join("profiles", "r04_f16_mfma_busy.json")


def fp16_mfma_busy():
    """Per-kernel MFMA-busy of the fp16 pass (PMC counters cannot be collected from inside this process): tools/pmc.sh + tools/mfma_busy.py
    commit it under profiles/ with the hash of csrc/kernels_f16.hpp; a file taken on other kernels is not reported."""
    import hashlib
    try:
        doc = json.load(open(os.path.join(ROOT, MFMA_BUSY_FILE)))
    except (OSError, ValueError):
        return {"source": f"{MFMA_BUSY_FILE} not present"}
    h = hashlib.sha256(open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "kernels_f16.hpp"), "rb").read()).hexdigest()[:16]
    if doc.get("kernels_f16_hash") != h:
        return {"source": f"{MFMA_BUSY_FILE} was measured on other kernels ({doc.get('kernels_f16_hash')}): re-run tools/pmc.sh + tools/mfma_busy.py"}
    return {"source": MFMA_BUSY_FILE, "definition": doc["what"],
            "per_kernel": {k: v["mfma_busy"] for k, v in doc["kernels"].items() if v["mfma_busy"] > 0}}


F16_TRAFFIC_FILE = os.path.join("profiles", "r04_f16_traffic.json")


def fp16_traffic(kernel, Bl):
    """HBM-side bytes per launch of `kernel` (a launch-table name) from the PMC passes of tools/f16_traffic.sh, committed under profiles/ with
    the hash of csrc/kernels_f16.hpp: (bytes, algorithmic bytes of the same launches, source).  None when the file is absent, was taken on
    other kernels or at another number of frames per launch."""
    import hashlib
    try:
        doc = json.load(open(os.path.join(ROOT, F16_TRAFFIC_FILE)))
    except (OSError, ValueError):
        return None, None, f"{F16_TRAFFIC_FILE} not present"
    h = hashlib.sha256(open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "kernels_f16.hpp"), "rb").read()).hexdigest()[:16]
    if doc.get("kernels_f16_hash") != h:
        return None, None, f"{F16_TRAFFIC_FILE} was measured on other kernels ({doc.get('kernels_f16_hash')}): re-run tools/f16_traffic.sh"
    e = doc.get("algorithmic", {}).get(kernel)
    if doc.get("frames_per_launch") != Bl or not e or not e.get("hbm_bytes_per_launch"):
        return None, None, f"{F16_TRAFFIC_FILE} holds no launches of {kernel} at {Bl} frames per launch"
    return e["hbm_bytes_per_launch"], e["algorithmic_bytes_per_launch"], F16_TRAFFIC_FILE


def fp16_record(ctx, B, steps, dt, layer_ms, world=1, solo_ms=None):
    """Roofline objects of the fp16 MFMA path from one timed run: `dt` seconds for `steps` passes over B frames per GPU,
    layer_ms = the library's per-layer hipEvent times of lane 0 (overlapped with the other lane's launches), solo_ms = the same
    launches with the chip to themselves (fp16_solo_layer_ms)."""
    lanes = ctx.num_lanes_fp16() if solo_ms is None else max(1, B // max(1, solo_ms[1]))
    if solo_ms is not None:
        solo_ms = solo_ms[0]
    Bl = B // lanes
    conv_ms = float(sum(layer_ms[l.idx] for l in net.CONVS))
    kern = ctx.fp16_layer_kernels()       # the launch table says which layers the dominant kernel runs (layer 8 runs it with layer 9 fused in: not counted)
    halo = [l for l in net.CONVS if kern.get(l.idx, "").startswith("k_conv_f16_halo<") and "+1x1" not in kern[l.idx]]
    halo_flops = 2.0 * Bl * sum(l.size * l.size * l.c * l.n * l.out_h * l.out_w for l in halo)
    halo_ms = float(sum((solo_ms if solo_ms is not None else layer_ms)[l.idx] for l in halo))
    halo_ach = halo_flops / (halo_ms * 1e-3) / 1e12             # solo: the launch has the chip to itself, no extrapolation
    halo_launch_ach = halo_ach
    if solo_ms is None:                                          # (multi-GPU line: no solo pass; lane 0's overlapped launches x lanes, upper bound)
        halo_ach = halo_launch_ach * lanes
    chip_ach = 2.0 * net.macs_per_frame() * B / (dt / steps) / 1e12
    traffic, traffic_alg, traffic_src = fp16_traffic("k_conv_f16_halo<256,2,16>", Bl)
    return {
        "value": world * B * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f16",
        "config": {"workload": f"YOLOv2 fp16 416x416 batch={B} per GPU, MFMA implicit-GEMM conv (fp32 accumulate)",
                   "batch_per_gpu": B, "global_batch": B * world, "lanes": lanes, "frames_per_launch": Bl,
                   "kernels": {str(i): k for i, k in sorted(kern.items())}},
        # dominant kernel: k_conv_f16_halo (3x3 layers at <= 52x52 with >= 128 output channels, as the launch table reports them)
        # - its layers' FLOPs / their hipEvent time; whole_pass = all conv layers (incl. fused pools and fused 1x1 layers)
        "roofline": {"bound": "mfma", "kernel": "k_conv_f16_halo", "launches_per_step": len(halo), "layers": [l.idx for l in halo],
                     "avg_launch_ms": halo_ms / len(halo), "achieved": halo_ach, "peak": MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": halo_ach / MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, mean over these launches; Infinity-Cache hits included)",
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": traffic_alg,
                     "algorithmic_flops_per_launch": halo_flops / len(halo), "frames_per_launch": Bl,
                     "timing": "solo" if solo_ms is not None else "overlapped_x_lanes",
                     "note": ("achieved = algorithmic FLOPs of one launch / its mean duration with the chip to itself (lanes off, batch = one "
                              "lane's share, hipEvents inside the library); the step itself runs two such launches side by side: whole_pass")
                             if solo_ms is not None else
                             "achieved = lanes x (FLOPs of lane 0's launch / its duration while the other lane's launches run beside it): an upper bound"},
        # chip level, independent of how the lanes' launches overlap: all conv FLOPs of the step / wall time of the step
        "whole_pass": {"scope": "all conv FLOPs of one step (both lanes) / ms_per_step", "achieved": chip_ach, "unit": "TFLOP/s",
                       "frac": chip_ach / MFMA_PEAK_TFLOPS},
        "mfma_busy": fp16_mfma_busy(),
        "layer_ms": [round(float(x), 4) for x in layer_ms], "conv_ms_per_step": conv_ms,
    }


def fp16_layer_table(kern, solo_ms, Bl):
    """SURVEY.md 8d for C4 ("MFMA utilisation + HBM GB/s per layer"): per LAUNCH of the fp16 launch table, with the chip to itself, the
    algorithmic FLOP/s and the algorithmic bytes/s - what the launch has to read and write as the tensors are stored (fp16 items; layer 0
    reads the fp32 frames, layer 30 writes the fp32 region tensor; a launch that fuses the pool or the 1x1 after it writes that layer's
    tensor only) - over its hipEvent time.  MFMA-busy per kernel name is the `mfma_busy` object beside it."""
    out = {}
    L = net.LAYERS
    for l in L:
        ms = float(solo_ms[l.idx])
        if ms <= 0 or l.idx not in kern or l.type not in (net.CONV, net.MAXPOOL, net.REORG):
            continue                                       # (a layer fused into the launch before it has no row: its bytes and FLOPs are that launch's)
        name = kern.get(l.idx, "")
        last = l                                           # the layer whose tensor the launch writes
        flops = 0.0
        wbytes = 0.0
        if l.type == net.CONV:
            flops = 2.0 * l.size * l.size * l.c * l.n * l.out_h * l.out_w
            wbytes = 2.0 * l.size * l.size * l.c * l.n
            nx = L[l.idx + 1] if l.idx + 1 < len(L) else None
            if nx is not None and nx.idx not in kern and nx.type in (net.MAXPOOL, net.CONV):
                last = nx                                  # fused pool / fused 1x1: no launch of its own
                if nx.type == net.CONV:
                    flops += 2.0 * nx.c * nx.n * nx.out_h * nx.out_w
                    wbytes += 2.0 * nx.c * nx.n
        in_b = l.c * l.h * l.w * (4.0 if l.idx == 0 else 2.0)
        out_b = last.n * last.out_h * last.out_w * (4.0 if last.idx == 30 else 2.0)
        byts = (in_b + out_b) * Bl + wbytes
        out[str(l.idx)] = {"kernel": name, "ms": round(ms, 4), "tflops": round(flops * Bl / (ms * 1e-3) / 1e12, 1),
                           "gbps": round(byts / (ms * 1e-3) / 1e9, 0)}
    return out


def fp16_error_vs_fp32(model, frame, gpu_region, threads):
    """Max |error| of the fp16 region tensor against the fp32 oracle (bit-exact restatement of the reference's fp32 path)."""
    import orclib
    orclib.oracle().orc_set_threads(threads)
    t0 = time.perf_counter()
    ref = orclib.forward_f32(model, frame)
    return float(np.abs(gpu_region.reshape(-1) - ref).max()), time.perf_counter() - t0


def bench_fp16(args, world, rank, local_rank, dev):
    """configs[3]: YOLOv2 fp16 MFMA path.  Same protocol as the int16 bench; rank 0 builds the fp32 weight set and ONE
    broadcast (the same one the int16 bench uses) puts it on every GPU."""
    B = args.batch
    model, ctx = init_context("fp16", rank, world, local_rank, dev)
    lo, hi = ydist.shard_range(B * world, rank, world)
    frames = torch.from_numpy(synth.frames(7, hi - lo, first=lo)).to(dev)
    region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.run_batch_fp16_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rccl, per_rank = rccl_record(ctx, B, args.steps, dt, dev)
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    layer_ms = ctx.layer_times_ms()
    ctx.set_profiling(False)
    solo = None
    if world == 1 and ctx.num_lanes_fp16() > 1:      # the per-kernel object from launches that have the chip to themselves
        lanes = ctx.num_lanes_fp16()
        region0 = region[0].clone()
        solo = (fp16_solo_layer_ms(ctx, frames, region, B // lanes, stream, dev), B // lanes)
        region[0].copy_(region0)
    if rank == 0:
        rec = fp16_record(ctx, B, args.steps, dt, layer_ms, world, solo_ms=solo)
        result = {"metric": "YOLOv2 fp16 416x416 frames/sec", "value": rec["value"], "unit": "frames/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic"}
        result.update({k: rec[k] for k in ("config", "roofline", "whole_pass", "mfma_busy", "layer_ms", "conv_ms_per_step")})
        if rccl is not None:
            result["rccl"], result["per_rank"] = rccl, per_rank
        if world == 1 and not args.no_cpu_baseline:
            import orclib
            orclib.oracle().orc_set_threads(1)
            t0c = time.perf_counter()
            ref = orclib.forward_f32(model, frames[0].cpu().numpy())
            cdt = time.perf_counter() - t0c
            g0 = region[0].cpu().numpy()
            result["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                      "sample": "1 frame through oracle/yolo2_oracle.c fp32 (bit-exact restatement of the reference's fp32 path), single thread",
                                      "seconds_per_frame": cdt, "gpu_max_abs_err_vs_cpu": float(np.abs(g0.reshape(-1) - ref).max()),
                                      "box_iou_vs_cpu": box_iou_vs_reference(g0, ref)}
        print_record(result)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def sub_fp16_b256(model, dev, steps=10, warmup=3):
    """configs[3] under the driver's clock: YOLOv2 fp16 at batch 256 on the MFMA path, as a sub-record of the default line."""
    B = 256
    ctx = hipdrv.Yolo2Hip(dev.index or 0)
    ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
    base = synth.frames(7, 8)                                   # 8 distinct frames tiled to 256 (the generator is CPU-bound)
    frames = torch.from_numpy(base).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
    region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    ctx.set_profiling(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch_fp16_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    lane_ms = ctx.layer_times_ms()
    ctx.set_profiling(False)
    lanes = ctx.num_lanes_fp16()
    region0 = region[0].cpu().numpy()
    assert torch.equal(region[0], region[B - 8]), "fp16 path: the same frame gave different results at different batch positions"
    solo = fp16_solo_layer_ms(ctx, frames, region, B // lanes, stream, dev)
    rec = fp16_record(ctx, B, steps, dt, lane_ms, solo_ms=(solo, B // lanes))
    rec["layer_ms_solo_one_lane"] = [round(float(x), 4) for x in solo]
    rec["layer_solo"] = fp16_layer_table(ctx.fp16_layer_kernels(), solo, B // lanes)
    rec["warmup"] = warmup
    threads = min(16, len(os.sched_getaffinity(0)))
    import orclib
    orclib.oracle().orc_set_threads(threads)
    t0 = time.perf_counter()
    ref = orclib.forward_f32(model, base[0])
    cdt = time.perf_counter() - t0
    err = float(np.abs(region0.reshape(-1) - ref).max())
    rec["max_abs_err_vs_fp32_oracle"] = err
    rec["box_iou_vs_fp32_oracle"] = box_iou_vs_reference(region0, ref)
    rec["error_check"] = f"frame 0 against oracle/yolo2_oracle.c fp32 on {threads} threads ({cdt:.1f} s); the region tensor spans about +-4.7"
    ctx.close()
    return rec


FP32_VALU_CYCLES_PER_STEP = 18     # 4 v_mul_f32 + 5 v_add_f32 (partial sum from +0, then the accumulator) at 2 issue cycles each


def sub_fp32_exact(model, dev, B=32, steps=3, warmup=1):
    """The exact fp32 form (configs[0]'s precision) on the tiled kernel: bit-identical to the reference's fp32 region tensor;
    VALU-issue bound like the int16 path (no FMA allowed: every product and sum is rounded like the reference's)."""
    ctx = hipdrv.Yolo2Hip(dev.index or 0)
    ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
    base = synth.frames(7, 4)
    frames = torch.from_numpy(base).to(dev).repeat(B // 4, 1, 1, 1).contiguous()
    region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        ctx.run_batch_fp32_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch_fp32_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    import orclib
    orclib.oracle().orc_set_threads(min(16, len(os.sched_getaffinity(0))))
    ref = orclib.forward_f32(model, base[1])
    exact = bool(np.array_equal(region[1].cpu().numpy().reshape(-1).view(np.uint32), ref.view(np.uint32)) and torch.equal(region[1], region[B - 3]))
    need = net.requant_steps_per_frame() * B / 64 * FP32_VALU_CYCLES_PER_STEP
    ctx.close()
    return {"metric": "YOLOv2 fp32 (reference arithmetic) 416x416 frames/sec", "value": B / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
            "steps": steps, "warmup": warmup, "dtype": "f32",
            "config": {"workload": f"YOLOv2 fp32 416x416 batch={B}, tiled conv in the reference's operation order (no FMA)"},
            "bit_exact_vs_fp32_oracle": exact,
            "valu_roofline": {"bound": "valu_issue", "achieved": need / dt / 1e12, "peak": VALU_PEAK_TCYCLES, "unit": "T SIMD issue cycles/s",
                              "frac": need / dt / 1e12 / VALU_PEAK_TCYCLES,
                              "note": "18 issue cycles per (4 channels x tap x 64 outputs): 4 v_mul_f32 + 5 v_add_f32 at 2 cycles each"}}


def sub_c5_b256(ctx, rank, world, dev, steps=5, warmup=2):
    """configs[4] (C5) at its own size, collective over all ranks: 2048 frames over 8 GPUs = a 256-frame shard per GPU (at N ranks:
    N x 256), same protocol as the headline (barrier + synchronize on both sides, max over ranks).  Every rank calls this.
    Frames: rank r holds global frames [256 r, 256 r + 32) tiled to its 256 (the frame generator is CPU-bound, 23 ms each)."""
    B = 256
    err = None
    try:
        ctx.set_batch(B)
        lo, _ = ydist.shard_range(B * world, rank, world)
        base = synth.frames(7, 32, first=lo)
        frames = torch.from_numpy(base).to(dev).repeat(B // 32, 1, 1, 1).contiguous()
        region = torch.empty((B, 425, 13, 13), dtype=torch.int16, device=dev)
    except Exception as e:      # noqa: BLE001 - a rank that cannot hold the 256-frame shard must not leave the others in a barrier
        err = e
    if not ydist.all_ok(err is None, dev):
        return {"error": f"rank {rank}: {err}" if err else "another rank could not set up its 256-frame shard"} if rank == 0 else None
    stream = torch.cuda.current_stream(dev)

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    fence()
    dt_rank = time.perf_counter() - t0
    dt = dt_rank
    rccl, per_rank = rccl_record(ctx, B, steps, dt_rank, dev)
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    same = bool(torch.equal(region[0], region[B - 32]))          # the same frame at two batch positions (two different lanes)
    if rank != 0:
        return None
    counts = ctx.layer_path_counts()
    cyc = 0.0
    for l in net.CONVS:
        cnt = counts[l.ord]
        mean_c = sum(cnt[k] * (CYCLES_PER_STEP[k] or 40) for k in range(5)) / sum(cnt)
        cyc += l.size * l.size * ((l.c + 3) // 4) * l.n * l.out_h * l.out_w * B / 64 * mean_c
    used = cyc / (dt / steps) / 1e12
    rec = {"metric": "YOLOv2 INT16 416x416 frames/sec", "value": world * B * steps / dt, "unit": "frames/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "scaling": "weak", "dtype": "int16",
           "config": {"workload": f"C5 shard size: YOLOv2 INT16 416x416 batch={B} per GPU ({B * world} frames over {world} GPU(s); "
                                  "BASELINE.json configs[4] = 2048 over 8)", "batch_per_gpu": B, "global_batch": B * world,
                      "lanes": ctx.num_lanes(), "conv_plan_source": ctx.plan_source()},
           "valu_roofline_frac": used / VALU_PEAK_TCYCLES, "same_frame_same_result_across_lanes": same}
    if rccl is not None:
        rec["rccl_nranks"], rec["per_rank"] = rccl["nranks"], per_rank
    return rec


def sub_e2e_u8(model, dev, B=64, chunks=16, thresh=0.05, nms=0.45):
    """SURVEY.md 8(d) "with and without H2D of frames", under the driver's clock: `chunks` x B synthetic 416x416x3 BYTE images in host
    memory -> yolo2_hip_run_images_u8_dets (pinned staging -> H2D -> letterbox -> int16 network -> region / boxes / NMS on the device
    -> detection records D2H; chunk n+1's upload and chunk n-1's download overlap chunk n's kernels) -> records on the host.  This is
    the PCIe-inclusive rate of the path with its two neighbours (never the headline `value`).  One chunk's records are bit-compared
    with the two-step route: region tensors to the host, yolo2_hip_postprocess_int16, best class per detection."""
    ctx = hipdrv.Yolo2Hip(dev.index or 0)
    ctx.load_model(model)
    ctx.set_batch(B)
    base = [np.clip(np.floor(f * 256.0), 0, 255).astype(np.uint8).transpose(1, 2, 0).copy() for f in synth.frames(31, 16)]
    imgs = [base[i % 16] for i in range(chunks * B)]
    hipdrv.run_images_dets(ctx._h, imgs[:2 * B], B, thresh, nms)          # untimed: staging buffers, streams, tail tables
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = hipdrv.run_images_dets(ctx._h, imgs, B, thresh, nms)
    dt = time.perf_counter() - t0
    # the check: chunk 0 through the region-tensor route
    region, q = ctx.run_images_host(imgs[:16], batch=16)
    buf = hipdrv.DevBuf(region)
    want = hipdrv.postprocess(ctx, buf.addr, 16, [416] * 16, [416] * 16, thresh, nms, final_q=q, cap=4096)
    buf.free()
    same = True
    for f in range(16):
        w = want["dets"][f]
        exp = [rows[np.argmax(rows["prob"])] for rows in (w[w["det"] == d] for d in np.unique(w["det"]))]
        exp = np.array([e for e in exp if e["prob"] > thresh], dtype=w.dtype) if exp else np.zeros(0, dtype=w.dtype)
        got = out["dets"][f]
        last = out["dets"][f + 16 * (chunks * B // 16 - 1)]       # the same image in the call's last chunk: same records but for the frame index
        fields = [n for n in got.dtype.names if n != "frame"]
        same = same and len(got) == len(exp) and np.array_equal(got, exp) and len(last) == len(got) and all(np.array_equal(last[n], got[n]) for n in fields)
    ctx.close()
    n = len(imgs)
    return {"metric": "YOLOv2 INT16 416x416 frames/sec, image bytes in host memory to detection records in host memory", "value": n / dt,
            "unit": "frames/s", "ms_per_call": dt * 1e3, "images_per_call": n, "chunk": B, "dtype": "int16",
            "records_per_frame": float(np.mean(out["counts"])), "thresh": thresh, "nms": nms,
            "config": {"workload": f"{n} synthetic 416x416x3 uint8 images (16 distinct) -> yolo2_hip_run_images_u8_dets in chunks of {B}: "
                                   "H2D + letterbox + int16 network + region/boxes/NMS on the device + records D2H, pipelined"},
            "pcie_bytes_per_frame_in": 416 * 416 * 3, "records_match_region_route_bit_exact": bool(same),
            "note": "PCIe-inclusive (SURVEY.md 8d); the headline value is device-resident by contract.  Reference loop for the behaviour: "
                    "linux_app/src/main.c:878-1288 (capture -> letterbox -> inference -> region/NMS -> records, one frame at a time)"}


def sub_fp32tol(model, dev, B=128, steps=8, warmup=2):
    """The north star's floating-point sentence under the driver's clock: the fp32 arithmetic of the path on the matrix cores INSIDE
    the 1e-3 box tolerance (yolo2_hip_run_batch_f32tol: split fp16 - hi + lo halves, three MFMAs per product).  FLOPs are counted ONCE
    (the network's 29.46 GFLOP per frame), not three times: the extra MFMAs are the price of the precision, not useful work."""
    ctx = hipdrv.Yolo2Hip(dev.index or 0)
    ctx.load_weights_fp32(model.weights_f32(), model.bias_f32())
    base = synth.frames(7, 8)
    frames = torch.from_numpy(base).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
    region = torch.empty((B, 425, 13, 13), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch_f32tol_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    assert torch.equal(region[0], region[B - 8]), "f32tol path: the same frame gave different results at different batch positions"
    import orclib
    threads = min(16, len(os.sched_getaffinity(0)))
    orclib.oracle().orc_set_threads(threads)
    ref = orclib.forward_f32(model, base[0])
    g0 = region[0].cpu().numpy()
    boxes = box_iou_vs_reference(g0, ref)
    ach = 2.0 * net.macs_per_frame() * B / dt / 1e12
    lanes = ctx.num_lanes_f32tol()
    kern = {i: ctx.f32tol_layer_kernel(i) for i in (0, 2, 4, 5, 8, 22, 30)}
    ctx.close()
    return {"metric": "YOLOv2 fp32-tolerance (split fp16 on MFMA) 416x416 frames/sec", "value": B / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
            "steps": steps, "warmup": warmup, "dtype": "f16x2 (hi + lo), f32 accumulate",
            "config": {"workload": f"YOLOv2 416x416 batch={B}, fp32 weights / activations carried as fp16 (hi, lo) pairs, a w = a_hi w_hi + a_lo w_hi + a_hi w_lo on "
                                   "v_mfma_f32_32x32x16_f16", "lanes": lanes, "kernels": kern},
            "roofline": {"bound": "mfma", "scope": "whole pass: the network's conv FLOPs counted ONCE / ms_per_step (3 MFMAs are issued per product)",
                         "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                         "mfma_issued_tflops": 3 * ach, "mfma_issued_frac": 3 * ach / MFMA_PEAK_TFLOPS, "traffic": None},
            "tolerance": {"bound_box_coord": 1e-3, "max_abs_coord_err": boxes["max_abs_coord_err"], "within": bool(boxes["max_abs_coord_err"] <= 1e-3),
                          "max_abs_raw_err_vs_fp32_oracle": float(np.abs(g0.reshape(-1) - ref).max()), "box_iou_min": boxes["box_iou_min"],
                          "max_abs_objectness_err": boxes["max_abs_objectness_err"],
                          "note": "frame 0 against oracle/yolo2_oracle.c fp32, all 845 cell/anchor slots (640x480 image geometry), no threshold"}}


def sub_latency_b1(ctx, frames, region, dev, n=30):
    """configs[1] under the driver's clock: one frame per call, one host sync per frame (device-resident in and out)."""
    stream = torch.cuda.current_stream(dev)
    ctx.set_batch(1)
    ppl = {l.idx: ctx.conv_launch_info(l.ord)["pixels_per_lane"] for l in net.CONVS}
    split = [i for i, v in ppl.items() if v == 0]
    ks = {i: -v for i, v in ppl.items() if v < 0}
    for _ in range(3):
        ctx.run_batch_ptr(frames.data_ptr(), 1, region.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        ctx.run_batch_ptr(frames.data_ptr(), 1, region.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return {"metric": "YOLOv2 INT16 416x416 single-frame latency", "value": ts[len(ts) // 2], "unit": "ms/frame", "higher_is_better": False,
            "min_ms": ts[0], "p90_ms": ts[int(0.9 * (len(ts) - 1))], "frames": n, "frames_per_s": 1e3 / ts[len(ts) // 2],
            "config": {"workload": "YOLOv2 INT16 416x416 batch=1, one host sync per frame, frame and region tensor device-resident"},
            "split_k_layers": split, "k_split_over_workgroups": ks,
            "note": "split_k_layers run k_conv_i16_splitk (the saturating chain split four or eight ways over the lanes of a wavefront, "
                    "clamp-affine maps combined with wavefront shuffles); k_split_over_workgroups {layer: splits} run k_conv_i16_ks "
                    "(the chain split over S workgroups with wave-uniform weights, triples applied in order by k_ks_finalize); "
                    "per layer as the plan table / set_batch's timing says"}


RANK_EXIT_GRACE_S = float(os.environ.get("YOLO2_BENCH_RANK_GRACE_S", "10"))


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, out_fd):
    """`python bench.py --gpus N` without a launcher around it: start N FRESH interpreter processes of this same file, one per
    GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would set them, relay
    rank 0's single JSON line to our stdout, and return the job's exit code: 0 only if every rank returned 0.  When a rank fails,
    the others get RANK_EXIT_GRACE_S to leave by themselves (bench.py's phases end collectively: ydist.all_ok) and are then
    terminated, so a dead rank never leaves a hung job behind.  This process has not imported torch and never touches the GPU."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, YOLO2_BENCH_LAUNCHER="bench.py")
        # rank 0's stdout is the record; the other ranks print nothing there, but if a library does it goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else 2, close_fds=True))
    print(f"bench.py[launcher]: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)   # never blocks the watchdog loop below
    reader.start()
    codes = [None] * n
    deadline = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes) and deadline is None:
            deadline = time.monotonic() + RANK_EXIT_GRACE_S
        if deadline is not None and time.monotonic() > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    print(f"bench.py[launcher]: rank {i} still running {RANK_EXIT_GRACE_S:.0f} s after another rank failed: terminating it",
                          file=sys.stderr, flush=True)
                    p.terminate()
                    try:
                        codes[i] = p.wait(5)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
        time.sleep(0.05)
    reader.join(5)
    line = b""
    for ln in (got[0] if got else b"").splitlines():
        if ln.strip().startswith(b"{"):
            line = ln.strip()
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py[launcher]: ranks failed (rank, exit code): {bad}", file=sys.stderr, flush=True)
        return next(c for _, c in bad if c > 0) if any(c > 0 for _, c in bad) else 1
    if not line:
        print("bench.py[launcher]: every rank returned 0 but rank 0 printed no record", file=sys.stderr, flush=True)
        return 1
    os.write(out_fd, line + b"\n")
    return 0


def dry_rank(rank, local_rank, world, emit):
    """--dry-launch: what a rank does in the launcher's rehearsal (CPU test, no GPU, no torch): the ranks meet over a real
    world-size-N gloo-free rendezvous of their own - a TCP socket on MASTER_ADDR:MASTER_PORT that rank 0 listens on - so the
    test proves that every rank got the SAME address and port, a distinct RANK / LOCAL_RANK and the right WORLD_SIZE; rank 0
    prints the one record.  YOLO2_BENCH_DRY_FAIL_RANK=r makes rank r exit 3 after the rendezvous (exit-code propagation), with
    YOLO2_BENCH_DRY_FAIL_EARLY=1 before it (the launcher must end the ranks left waiting)."""
    import socket
    addr, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
    mine = {"rank": rank, "local_rank": local_rank, "world_size": world, "master": f"{addr}:{port}", "pid": os.getpid(),
            "launcher": os.environ.get("YOLO2_BENCH_LAUNCHER", "external")}
    fail = int(os.environ.get("YOLO2_BENCH_DRY_FAIL_RANK", "-1"))
    if rank == fail and os.environ.get("YOLO2_BENCH_DRY_FAIL_EARLY") == "1":
        sys.exit(3)          # dies before the rendezvous: the others would wait for it - the launcher's watchdog ends them
    if rank == 0:
        rows = [mine]
        with socket.socket() as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(60)
            for _ in range(world - 1):
                c, _a = srv.accept()
                with c:
                    rows.append(json.loads(c.makefile().readline()))
        rows.sort(key=lambda r: r["rank"])
        emit({"dry_launch": True, "n_gpus": world, "ranks": rows})
    else:
        t_end = time.monotonic() + 60
        while True:
            try:
                with socket.create_connection((addr, port), timeout=5) as c:
                    c.sendall((json.dumps(mine) + "\n").encode())
                break
            except OSError:
                if time.monotonic() > t_end:
                    raise
                time.sleep(0.05)
    sys.exit(3 if rank == fail else 0)


def main():
    # stdout carries exactly ONE line, the JSON record.  Libraries chat on fd 1 (RCCL prints a version banner there when a
    # communicator is created under NCCL_DEBUG=VERSION/WARN), so everything written to fd 1 during the run is sent to stderr
    # and the record goes out through a private duplicate of the original stdout at the very end.
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        os.write(out_fd, (json.dumps(record) + "\n").encode())

    global print_record
    print_record = emit
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (configs[2]: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the latency_b1 (configs[1]) and fp16_b256 (configs[3]) sub-records of the default line")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the --gpus N launcher without a GPU: the ranks only report the environment they were started with")
    ap.add_argument("--precision", choices=["int16", "fp16"], default="int16",
                    help="int16 = the headline bit-exact path; fp16 = MFMA implicit-GEMM path (configs[3], use --batch 256)")
    args = ap.parse_args()

    # ---- which process is this?  (a) a rank: RANK / LOCAL_RANK / WORLD_SIZE are in the environment (torch.distributed.run, or
    # the launcher below);  (b) `python bench.py --gpus N` typed plainly with N > 1: the launcher - it starts N fresh rank
    # processes and relays rank 0's line;  (c) one GPU, no launcher needed.  Decided before torch is imported: the launcher never
    # touches the GPU (a process that has initialised HIP must not fork or exec its ranks).
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], out_fd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher (torch.distributed.run or bench.py itself) "
                  f"must start exactly --gpus ranks", file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:       # a rank of the launcher's rehearsal: report what it was given, touch nothing
        return dry_rank(rank, local_rank, world, emit)
    load_runtime()
    # device_count() does not initialise the GPU on this image; every rank sees the same number, so every rank leaves here
    # together and nobody waits in a rendezvous for a rank that has no device
    if torch.cuda.device_count() < max(1, world) or not torch.cuda.is_available():
        print(f"bench.py[rank {rank}]: --gpus {world} needs {world} GPU(s), this node shows {torch.cuda.device_count()}"
              " (the HIP path has no CPU fallback)", file=sys.stderr, flush=True)
        sys.exit(4)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # YOLO2_BENCH_FORCE_DIST=1 initialises RCCL even for one rank (rehearses the multi-GPU code path on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("YOLO2_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        # a rank that dies outright must not hang the others for ever: collectives of the process group give up after this long
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("YOLO2_BENCH_DIST_TIMEOUT_S", "300"))))

    B = args.batch
    if args.precision == "fp16":
        return bench_fp16(args, world, rank, local_rank, dev)
    # ---- init: rank 0 builds the synthetic weight set; ONE broadcast puts it on every GPU.  The broadcast is the
    # library's (yolo2_hip_load_weights_int16_bcast: ncclBroadcast from librccl.so over its own communicator, the same
    # routine the C host's `yolov2_detect --devices` uses); torch.distributed only carries the 128-byte communicator id,
    # the barrier and the max-over-ranks of the timing.
    model, ctx = init_context("int16", rank, world, local_rank, dev)
    # ---- the batch plan and this rank's shard of the global synthetic batch, resident in HBM before timing (local phase 3)
    err = None
    try:
        ctx.set_batch(B)
        lo, hi = ydist.shard_range(B * world, rank, world)
        frames = torch.from_numpy(synth.frames(7, hi - lo, first=lo)).to(dev)
        region = torch.empty((B, 425, 13, 13), dtype=torch.int16, device=dev)
    except Exception as e:      # noqa: BLE001
        err = e
    if not ydist.all_ok(err is None, dev):
        leave_job(rank, f"set_batch / frame upload failed: {err}" if err else "another rank failed in set_batch / frame upload", ctx)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.run_batch_ptr(frames.data_ptr(), B, region.data_ptr(), stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)      # hipEvent pairs around every layer launch, on `stream`
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rccl, per_rank = rccl_record(ctx, B, args.steps, dt, dev)
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    layer_ms = ctx.layer_times_ms()
    ctx.set_profiling(False)

    if rank == 0:
        fps = world * B * args.steps / dt
        paths = ctx.layer_paths()
        lanes = ctx.num_lanes()          # every layer is `lanes` concurrent part-batch launches (3 at batch 64: 21 + 21 + 22)
        Bl = B // lanes                  # frames per launch of lane 0, the one the per-layer hipEvents time (the remainder goes to the last lanes)
        # dominant kernel = the conv kernel instantiation with the largest total time
        groups = {}
        fused = ctx.pool_fused_layers()      # convs that run as k_conv_i16_pool (conv + leaky + 2x2 pool in one kernel)
        for l in net.CONVS:
            info = ctx.conv_launch_info(l.ord)
            key = (l.size, "pool" if l.idx in fused else info["pixels_per_lane"], paths[l.ord])
            g = groups.setdefault(key, {"ms": 0.0, "launches": 0, "bytes": 0.0, "steps": 0, "layers": []})
            g["ms"] += float(layer_ms[l.idx])
            g["launches"] += 1
            g["bytes"] += conv_layer_bytes(l, Bl)
            g["steps"] += l.size * l.size * ((l.c + 3) // 4) * l.n * l.out_h * l.out_w * Bl
            g["layers"].append(l.idx)
        key, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = g["ms"] / g["launches"]
        ach = (g["bytes"] / g["launches"]) / (avg_ms * 1e-3) / 1e9
        kname = f"k_conv_i16_pool<MODE={key[2]}>" if key[1] == "pool" else f"k_conv_i16<KS={key[0]},P={key[1]},MODE={key[2]}>"
        conv_ms = float(sum(layer_ms[l.idx] for l in net.CONVS))
        # VALU issue roofline over the whole step (chip level, independent of how launches overlap):
        # SIMD issue cycles the conv steps of one batch need at the measured instruction costs
        # / cycles available in ms_per_step on 1024 SIMDs at the 2.4 GHz maximum clock
        need = 0.0
        counts = ctx.layer_path_counts()
        for l in net.CONVS:
            cnt = counts[l.ord]
            if cnt[2]:              # a 64-bit block has no fixed cycle count
                need = None
                break
            mean_c = sum(cnt[k] * CYCLES_PER_STEP[k] for k in (0, 1, 3, 4)) / sum(cnt)
            need += l.size * l.size * ((l.c + 3) // 4) * l.n * l.out_h * l.out_w * B / 64 * mean_c
        valu = None
        if need is not None:
            used = need / (dt / args.steps) / 1e12
            valu = {"bound": "valu_issue", "scope": "all conv layers of one step / wall time of the step",
                    "achieved": used, "peak": VALU_PEAK_TCYCLES, "unit": "T SIMD issue cycles/s",
                    "frac": used / VALU_PEAK_TCYCLES,
                    "note": "SIMD issue cycles per wave per requant step (4 channels x tap x 64 outputs): form D 12, form C 14, "
                            "form B 16, form A 20 at the measured gfx950 issue costs; peak = 1024 SIMDs x 2.4 GHz. "
                            "The int16 path is integer-VALU bound, not HBM bound (DESIGN.md 4.1); the same kernel with staging, barriers, "
                            "scalar loads and LDS reads compiled out - a pure VALU loop over the step sequence - runs at the same rate "
                            "(profiles/r03_i16_ablation_b256.txt), and a register-only loop of the sequence costs 13.5 of these cycles per "
                            "step, not 12 (profiles/r03_ubench_step.txt)"}
        traffic, traffic_src = hbm_traffic_per_launch(key[0], B)
        fam = [l for l in net.CONVS if l.size == key[0] and l.idx not in fused]     # the scope of `traffic`: all k_conv_i16 launches of this kernel size
        fam_alg = sum(conv_layer_bytes(l, Bl) for l in fam) / len(fam)
        result = {
            "metric": "YOLOv2 INT16 416x416 frames/sec", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"YOLOv2 INT16 416x416 batch={B} per GPU, bit-exact int16 conv/bias/leaky/maxpool/reorg path",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"frames sharded x{world}, weights broadcast once",
                       "conv_paths": paths, "conv_path_block_counts": ctx.layer_path_counts(),
                       "lanes": lanes, "frames_per_launch": Bl, "conv_pool_fused_layers": fused,
                       "conv_plans": {l.idx: ctx.conv_plan(l.ord) for l in net.CONVS}, "conv_plan_source": ctx.plan_source(),
                       "options": ctx.options()},
            "roofline": {"bound": "hbm", "kernel": kname, "launches_per_step": g["launches"], "layers": g["layers"],
                         "avg_launch_ms": avg_ms, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "traffic_scope": f"mean over the {len(fam)} conv{key[0]}x{key[0]} launches per lane and step (k_conv_i16 and k_conv_i16_w16 "
                                          "instantiations of that kernel size); L2-to-fabric bytes, Infinity Cache hits included",
                         "algorithmic_bytes_per_launch_same_scope": fam_alg,
                         "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                         "binds": False,
                         "note": "BASELINE.json asks for the HBM fraction of the int16 path; by construction it is a few per cent "
                                 "(SURVEY.md 8d): the path is integer-VALU-issue bound, see binding_roofline"},
            "binding_roofline": "valu_roofline",
            "valu_roofline": valu,
            "layer_ms": [round(float(x), 4) for x in layer_ms],
            "conv_ms_per_step": conv_ms,
        }
        if rccl is not None:
            result["rccl"], result["per_rank"] = rccl, per_rank
        if world == 1 and not args.no_cpu_baseline:
            idx = [0, B - 1]    # first frame of the first lane, last frame of the last
            result["cpu_baseline"] = cpu_baseline(model, [frames[i].cpu().numpy() for i in idx],
                                                  [region[i].cpu().numpy() for i in idx])
    r0 = region[0].clone()
    if not args.no_sub_records:
        # configs[4] at its own shard size (256 frames per GPU), on EVERY rank count incl. 1: a collective sub-record
        c5 = sub_c5_b256(ctx, rank, world, dev)
        if rank == 0:
            result["c5_b256_per_gpu"] = c5
    if rank == 0:
        if world == 1 and not args.no_sub_records:
            # configs[1] and configs[3] under the same clock as the headline: ~10 s together
            result["latency_b1"] = sub_latency_b1(ctx, frames, region, dev)
            result["latency_b1"]["matches_batched_result_bit_exact"] = bool(torch.equal(region[0], r0))
            ctx.close()
            result["e2e_u8_b64"] = sub_e2e_u8(model, dev)
            result["fp16_b256"] = sub_fp16_b256(model, dev)
            result["fp32tol_b128"] = sub_fp32tol(model, dev)
            result["fp32_exact_b32"] = sub_fp32_exact(model, dev)
        print_record(result)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
