"""CPU tests of host-side logic: C-ABI surface, integer identities the kernels rely on,
layer table, frame sharding with a real world_size-2 gloo group."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest

from yolo2_amd import hipdrv, net, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    hipdrv.build()
    L = ctypes.CDLL(hipdrv.LIB_PATH)     # loads without a GPU; no compute call is made here
    hdr = open(os.path.join(ROOT, "include", "yolo2_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:yolo2|memory|dma_buffer)_[a-z0-9_]+)\s*\(", hdr))
    declared |= {"yolo2_weight_len", "yolo2_bias_len"}
    declared -= {"yolo2_hip_det", "yolo2_hip_multi", "yolo2_hip_ctx"}      # type names, not entry points
    assert declared == set(hipdrv.EXPORTS), declared ^ set(hipdrv.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_integration_doc_indexes_every_entry_point():
    """INTEGRATION.md section E is generated from include/yolo2_hip.h (tools/abi_index.py): current, and it names every export."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr or r.stdout
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    index = doc[doc.index("<!-- abi-index:begin"):doc.index("<!-- abi-index:end -->")]
    missing = [name for name in hipdrv.EXPORTS if f"| `{name}` |" not in index and name not in ("yolo2_weight_len", "yolo2_bias_len")]
    assert not missing, missing


def test_no_gpu_means_loud_failure_not_fallback():
    L = hipdrv.lib()
    if L.yolo2_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    assert L.yolo2_accel_init() == hipdrv.YOLO2_INIT_ERROR
    h = ctypes.c_void_p(0)
    assert L.yolo2_hip_create(0, ctypes.byref(h)) == hipdrv.YOLO2_INIT_ERROR
    assert b"no CPU fallback" in L.yolo2_hip_last_error()
    with pytest.raises(hipdrv.Yolo2HipError):
        hipdrv.Yolo2Hip(0)


def test_leaky_magic_division_is_exact():
    """kernels_int16.hpp leaky_i16: floor(u/10) == (u*52429)>>19 for u in [0, 32768]."""
    u = np.arange(0, 32769, dtype=np.uint64)
    assert np.array_equal((u * 52429) >> 19, u // 10)


def test_preshifted_accumulator_identity():
    """Form B of the conv step (kernels_int16.hpp): with Bv = acc*2^s + r,
    clamp(((Bv + p) & ~(2^s-1)) | r) == sat16(acc + ((p + r) >> s))*2^s + r."""
    rng = np.random.default_rng(0)
    for s in (1, 5, 14, 15):
        r = 1 << (s - 1)
        acc = rng.integers(-32768, 32768, 200000).astype(np.int64)
        p = rng.integers(-(1 << 30), 1 << 30, 200000).astype(np.int64)
        want = np.clip(acc + ((p + r) >> s), -32768, 32767) * (1 << s) + r
        bv = acc * (1 << s) + r
        got = np.clip(((bv + p) & ~((1 << s) - 1)) | r, (-32768 << s) + r, (32767 << s) + r)
        assert np.array_equal(want, got)


def test_strip_layer_pad_matches_reference_loader():
    m = synth.SynthModel(seed=2)
    filed = synth.SynthModel._with_layer_pad(m.bias)
    dst = np.zeros(net.N_BIAS, dtype=np.int16)
    lens = (ctypes.c_int * 23)(*net.BIAS_LEN)
    n = hipdrv.lib().yolo2_strip_int16_layer_pad(filed.ctypes.data_as(ctypes.c_void_p), filed.size, lens, 23,
                                                 dst.ctypes.data_as(ctypes.c_void_p))
    assert n == net.N_BIAS and np.array_equal(dst, m.bias_i16())
    assert hipdrv.lib().yolo2_strip_int16_layer_pad(filed.ctypes.data_as(ctypes.c_void_p), 100, lens, 23,
                                                    dst.ctypes.data_as(ctypes.c_void_p)) == -1
    wl = (ctypes.c_int * 23).in_dll(hipdrv.lib(), "yolo2_weight_len")
    bl = (ctypes.c_int * 23).in_dll(hipdrv.lib(), "yolo2_bias_len")
    assert list(wl) == net.WEIGHT_LEN and list(bl) == net.BIAS_LEN


def test_layer_table_totals():
    assert net.N_WEIGHTS == 50941792 and net.N_BIAS == 10761
    assert net.macs_per_frame() == 14732084224
    assert net.requant_steps_per_frame() == 3695481088
    assert net.activation_elems_per_frame() == 38663313
    assert len(net.LAYERS) == 32 and len(net.CONVS) == 23


def test_shard_range_covers_everything():
    from yolo2_amd import dist as ydist
    for total, world in [(2048, 8), (10, 4), (3, 8), (64, 1), (0, 3), (2047, 8), (13, 3)]:
        got = [ydist.shard_range(total, r, world) for r in range(world)]
        # the C host's arithmetic (yolo2_hip_shard_range, used by yolo2_hip_multi_run_*) is the same
        assert got == [hipdrv.shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
    from yolo2_amd import dist as ydist, synth as s
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = s.SynthModel(seed=4) if rank == 0 else None
    w, b, wq, bq, aq = ydist.broadcast_model(model, torch.device("cpu"))
    # the launcher's part of the library's rank API: rank 0's 128-byte communicator id reaches every rank
    uid = ydist.exchange_unique_id(lambda: bytes(range(100, 228)), torch.device("cpu"))
    assert uid == bytes(range(100, 228)), "communicator id did not arrive"
    lo, hi = ydist.shard_range(10, rank, world)
    frames = s.frames(11, hi - lo, first=lo)
    # every rank reports a checksum of what it received and of its frame shard
    q.put((rank, int(w.to(torch.int64).sum()), int(b.to(torch.int64).sum()), list(map(int, aq)),
           lo, hi, float(frames.astype(np.float64).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_frame_sharding_world2_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    ref = synth.SynthModel(seed=4)
    for rank, ws, bs, aq, lo, hi, fs in res:
        assert ws == int(ref.weights_i16().astype(np.int64).sum())
        assert bs == int(ref.bias_i16().astype(np.int64).sum())
        assert aq == list(map(int, ref.act_q))
    assert (res[0][4], res[0][5], res[1][4], res[1][5]) == (0, 5, 5, 10)
    whole = synth.frames(11, 10).astype(np.float64)
    assert abs(res[0][6] - whole[:5].sum()) < 1e-6 and abs(res[1][6] - whole[5:].sum()) < 1e-6


def test_shard_range_rejects_bad_arguments_and_multi_needs_a_gpu():
    L = hipdrv.lib()
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    assert L.yolo2_hip_shard_range(10, 3, 3, ctypes.byref(lo), ctypes.byref(hi)) == hipdrv.YOLO2_ERROR
    assert L.yolo2_hip_shard_range(10, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) == hipdrv.YOLO2_ERROR
    if L.yolo2_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    devs = (ctypes.c_int * 2)(0, 1)
    m = ctypes.c_void_p(0)
    assert L.yolo2_hip_multi_create(devs, 2, ctypes.byref(m)) == hipdrv.YOLO2_INIT_ERROR     # loud, no CPU fallback
    assert b"no HIP device" in L.yolo2_hip_last_error()
    assert L.dma_buffer_init() == -1


def test_committed_traffic_measurement_belongs_to_the_int16_kernels_in_the_tree():
    """`roofline.traffic` is read from a committed PMC measurement (tools/traffic.sh); bench.py only reports it when the file's
    hash equals the hash of the int16 device sources.  This keeps the two together: a kernel change without a new measurement
    fails here instead of silently turning the bench field into null."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    doc = json.load(open(os.path.join(root, bench.TRAFFIC_FILE)))
    assert doc["kernel_source_hash"] == bench.kernel_source_hash(), "int16 kernels changed: re-run tools/traffic.sh and commit its JSON"
    assert doc["hashed_sources"] == list(bench.INT16_DEVICE_SOURCES)
    fam = doc["kernels"]["y2::k_conv_i16<KS=3,...>"]
    assert fam["launches_per_step"] % 3 == 0 and fam["hbm_bytes_per_launch"] > 0


def test_fast_div_magic_is_exact_for_the_layer_geometries():
    """The multiply-high division of the fp16 kernels' prologues (kernels_f16.hpp::fast_div_magic / fast_div, Granlund-Montgomery
    round-up form), restated: exact for every numerator a lane can form, for every H*W and W of the network."""
    def magic(d):
        l = 0
        while (1 << l) < d:
            l += 1
        return ((((1 << l) - d) << 32) // d + 1) & 0xFFFFFFFF, l - 1

    def fdiv(n, m, s):
        t = (m * n) >> 32
        return ((t + ((n - t) >> 1)) & 0xFFFFFFFF) >> s

    rng = np.random.default_rng(5)
    for d in sorted({l.w for l in net.CONVS} | {l.h * l.w for l in net.CONVS}):
        m, s = magic(d)
        ns = np.concatenate([np.arange(0, 4 * d + 2), rng.integers(0, 2 ** 32, 20000), np.array([2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1])])
        for k in (1, 7, 4096 * 169, 2 ** 32 // d - 1):
            ns = np.concatenate([ns, np.array([k * d - 1, k * d, k * d + 1])])
        for n in ns.tolist():
            n &= 0xFFFFFFFF
            assert fdiv(n, m, s) == n // d, (d, n)


# ------------------------------------------------------------------ round 3: launch-time extent check, failure containment

def test_fp16_plan_store_check_rejects_the_layer6_mismatch():
    """Round 2's abort inside run_batch_fp16: a kernel that stores the FULL-resolution tensor and ignores a fused
    pool (store kind 0: the persistent halo kernel) was handed layer 6's POOLED tensor - 104 x 104 offsets into a 52 x 52 tensor.
    The check every step of the fp16 launch table passes turns that into YOLO2_ERROR before anything is launched; here on explicit
    numbers, no GPU needed."""
    L = hipdrv.lib()
    B = 128
    ok = lambda *a: L.yolo2_hip_f16_store_check(*a)
    # the plan as it is: layer 6 on a pool-honouring kernel (kind 1) with pool = 1 into the 52 x 52 x 128 tensor of layer 7
    assert ok(1, 1, B, 104, 104, 128, 0, 128, B, 52, 52, 128) == hipdrv.YOLO2_SUCCESS
    # ... layer 4 on the full-resolution-only persistent kernel into its own 104 x 104 tensor
    assert ok(0, 0, B, 104, 104, 128, 0, 128, B, 104, 104, 128) == hipdrv.YOLO2_SUCCESS
    # the deliberately mismatched plan: full-resolution-only kernel, pooled destination (what round 2 launched for one commit)
    assert ok(0, 1, B, 104, 104, 128, 0, 128, B, 52, 52, 128) == hipdrv.YOLO2_ERROR
    assert b"cannot fuse the pool" in L.yolo2_hip_last_error()
    # the same kernel with the pool flag cleared but still the pooled tensor: caught by the geometry
    assert ok(0, 0, B, 104, 104, 128, 0, 128, B, 52, 52, 128) == hipdrv.YOLO2_ERROR
    assert b"104 x 104" in L.yolo2_hip_last_error() and b"52 x 52" in L.yolo2_hip_last_error()
    # a pooled store into the full-resolution tensor, a batch mismatch, channels outside the item, item size mismatch
    assert ok(2, 1, B, 104, 104, 128, 0, 128, B, 104, 104, 128) == hipdrv.YOLO2_ERROR
    assert ok(1, 0, B, 52, 52, 256, 0, 256, B // 2, 52, 52, 256) == hipdrv.YOLO2_ERROR
    assert ok(1, 0, B, 13, 13, 1280, 256, 1024, B, 13, 13, 1280) == hipdrv.YOLO2_SUCCESS        # conv 24 into the concat tensor
    assert ok(1, 0, B, 13, 13, 1280, 512, 1024, B, 13, 13, 1280) == hipdrv.YOLO2_ERROR          # ... one block too far
    assert ok(1, 0, B, 13, 13, 1024, 0, 1024, B, 13, 13, 1280) == hipdrv.YOLO2_ERROR
    assert ok(7, 0, B, 13, 13, 1024, 0, 1024, B, 13, 13, 1024) == hipdrv.YOLO2_ERROR            # unknown store kind


def _containment_worker(rank, world, port, q, fail_rank):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
    from yolo2_amd import dist as ydist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    dev = torch.device("cpu")
    err = None
    try:
        if rank == fail_rank:
            raise RuntimeError("local step failed on purpose")
    except RuntimeError as e:
        err = e
    ok = ydist.all_ok(err is None, dev)          # every rank learns that SOME rank failed ...
    rows = ydist.gather_row([rank, 10.0 + rank], dev)
    q.put((rank, ok, rows.tolist()))
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)                     # ... and all of them leave together, non-zero


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_bench_failure_containment_world2_gloo(fail_rank):
    """bench.py's phases end with ydist.all_ok: when one rank's local step fails, NO rank proceeds to the next barrier alone -
    all of them see ok == False and exit non-zero; per-rank rows travel with one all-gather (the `per_rank` field)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_containment_worker, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    want_ok = fail_rank < 0
    assert [r[1] for r in res] == [want_ok, want_ok]
    assert all(p.exitcode == (0 if want_ok else 3) for p in ps), [p.exitcode for p in ps]
    for r in res:
        assert r[2] == [[0.0, 10.0], [1.0, 11.0]]


def test_no_planning_or_launch_path_reads_the_environment():
    """26 YOLO2_* variables were read at plan time (several per plan_conv call), 4 more on the fp16 side.  Now ONE
    option set per context (Y2Options, y2_internal.hpp) is filled from the environment inside yolo2_hip_create and changed only by
    yolo2_hip_set_option: the only getenv in the library's sources is Y2Options::from_env (yolo2_plan.hip)."""
    csrc = os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        if f == "yolo2_plan.hip":
            assert code.count("getenv(") == 1 and "getenv(" in code[code.index("Y2Options Y2Options::from_env()"):code.index("std::string Y2Options::describe()")]
        else:
            assert "getenv" not in code, f
    plan = open(os.path.join(csrc, "yolo2_fp16.hip")).read()
    assert "F16Switches::from_options(c->opt)" in plan


def test_options_are_named_validated_and_parsed_once(monkeypatch):
    """yolo2_hip_set_option on the process-wide set (ctx = NULL: what the driver tier plans with): names are checked, values are
    range-checked, NULL restores the default; README.md documents every name the library knows."""
    L = hipdrv.lib()
    so = lambda n, v: L.yolo2_hip_set_option(None, n, v)
    assert so(b"force_path", b"3") == hipdrv.YOLO2_SUCCESS and so(b"force_path", None) == hipdrv.YOLO2_SUCCESS
    assert so(b"force_path", b"7") == hipdrv.YOLO2_ERROR and b"out of range" in L.yolo2_hip_last_error()
    assert so(b"force_path", b"x") == hipdrv.YOLO2_ERROR
    assert so(b"no_such_switch", b"1") == hipdrv.YOLO2_ERROR and b"no_such_switch" in L.yolo2_hip_last_error()
    assert so(b"no_ks", b"1") == hipdrv.YOLO2_SUCCESS and so(b"no_ks", b"0") == hipdrv.YOLO2_SUCCESS
    src = open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "yolo2_plan.hip")).read()
    names = set(re.findall(r'\{"([a-z0-9_]+)", &Y2Options::', src))
    assert len(names) >= 40
    readme = open(os.path.join(ROOT, "README.md")).read()
    missing = sorted(n for n in names if f"`{n}`" not in readme)
    assert not missing, f"README.md does not document: {missing}"


def test_ksplit_scratch_rule_on_plain_numbers():
    """VERDICT r3 item 6.  Round 3's GPU memory fault (gpurun_out/r3f/b64_hi.err, pytest abort in test_fullnet_ksplit_across_workgroups)
    was k_conv_i16_ks storing its triples past / without the context's scratch.  The rule plan_conv applies to every K-split plan and
    launch_conv applies again before the launch, replayed without a GPU: (a) the over-capacity case - round 3 sized the scratch for
    two layer shapes (16 splits x 64 items x 169 pixels), then planned 16 splits of a 52 x 52 layer into it; (b) the no-scratch case -
    a batch-64 lane context never allocates scratch (cap 0) but carried a forced ks field."""
    L = hipdrv.lib()
    chk = lambda *a: L.yolo2_hip_i16_plan_check(*a)
    r3_scratch = 16 * 24 * 256 * 169                       # what round 3 first allocated: 16 splits of the 13 x 13 x 1024 layers
    assert chk(16, 256, 169, r3_scratch) == hipdrv.YOLO2_SUCCESS
    assert chk(16, 64, 2704, r3_scratch) == hipdrv.YOLO2_ERROR          # layer 8/10 (52 x 52 x 256): (a) over capacity
    assert b"do not fit" in L.yolo2_hip_last_error()
    assert chk(8, 64, 2704, 8 * 24 * 64 * 2704) == hipdrv.YOLO2_SUCCESS and chk(8, 64, 2704, 8 * 24 * 64 * 2704 - 1) == hipdrv.YOLO2_ERROR
    assert chk(2, 64, 2704 * 22, 0) == hipdrv.YOLO2_ERROR               # (b): a 22-frame lane of a batch-64 context has no scratch
    assert b"no triple scratch" in L.yolo2_hip_last_error()
    for s in (0, 1, 3, 5, 32, -4):
        assert chk(s, 64, 169, 1 << 30) == hipdrv.YOLO2_ERROR
    assert chk(4, 0, 169, 1 << 30) == hipdrv.YOLO2_ERROR and chk(4, 64, 0, 1 << 30) == hipdrv.YOLO2_ERROR
    # what the scratch is sized for today: 16 splits of every layer at <= 52 x 52, per frame, batches <= 4
    for frames in (1, 2, 4):
        cap = 16 * 24 * 64 * 2704 * frames
        for cg, npix in ((64, 2704), (128, 676), (256, 169), (256, 169)):
            assert chk(16, cg, npix * frames, cap) == hipdrv.YOLO2_SUCCESS
    # the launcher refuses as well: launch_conv returns YOLO2_ERROR for a ks plan without fitting scratch (source check; the GPU
    # suite's test_fullnet_ksplit_across_workgroups drives it for real)
    src = open(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "csrc", "yolo2_int16.hip")).read()
    body = src[src.index("static int launch_conv("):src.index("// Resolve the per-layer Q values")]
    assert "if (p.ks) {" in body and "!ks_trip || !y2_ks_fits(p.ks, p.args.CGout, p.args.npix, ks_trip_bytes)" in body and "not launched" in body


def _cache_text(hash_, lines, bounds_mb=None):
    """A weight-side plan cache as the library writes it (yolo2_plan.hip Y2PlanCache::save), built by hand."""
    def fnv(data, h=1469598103934665603):
        for b in data:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    body = "Y2PLAN 2 gfx950\n# comment\n" + f"hash {hash_:016x}\n"
    for o, n in enumerate(net.BIAS_LEN):
        mb = (n + 31) // 32 if bounds_mb is None else bounds_mb
        body += f"bound {o} 1000 50 {mb}" + " 900 40 300 4 2" * mb + "\n"
    for ln in lines:
        body += "plan " + ln + "\n"
    return body + f"sum {fnv(body.encode()):016x}\n"


def test_weight_side_plan_cache_stale_or_damaged_costs_time_not_correctness(tmp_path):
    """SURVEY 8(f).2 / VERDICT r3 item 5: <weights>.y2plan holds bounds, forms and timed plans of ONE weight set.  The loader takes
    it only if header, weight hash and body checksum all match; here on files built by hand, no GPU: the intact file is accepted, a
    stale one (other hash), a damaged one (one flipped digit), a truncated one, one with an out-of-range plan line, one with trailing
    data and a file of another format are each refused as a whole - the loader then computes the bounds and times the batch as if
    no file existed (tests/test_gpu_parity.py::test_weight_side_plan_cache_* runs that on the GPU)."""
    L = hipdrv.lib()
    n = hipdrv.C.c_int(0)
    chk = lambda path, h: L.yolo2_hip_plan_cache_check(str(path).encode(), h, hipdrv.C.byref(n))
    H = 0x1234ABCD5678EF01
    good = tmp_path / "good.y2plan"
    lines = ["1 0 0 3 1 0 0 1 0 1 0 0", "1 2 0 4 2 27306 0 1 0 0 0 0", "1 23 0 4 1 0 0 1 0 0 0 8"]
    good.write_text(_cache_text(H, lines))
    assert chk(good, H) == hipdrv.YOLO2_SUCCESS and n.value == 3
    assert chk(good, H + 1) == hipdrv.YOLO2_ERROR and b"another weight set" in L.yolo2_hip_last_error()            # stale
    txt = good.read_text()
    bad = tmp_path / "flip.y2plan"
    bad.write_text(txt.replace("bound 5 1000", "bound 5 1001"))
    assert chk(bad, H) == hipdrv.YOLO2_ERROR and b"checksum" in L.yolo2_hip_last_error()                           # damaged
    bad.write_text(txt[:len(txt) // 2])
    assert chk(bad, H) == hipdrv.YOLO2_ERROR                                                                          # truncated
    bad.write_text(_cache_text(H, ["1 0 0 3 3 0 0 1 0 1 0 0"]))
    assert chk(bad, H) == hipdrv.YOLO2_ERROR and b"bad plan line" in L.yolo2_hip_last_error()                        # P = 3 is no tile shape
    bad.write_text(txt + "plan 1 4 0 4 1 0 0 1 0 0 0 0\n")
    assert chk(bad, H) == hipdrv.YOLO2_ERROR                                                                          # data after the checksum
    bad.write_text(txt.replace("Y2PLAN 2 gfx950", "Y2PLAN 1 gfx942"))
    assert chk(bad, H) == hipdrv.YOLO2_ERROR
    assert chk(tmp_path / "missing.y2plan", H) == hipdrv.YOLO2_ERROR and b"no cache file" in L.yolo2_hip_last_error()
    bad.write_text(_cache_text(H, lines).replace("bound 22 ", "bound 23 "))
    assert chk(bad, H) == hipdrv.YOLO2_ERROR


def test_committed_plan_table_is_well_formed():
    """config/plan_gfx950.txt (csrc/yolo2_int16.hip "the plan table") replaces the per-process autotune for the batches it holds.
    The loader drops any line outside what the planner can produce and autotunes a batch with a missing launch, so a damaged
    table costs time, not correctness - but the committed one must be complete: every listed batch has a line for launch 0 of
    all 23 conv layers, every field is in the planner's range, the bench's batch (64 = lanes of 21 + 21 + 22) and the
    single frame are there."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "yolo-fpga-accelerator_amd"))
    from yolo2_amd import net
    rows = {}
    for ln in open(os.path.join(root, "yolo-fpga-accelerator_amd", "config", "plan_gfx950.txt")):
        if ln.startswith("#") or not ln.strip():
            continue
        v = [int(x) for x in ln.split()]
        assert len(v) == 12, ln
        B, L, S, path, P, pad, splitk, pp, w16, fuse, hiacc, ks = v
        assert B > 0 and 0 <= L < 32 and 0 <= S <= 8 and path in (0, 1, 2, 3, 4) and P in (1, 2, 4, 8), ln
        assert pad in (0, 160 * 1024 // 6, 160 * 1024 // 4) and splitk in (0, 4, 8) and pp in (1, 2, 4), ln
        assert w16 in (0, 1) and fuse in (0, 1) and hiacc in (0, 1) and ks in (0, 2, 4, 8, 16), ln
        assert not (ks and (splitk or w16 or hiacc)) and not (splitk and w16), ln     # one kernel per launch
        assert not fuse or (L in (0, 2, 6, 10, 16) and S == 0), ln                    # only the convs in front of a pool fuse
        assert not ks or B <= 4, ln                                                   # the triple scratch exists for batches <= 4
        rows[(B, L, S)] = v
    convs = [l.idx for l in net.LAYERS if l.type == net.CONV]
    batches = sorted({k[0] for k in rows})
    assert {1, 21, 22, 64, 128, 256} <= set(batches), batches
    for B in batches:
        assert all((B, L, 0) in rows for L in convs), B
        for (b, L, S) in rows:
            assert b != B or S == 0 or (b, L, S - 1) in rows      # launches of a layer are numbered without holes


# ------------------------------------------------------------------ round 4: bench.py starts its own ranks

def _run_bench(args, env=None, timeout=120):
    import subprocess
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_launcher_starts_one_fresh_rank_per_gpu_dry():
    """`python bench.py --gpus 2` typed plainly (no torch.distributed.run, no WORLD_SIZE) must start its ranks itself
    (it exited 2).  --dry-launch rehearses exactly that launcher without a GPU: two fresh processes, distinct RANK /
    LOCAL_RANK, one WORLD_SIZE, one rendezvous address that rank 1 really reaches rank 0 on; ONE JSON line on stdout."""
    import json
    p = _run_bench(["--gpus", "2", "--dry-launch"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["dry_launch"] and rec["n_gpus"] == 2
    rows = rec["ranks"]
    assert [r["rank"] for r in rows] == [0, 1] and [r["local_rank"] for r in rows] == [0, 1]
    assert all(r["world_size"] == 2 and r["launcher"] == "bench.py" for r in rows)
    assert rows[0]["master"] == rows[1]["master"] and rows[0]["master"].startswith("127.0.0.1:")
    assert len({r["pid"] for r in rows} | {os.getpid()}) == 3          # fresh processes, not threads of the launcher


@pytest.mark.parametrize("early", ["0", "1"])
def test_bench_launcher_propagates_a_failed_rank(early):
    """A rank that exits non-zero makes the launcher exit non-zero with that code, prints no record, and - when the rank died
    before the rendezvous, so that rank 0 waits for it - ends the waiting ranks after the grace period instead of hanging."""
    p = _run_bench(["--gpus", "2", "--dry-launch"], {"YOLO2_BENCH_DRY_FAIL_RANK": "1", "YOLO2_BENCH_DRY_FAIL_EARLY": early,
                                                      "YOLO2_BENCH_RANK_GRACE_S": "1"}, timeout=60)
    assert p.returncode == 3, (p.returncode, p.stderr)
    assert p.stdout.strip() == ""
    assert "ranks failed" in p.stderr
    if early == "1":
        assert "terminating it" in p.stderr


def test_bench_without_enough_gpus_says_so_from_the_ranks():
    """On a node with fewer GPUs than --gpus every child rank says 'needs N GPU(s)' and leaves with code 4 at once (before any
    rendezvous), and the launcher returns that code: not exit 2 from argument handling, not a hang."""
    L = hipdrv.lib()
    have = L.yolo2_hip_device_count()
    n = have + 2
    p = _run_bench(["--gpus", str(n), "--steps", "1", "--warmup", "0"], timeout=300)
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    assert p.stderr.count(f"needs {n} GPU(s)") == n
    assert p.stdout.strip() == ""


def test_bench_rank_environment_without_launcher_is_still_accepted():
    """The driver's form - torch.distributed.run sets RANK / WORLD_SIZE and runs bench.py as a rank - keeps working: with
    WORLD_SIZE present bench.py is a rank, never a launcher (dry: the ranks only report)."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.pop("YOLO2_BENCH_LAUNCHER", None)
        ps.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=e,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=90) for p in ps]
    assert [p.returncode for p in ps] == [0, 0], outs
    rec = json.loads(outs[0][0].strip())
    assert [r["launcher"] for r in rec["ranks"]] == ["external", "external"] and outs[1][0].strip() == ""
    # a WORLD_SIZE that contradicts --gpus is still an argument error (exit 2)
    e = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=e, capture_output=True, text=True)
    assert p.returncode == 2 and "must start exactly" in p.stderr


def test_fp16_traffic_tool_and_bench_attachment(tmp_path):
    """tools/f16_traffic.py on synthetic counter files (the shape rocprofv3 --pmc writes): bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch and
    kernel name, the calibration row on k_maxpool2_f16, algorithmic bytes per launch-table name, the hash of csrc/kernels_f16.hpp; and
    bench.fp16_traffic() hands the halo kernel's figure to the fp16 record only when hash and frames per launch match."""
    import csv
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_names import demangle
    assert demangle("_ZN2y215k_conv_f16_haloILi256ELi2ELi16ELi32ELb0ELb0EEEvPKDF16_S2_") == "y2::k_conv_f16_halo<256,2,16,32,false,false>"
    assert demangle("_ZN2y214k_maxpool2_f16EPKDF16_PS0_iiiiiii") == "y2::k_maxpool2_f16"
    assert demangle("void y2::k_conv_i16<3, 1, 4, 2, 1>(int2 const*)") == "y2::k_conv_i16<3, 1, 4, 2, 1>"
    kernels = {"10": "k_conv_f16_halo<256,2,16>", "11": "k_maxpool2_f16", "17": "k_maxpool2_f16"}
    line = {"config": {"frames_per_launch": 2, "kernels": kernels}}
    pool_read = sum(l.c * l.h * l.w * 2 * 2 for l in net.LAYERS if l.idx in (11, 17)) / 2       # bytes one pool launch reads, mean of the two
    names = {"_ZN2y215k_conv_f16_haloILi256ELi2ELi16ELi32ELb0ELb0EEEvPKDF16_S2_": (6, 3000.0, 500.0),
             "_ZN2y214k_maxpool2_f16EPKDF16_PS0_iiiiiii": (4, pool_read / 2 / 1024, 100.0)}       # (launches, FETCH_SIZE KB, WRITE_SIZE KB) per launch
    for k, cnt in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        d = tmp_path / cnt / "run"
        d.mkdir(parents=True)
        (tmp_path / cnt / "bench.json").write_text("some log line\n" + json.dumps(line) + "\n")
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            i = 0
            for name, v in names.items():
                for _ in range(v[0]):
                    i += 1
                    w.writerow({"Dispatch_Id": i, "Kernel_Name": name, "Counter_Name": cnt, "Counter_Value": v[1 + k]})
    out = tmp_path / "traffic.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "f16_traffic.py"), str(tmp_path), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    doc = json.loads(out.read_text())
    halo = doc["kernels"]["y2::k_conv_f16_halo<256,2,16,32,false,false>"]
    assert halo["launches"] == 6 and halo["read_bytes_per_launch"] == 2 * 3000 * 1024 and halo["write_bytes_per_launch"] == 500 * 1024
    assert abs(doc["calibration_on_k_maxpool2_f16"]["read_ratio"] - 1.0) < 1e-9 and doc["calibration_on_k_maxpool2_f16"]["layers"] == [11, 17]
    l10 = net.LAYERS[10]
    alg = doc["algorithmic"]["k_conv_f16_halo<256,2,16>"]
    assert alg["layers"] == [10] and alg["hbm_bytes_per_launch"] == halo["hbm_bytes_per_launch"]
    assert alg["algorithmic_bytes_per_launch"] == (l10.c * l10.h * l10.w + l10.n * l10.out_h * l10.out_w) * 2 * 2 + l10.n * l10.c * 9 * 2 + l10.n * 4
    # bench.py attaches it only for the tree and the launch size it was measured on
    sys.path.insert(0, ROOT)
    import bench
    saved = bench.F16_TRAFFIC_FILE
    try:
        bench.F16_TRAFFIC_FILE = str(out)
        t, a, src = bench.fp16_traffic("k_conv_f16_halo<256,2,16>", 2)
        assert t == halo["hbm_bytes_per_launch"] and a == alg["algorithmic_bytes_per_launch"] and src == str(out)
        assert bench.fp16_traffic("k_conv_f16_halo<256,2,16>", 128)[0] is None          # other frames per launch
        doc["kernels_f16_hash"] = "0" * 16
        out.write_text(json.dumps(doc))
        t, a, src = bench.fp16_traffic("k_conv_f16_halo<256,2,16>", 2)
        assert t is None and "other kernels" in src
        bench.F16_TRAFFIC_FILE = str(tmp_path / "absent.json")
        assert bench.fp16_traffic("k_conv_f16_halo<256,2,16>", 2)[0] is None
    finally:
        bench.F16_TRAFFIC_FILE = saved
    # the committed file: either it belongs to the committed kernels (then the default bench line carries a sane figure) or bench.py says why not
    t, a, src = bench.fp16_traffic("k_conv_f16_halo<256,2,16>", 128)
    if t is None:
        assert "other kernels" in src or "not present" in src, src      # (kernels edited since the last tools/final_profiles.sh: re-run it on the GPU box)
    else:
        assert a and t > a and src == bench.F16_TRAFFIC_FILE, src
