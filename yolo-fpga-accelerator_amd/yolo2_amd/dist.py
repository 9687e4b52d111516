"""Frame sharding across GPUs: frames are independent, so the only collective is
ONE broadcast of the weight blobs at init (RCCL over xGMI when the backend is nccl); there are
no steady-state collectives.  Works on any torch.distributed backend (gloo in the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist

from . import net


def shard_range(total: int, rank: int, world: int):
    """Contiguous frame range [lo, hi) of `rank`; sizes differ by at most one.  (Same arithmetic as the library's
    yolo2_hip_shard_range, which the C host uses; tests/test_host_logic.py holds the two against each other.)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def exchange_unique_id(make_id, device: torch.device, src: int = 0) -> bytes:
    """The launcher's part of the library's one-process-per-GPU model (include/yolo2_hip.h, multi-GPU (b)): rank `src`
    makes the 128-byte RCCL id (make_id = hipdrv.rccl_unique_id -> ncclGetUniqueId), the job's process group carries it
    to every rank.  After that the weight broadcast itself is the LIBRARY's ncclBroadcast, not torch's."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    t = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.frombuffer(bytearray(make_id()), dtype=torch.uint8))
    if dist.is_initialized():
        dist.broadcast(t, src)
    return bytes(t.cpu().numpy().tobytes())


def all_ok(ok: bool, device: torch.device) -> bool:
    """True on every rank iff `ok` on every rank (one tiny all-reduce).  Failure containment for a one-process-per-GPU
    job: a rank whose local step failed must not leave the others blocked in the next barrier - every rank calls this
    after each phase and all of them leave together when any of them failed."""
    if not dist.is_initialized():
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def gather_row(values, device: torch.device):
    """Every rank contributes a row of floats; every rank gets the [world][len] table (one all-gather)."""
    row = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return row.cpu().numpy()[None, :]
    out = [torch.empty_like(row) for _ in range(dist.get_world_size())]
    dist.all_gather(out, row)
    return torch.stack(out).cpu().numpy()


def pack_q_tables(weight_q, bias_q, act_q) -> torch.Tensor:
    """[n_wq, n_bq, n_aq, values...] as one int32 tensor of fixed length 3 + 3*64."""
    t = torch.zeros(3 + 3 * 64, dtype=torch.int32)
    for i, q in enumerate((weight_q, bias_q, act_q)):
        q = np.asarray(q, dtype=np.int32)
        assert q.size <= 64
        t[i] = q.size
        t[3 + 64 * i: 3 + 64 * i + q.size] = torch.from_numpy(q.copy())
    return t


def unpack_q_tables(t: torch.Tensor):
    t = t.cpu().numpy()
    return tuple(t[3 + 64 * i: 3 + 64 * i + int(t[i])].astype(np.int32) for i in range(3))


def broadcast_model(model, device: torch.device, src: int = 0):
    """Rank `src` holds a SynthModel (others pass None).  Returns (weights int16 tensor on
    `device`, bias int16 tensor, weight_q, bias_q, act_q) on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        w = torch.from_numpy(model.weights_i16()).to(device)
        b = torch.from_numpy(model.bias_i16()).to(device)
        q = pack_q_tables(model.weight_q, model.bias_q, model.act_q).to(device)
    else:
        w = torch.empty(net.N_WEIGHTS, dtype=torch.int16, device=device)
        b = torch.empty(net.N_BIAS, dtype=torch.int16, device=device)
        q = torch.empty(3 + 3 * 64, dtype=torch.int32, device=device)
    if dist.is_initialized():
        # byte views: gloo (CPU tests) has no int16 collectives; on nccl (= RCCL) it is the same copy
        dist.broadcast(w.view(torch.uint8), src)   # 101,883,584 B: the one large collective of the job
        dist.broadcast(b.view(torch.uint8), src)
        dist.broadcast(q, src)
    wq, bq, aq = unpack_q_tables(q)
    return w, b, wq, bq, aq


def broadcast_model_f32(model, device: torch.device, src: int = 0):
    """fp32 twin (weights_reorg.bin / bias.bin contents) for the fp16 MFMA path: rank `src` holds the model."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        w = torch.from_numpy(model.weights_f32()).to(device)
        b = torch.from_numpy(model.bias_f32()).to(device)
    else:
        w = torch.empty(net.N_WEIGHTS, dtype=torch.float32, device=device)
        b = torch.empty(net.N_BIAS, dtype=torch.float32, device=device)
    if dist.is_initialized():
        dist.broadcast(w.view(torch.uint8), src)   # 203,767,168 B
        dist.broadcast(b.view(torch.uint8), src)
    return w, b
