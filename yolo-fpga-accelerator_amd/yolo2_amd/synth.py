"""Seeded synthetic weights / frames in the reference's on-disk formats.

Real YOLOv2 weights are not obtainable offline (weights/README.md:19-66 of the reference
fetches them over the network), so every run here uses synthetic data of the reference's
shapes.  The generator is integer-only after a per-layer scale is fixed, so the same seed
gives the same bytes on every machine.

On-disk formats produced (hls/models/yolov2/yolo2_model.cpp:171-193):
  weights_reorg_int16.bin  int16, per conv layer blocks m0(32)/n0(4)/[kk][tm][tn], 1 pad elem after odd layers
  bias_int16.bin           int16, dense per layer, 1 pad elem after odd layers (only the last, 425)
  weight_int16_Q.bin, bias_int16_Q.bin, iofm_Q.bin   int32 tables
  weights_reorg.bin, bias.bin                        fp32 twins (dequantised int16 values)
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np

from . import net

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _counter(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return (np.uint64(seed) * np.uint64(0x100000001B3) + np.arange(n, dtype=np.uint64)) & _M64


def gauss_int(seed: int, n: int) -> np.ndarray:
    """Irwin-Hall(4) of 16-bit uniforms, centred: int64 in [-131070, 131070], std = 65536/sqrt(3)."""
    u = splitmix64(_counter(seed, n))
    s = (u & np.uint64(0xFFFF)) + ((u >> np.uint64(16)) & np.uint64(0xFFFF)) \
        + ((u >> np.uint64(32)) & np.uint64(0xFFFF)) + (u >> np.uint64(48))
    return s.astype(np.int64) - 131070


GAUSS_STD = 65536.0 / math.sqrt(3.0)


def scaled_gauss_i16(seed: int, n: int, std_int: float) -> np.ndarray:
    mult = int(round(std_int / GAUSS_STD * (1 << 24)))
    v = (gauss_int(seed, n) * mult) >> 24
    return np.clip(v, -32768, 32767).astype(np.int16)


def frames(seed: int, count: int, first: int = 0) -> np.ndarray:
    """float32 [count][3][416][416] in [0,1): frame f depends only on (seed, f)."""
    out = np.empty((count, 3, 416, 416), dtype=np.float32)
    n = 3 * 416 * 416
    for f in range(count):
        u = splitmix64(_counter((seed << 20) + first + f + 1, n))
        out[f] = ((u >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(3, 416, 416)
    return out


def reorg_weights(w_nat: np.ndarray, C: int, N: int, K: int) -> np.ndarray:
    """Natural [N][C][K][K] -> the accelerator's tiled stream.
    Same format as src/models/yolov2/yolov2_weight_gen.cpp:34-68 (m0 step 32, n0 step 4,
    block [k*k][TM_MIN][TN_MIN], no padding of partial tiles); checked byte-for-byte against
    that tool in tests/test_oracle_vs_ref.py."""
    w = w_nat.reshape(N, C, K * K)
    parts = []
    for m0 in range(0, N, net.TM):
        tm = min(net.TM, N - m0)
        for n0 in range(0, C, net.TN):
            tn = min(net.TN, C - n0)
            blk = w[m0:m0 + tm, n0:n0 + tn, :]            # [tm][tn][kk]
            parts.append(np.ascontiguousarray(blk.transpose(2, 0, 1)).reshape(-1))
    return np.concatenate(parts)


STD_Q = dict(weight_q=14, bias_q=12, act_q_in=14, act_q=9)


class SynthModel:
    """Synthetic int16 YOLOv2 weight set (+ exact fp32 twin)."""

    def __init__(self, seed: int = 1, weight_q: Optional[List[int]] = None,
                 bias_q: Optional[List[int]] = None, act_q: Optional[List[int]] = None,
                 gain: float = 1.0, obj_bias: float = 0.0):
        nconv = len(net.CONVS)
        self.seed = seed
        self.weight_q = np.array(weight_q if weight_q is not None else [STD_Q["weight_q"]] * nconv, dtype=np.int32)
        self.bias_q = np.array(bias_q if bias_q is not None else [STD_Q["bias_q"]] * nconv, dtype=np.int32)
        self.act_q = np.array(act_q if act_q is not None else [STD_Q["act_q_in"]] + [STD_Q["act_q"]] * nconv,
                              dtype=np.int32)
        self.w_nat: List[np.ndarray] = []     # int16 [N][C][K][K]
        self.w_reorg: List[np.ndarray] = []   # int16 stream
        self.bias: List[np.ndarray] = []      # int16 [N]
        for l in net.CONVS:
            fan_in = l.c * l.size * l.size
            std = gain * math.sqrt(2.0 / fan_in)
            qw, qb = int(self.weight_q[l.ord]), int(self.bias_q[l.ord])
            w = scaled_gauss_i16(seed * 1000 + l.ord * 2 + 1, l.n * fan_in, std * (1 << qw))
            b = scaled_gauss_i16(seed * 1000 + l.ord * 2 + 2, l.n, 0.05 * (1 << qb))
            if l.ord == nconv - 1 and obj_bias != 0.0:
                b = b.copy()
                b[4::85] = np.int16(max(-32768, min(32767, int(round(obj_bias * (1 << qb))))))
            self.w_nat.append(w.reshape(l.n, l.c, l.size, l.size))
            self.w_reorg.append(reorg_weights(w, l.c, l.n, l.size))
            self.bias.append(b)

    # ---- flat blobs in the form the accelerator consumes (no per-layer file pad)
    def weights_i16(self) -> np.ndarray:
        return np.concatenate(self.w_reorg)

    def bias_i16(self) -> np.ndarray:
        return np.concatenate(self.bias)

    def weights_f32(self) -> np.ndarray:
        return np.concatenate([w.astype(np.float32) * np.float32(2.0 ** -int(self.weight_q[i]))
                               for i, w in enumerate(self.w_reorg)])

    def weights_nat_f32(self) -> np.ndarray:
        return np.concatenate([w.reshape(-1).astype(np.float32) * np.float32(2.0 ** -int(self.weight_q[i]))
                               for i, w in enumerate(self.w_nat)])

    def bias_f32(self) -> np.ndarray:
        return np.concatenate([b.astype(np.float32) * np.float32(2.0 ** -int(self.bias_q[i]))
                               for i, b in enumerate(self.bias)])

    # ---- files exactly as yolo2_model.cpp:158-227 reads them
    @staticmethod
    def _with_layer_pad(parts: List[np.ndarray]) -> np.ndarray:
        out = []
        for p in parts:
            out.append(p)
            if len(p) & 1:
                out.append(np.zeros(1, dtype=p.dtype))
        return np.concatenate(out)

    def write_files(self, weights_dir: str, fp32: bool = True, int16: bool = True,
                    natural: bool = False) -> Dict[str, str]:
        os.makedirs(weights_dir, exist_ok=True)
        paths = {}

        def put(name, arr):
            p = os.path.join(weights_dir, name)
            arr.tofile(p)
            paths[name] = p

        if int16:
            put("weights_reorg_int16.bin", self._with_layer_pad(self.w_reorg))
            put("bias_int16.bin", self._with_layer_pad(self.bias))
            put("weight_int16_Q.bin", self.weight_q)
            put("bias_int16_Q.bin", self.bias_q)
            put("iofm_Q.bin", self.act_q)
        if fp32:
            put("weights_reorg.bin", self.weights_f32())
            put("bias.bin", self.bias_f32())
        if natural:
            put("weights.bin", self.weights_nat_f32())
            put("weight_int16.bin", self._with_layer_pad([w.reshape(-1) for w in self.w_nat]))
        return paths
