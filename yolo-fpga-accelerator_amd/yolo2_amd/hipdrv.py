"""ctypes binding of libyolo2_hip.so (include/yolo2_hip.h).

This is host plumbing only: every compute call goes through the C ABI into the hand-written
HIP kernels.  There is no CPU fallback: if the library is missing or no GPU is present the
calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import net

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# YOLO2_HIP_LIB: another build of the same library (A/B runs of compile-time kernel variants, tools/ab.sh)
LIB_PATH = os.environ.get("YOLO2_HIP_LIB") or os.path.join(PKG_DIR, "libyolo2_hip.so")

YOLO2_SUCCESS, YOLO2_ERROR, YOLO2_TIMEOUT, YOLO2_INIT_ERROR, YOLO2_MMAP_ERROR, YOLO2_DMA_ERROR = 0, -1, -2, -3, -4, -5

EXPORTS = [
    "yolo2_accel_init", "yolo2_accel_cleanup", "yolo2_hip_select_device", "yolo2_hip_device_count",
    "yolo2_hip_last_error", "yolo2_set_q_values", "yolo2_is_busy", "yolo2_is_done", "yolo2_wait_for_completion",
    "yolo2_execute_conv_layer", "yolo2_execute_maxpool_layer", "yolo2_execute_conv_layer_f32",
    "memory_allocate_ddr", "memory_free_ddr", "memory_allocate_weights", "memory_allocate_bias",
    "memory_allocate_inference_buffer", "memory_get_phys_addr", "memory_flush_cache", "memory_invalidate_cache",
    "yolo2_hip_alloc", "yolo2_hip_free", "yolo2_hip_memcpy_h2d", "yolo2_hip_memcpy_d2h", "yolo2_hip_memset",
    "yolo2_hip_create", "yolo2_hip_destroy", "yolo2_hip_load_weights_int16", "yolo2_hip_load_weights_int16_dev",
    "yolo2_hip_layer_path", "yolo2_hip_set_batch", "yolo2_hip_run_batch_int16", "yolo2_hip_run_batch_int16_host",
    "yolo2_hip_debug_layer_output", "yolo2_hip_set_profiling", "yolo2_hip_layer_times_ms",
    "yolo2_hip_conv_launch_info", "yolo2_strip_int16_layer_pad", "yolo2_weight_len", "yolo2_bias_len",
    "yolo2_hip_num_layers", "yolo2_hip_layer_desc",
    "yolo2_hip_layer_path_counts", "yolo2_hip_run_frames_int16", "yolo2_hip_num_lanes", "yolo2_hip_load_weights_fp32", "yolo2_hip_run_batch_fp16", "yolo2_hip_run_batch_fp16_host",
    "yolo2_hip_letterbox_u8", "yolo2_hip_run_images_u8_host", "yolo2_hip_last_layer_path", "yolo2_hip_run_frame_fp32_host", "yolo2_hip_num_lanes_fp16",
    "yolo2_hip_layer_pool_fused", "yolo2_get_status", "yolo2_read_reg", "yolo2_write_reg", "yolo2_hip_driver_calls",
    "dma_buffer_init", "dma_buffer_cleanup", "dma_buffer_alloc", "dma_buffer_free", "dma_buffer_sync_for_device",
    "dma_buffer_sync_for_cpu", "dma_buffer_get_phys",
    "yolo2_hip_run_batch_fp32", "yolo2_hip_run_batch_fp32_host", "yolo2_hip_load_weights_fp32_dev", "yolo2_hip_postprocess_int16", "yolo2_hip_postprocess_f32",
    "yolo2_hip_shard_range", "yolo2_hip_multi_create", "yolo2_hip_multi_destroy", "yolo2_hip_multi_num_devices",
    "yolo2_hip_multi_uses_rccl", "yolo2_hip_multi_ctx", "yolo2_hip_multi_load_weights_int16", "yolo2_hip_multi_load_weights_fp32",
    "yolo2_hip_multi_run_frames_int16", "yolo2_hip_multi_run_images_u8_host", "yolo2_hip_rccl_unique_id",
    "yolo2_hip_rccl_init_rank", "yolo2_hip_rccl_finalize", "yolo2_hip_load_weights_int16_bcast", "yolo2_hip_load_weights_fp32_bcast",
    "yolo2_hip_rccl_info", "yolo2_hip_multi_rccl_info", "yolo2_hip_ctx_device", "yolo2_hip_alloc_on",
    "yolo2_hip_fp16_layer_kernel", "yolo2_hip_f16_store_check",
    "yolo2_hip_run_images_u8_dets", "yolo2_hip_multi_run_images_u8_dets", "yolo2_hip_set_fp16_lanes", "yolo2_hip_conv_plan_string", "yolo2_hip_plan_source",
    "yolo2_hip_set_option", "yolo2_hip_options_string", "yolo2_hip_set_plan_cache", "yolo2_hip_plan_cache_info", "yolo2_hip_plan_cache_check",
    "yolo2_hip_i16_plan_check", "yolo2_hip_ks_scratch_bytes",
    "yolo2_hip_run_batch_f32tol", "yolo2_hip_run_batch_f32tol_host", "yolo2_hip_f32tol_layer_kernel", "yolo2_hip_num_lanes_f32tol",
]


def letterbox_u8(image_hwc: np.ndarray, net_w: int = 416, net_h: int = 416) -> np.ndarray:
    """uint8 [h][w][3] (or [h][w]) host image -> float [3][net_h][net_w] frame, letterboxed on the GPU."""
    im = np.ascontiguousarray(image_hwc, dtype=np.uint8)
    ch = 1 if im.ndim == 2 else im.shape[2]
    L = lib()
    src, dst = C.c_uint64(0), C.c_uint64(0)
    out = np.empty((3, net_h, net_w), dtype=np.float32)
    check(L.yolo2_hip_alloc(im.nbytes, C.byref(src)), "alloc")
    check(L.yolo2_hip_alloc(out.nbytes, C.byref(dst)), "alloc")
    try:
        check(L.yolo2_hip_memcpy_h2d(src, im.ctypes.data_as(C.c_void_p), im.nbytes), "h2d")
        check(L.yolo2_hip_letterbox_u8(src, im.shape[1], im.shape[0], ch, dst, net_w, net_h, None), "yolo2_hip_letterbox_u8")
        check(L.yolo2_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), dst, out.nbytes), "d2h")
    finally:
        L.yolo2_hip_free(src)
        L.yolo2_hip_free(dst)
    return out


class Yolo2HipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(PKG_DIR, "csrc", f) for f in os.listdir(os.path.join(PKG_DIR, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "yolo2_hip.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.run(["make", "-s", "-C", PKG_DIR], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Yolo2HipError(f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR}` "
                            "(the GPU path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    u64, i32, u32, vp = C.c_uint64, C.c_int, C.c_uint32, C.c_void_p
    L.yolo2_hip_last_error.restype = C.c_char_p
    L.yolo2_execute_conv_layer.argtypes = [u64] * 4 + [i32] * 23 + [u32]
    L.yolo2_execute_maxpool_layer.argtypes = [u64] * 2 + [i32] * 14 + [u32]
    L.yolo2_execute_conv_layer_f32.argtypes = [u64] * 4 + [i32] * 10 + [u32]
    L.yolo2_hip_alloc.argtypes = [C.c_size_t, C.POINTER(u64)]
    L.yolo2_hip_free.argtypes = [u64]
    L.yolo2_hip_memcpy_h2d.argtypes = [u64, vp, C.c_size_t]
    L.yolo2_hip_memcpy_d2h.argtypes = [vp, u64, C.c_size_t]
    L.yolo2_hip_memset.argtypes = [u64, i32, C.c_size_t]
    L.yolo2_hip_create.argtypes = [i32, C.POINTER(vp)]
    L.yolo2_hip_destroy.argtypes = [vp]
    L.yolo2_hip_load_weights_int16.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, i32, vp, i32, vp, i32]
    L.yolo2_hip_load_weights_int16_dev.argtypes = [vp, u64, C.c_size_t, u64, C.c_size_t, vp, i32, vp, i32, vp, i32]
    L.yolo2_hip_layer_path.argtypes = [vp, i32]
    L.yolo2_hip_num_lanes.argtypes = [vp]
    L.yolo2_hip_layer_pool_fused.argtypes = [vp, i32]
    L.yolo2_hip_run_frames_int16.argtypes = [vp, vp, i32, i32, vp, C.POINTER(i32)]
    L.yolo2_hip_layer_path_counts.argtypes = [vp, i32, C.POINTER(i32)]
    L.yolo2_hip_set_batch.argtypes = [vp, i32]
    L.yolo2_hip_run_batch_int16.argtypes = [vp, u64, i32, u64, C.POINTER(i32), vp]
    L.yolo2_hip_run_batch_int16_host.argtypes = [vp, vp, i32, vp, C.POINTER(i32)]
    L.yolo2_hip_debug_layer_output.argtypes = [vp, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.yolo2_hip_set_profiling.argtypes = [vp, i32]
    L.yolo2_hip_layer_times_ms.argtypes = [vp, vp]
    L.yolo2_hip_conv_launch_info.argtypes = [vp, i32] + [C.POINTER(i32)] * 5
    L.yolo2_strip_int16_layer_pad.argtypes = [vp, C.c_size_t, vp, i32, vp]
    L.yolo2_strip_int16_layer_pad.restype = C.c_long
    L.yolo2_hip_load_weights_fp32.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    L.yolo2_hip_run_batch_fp16.argtypes = [vp, u64, i32, u64, vp]
    L.yolo2_hip_run_batch_fp16_host.argtypes = [vp, vp, i32, vp]
    L.yolo2_hip_run_frame_fp32_host.argtypes = [vp, vp, vp]
    L.yolo2_hip_run_batch_fp32.argtypes = [vp, u64, i32, u64, vp]
    L.yolo2_hip_run_batch_fp32_host.argtypes = [vp, vp, i32, vp]
    L.yolo2_hip_letterbox_u8.argtypes = [u64, i32, i32, i32, u64, i32, i32, vp]
    L.yolo2_hip_run_images_u8_host.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
    L.memory_get_phys_addr.restype = u64
    pi32 = C.POINTER(i32)
    L.yolo2_hip_load_weights_fp32_dev.argtypes = [vp, u64, C.c_size_t, u64, C.c_size_t]
    L.yolo2_hip_postprocess_int16.argtypes = [vp, u64, i32, i32, vp, vp, C.c_float, C.c_float, vp, i32, vp, vp, vp, vp, vp]
    L.yolo2_hip_postprocess_f32.argtypes = [vp, u64, i32, vp, vp, C.c_float, C.c_float, vp, i32, vp, vp, vp, vp, vp]
    L.yolo2_hip_shard_range.argtypes = [i32, i32, i32, pi32, pi32]
    # (an older build loaded through YOLO2_HIP_LIB for an A/B run may lack the newer entries: their signatures are set when present)
    def sig(name, argtypes, restype=None):
        f = getattr(L, name, None)
        if f is not None:
            f.argtypes = argtypes
            if restype is not None:
                f.restype = restype
    sig("yolo2_hip_set_option", [vp, C.c_char_p, C.c_char_p])
    sig("yolo2_hip_options_string", [vp, C.c_char_p, i32])
    sig("yolo2_hip_set_plan_cache", [vp, C.c_char_p])
    sig("yolo2_hip_plan_cache_info", [vp, C.POINTER(u64), pi32, pi32, pi32])
    sig("yolo2_hip_plan_cache_check", [C.c_char_p, u64, pi32])
    sig("yolo2_hip_i16_plan_check", [i32, i32, i32, C.c_size_t])
    sig("yolo2_hip_run_batch_f32tol", [vp, u64, i32, u64, vp])
    sig("yolo2_hip_run_batch_f32tol_host", [vp, vp, i32, vp])
    sig("yolo2_hip_f32tol_layer_kernel", [vp, i32], C.c_char_p)
    sig("yolo2_hip_num_lanes_f32tol", [vp])
    sig("yolo2_hip_ks_scratch_bytes", [vp], C.c_size_t)
    L.yolo2_hip_multi_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.yolo2_hip_multi_destroy.argtypes = [vp]
    L.yolo2_hip_multi_num_devices.argtypes = [vp]
    L.yolo2_hip_multi_uses_rccl.argtypes = [vp]
    L.yolo2_hip_multi_ctx.argtypes = [vp, i32]
    L.yolo2_hip_multi_ctx.restype = vp
    L.yolo2_hip_multi_load_weights_int16.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, i32, vp, i32, vp, i32]
    L.yolo2_hip_multi_load_weights_fp32.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    L.yolo2_hip_multi_run_frames_int16.argtypes = [vp, vp, i32, i32, vp, pi32]
    L.yolo2_hip_multi_run_images_u8_host.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, pi32]
    L.yolo2_hip_rccl_unique_id.argtypes = [vp]
    L.yolo2_hip_rccl_init_rank.argtypes = [vp, i32, vp, i32, i32]
    L.yolo2_hip_rccl_finalize.argtypes = [vp]
    L.yolo2_hip_load_weights_int16_bcast.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, i32, vp, i32, vp, i32, i32]
    L.yolo2_hip_load_weights_fp32_bcast.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, i32]
    L.yolo2_hip_run_images_u8_dets.argtypes = [vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, i32, vp, i32, vp, C.POINTER(i32)]
    L.yolo2_hip_multi_run_images_u8_dets.argtypes = [vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, i32, vp, i32, vp, C.POINTER(i32)]
    L.yolo2_hip_set_fp16_lanes.argtypes = [vp, i32]
    L.yolo2_hip_conv_plan_string.argtypes = [vp, i32, C.c_char_p, i32]
    L.yolo2_hip_plan_source.argtypes = [vp]
    L.yolo2_hip_rccl_info.argtypes = [vp, vp]
    L.yolo2_hip_multi_rccl_info.argtypes = [vp, vp]
    L.yolo2_hip_ctx_device.argtypes = [vp]
    L.yolo2_hip_alloc_on.argtypes = [vp, C.c_size_t, C.POINTER(u64)]
    L.yolo2_hip_fp16_layer_kernel.argtypes = [vp, i32]
    L.yolo2_hip_fp16_layer_kernel.restype = C.c_char_p
    L.yolo2_hip_f16_store_check.argtypes = [i32] * 12
    L.yolo2_get_status.restype = u32
    L.yolo2_read_reg.restype = u32
    L.yolo2_read_reg.argtypes = [u32]
    L.yolo2_write_reg.argtypes = [u32, u32]
    L.yolo2_hip_driver_calls.restype = C.c_long
    L.yolo2_set_q_values.argtypes = [i32] * 4
    L.dma_buffer_alloc.argtypes = [C.c_size_t, vp]
    L.dma_buffer_free.argtypes = [vp]
    L.dma_buffer_get_phys.restype = u64
    L.dma_buffer_get_phys.argtypes = [vp, C.c_size_t]
    L.dma_buffer_sync_for_device.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.dma_buffer_sync_for_cpu.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.memory_get_phys_addr.argtypes = [vp]
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != YOLO2_SUCCESS:
        raise Yolo2HipError(f"{what} failed with status {rc}: {lib().yolo2_hip_last_error().decode()}")


# ------------------------------------------------------------------ tier 1 helpers (tests)

class DevBuf:
    """HBM buffer holding a numpy array (the analogue of a udmabuf + its physical address)."""

    def __init__(self, arr: np.ndarray = None, nbytes: int = None, fill: int = 0):
        self.nbytes = arr.nbytes if arr is not None else nbytes
        a = C.c_uint64(0)
        check(lib().yolo2_hip_alloc(max(self.nbytes, 16), C.byref(a)), "yolo2_hip_alloc")
        self.addr = a.value
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            check(lib().yolo2_hip_memcpy_h2d(self.addr, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
        else:
            check(lib().yolo2_hip_memset(self.addr, fill, self.nbytes), "memset")

    def get(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(lib().yolo2_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.addr, out.nbytes), "d2h")
        return out

    def free(self):
        if self.addr:
            lib().yolo2_hip_free(self.addr)
            self.addr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def w8(w):
    return (w + 7) // 8 * 8


def conv_layer_i16(x, w_reorg, bias, C_, N, K, stride, W, H, pad, leaky, Qw, Qa_in, Qa_out, Qb, fill=0):
    """Calls yolo2_execute_conv_layer the way linux_app/src/yolo2_inference.c:371-381 does."""
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    TR = min(min((27 - K) // stride + 1, 13), OH)
    TC = min(min((27 - K) // stride + 1, 13), OW)
    TM, TN = min(N, 32), min(C_, 4)
    mLoops = -(-N // TM)
    bx, bw, bb = DevBuf(x), DevBuf(w_reorg), DevBuf(bias)
    out0 = np.full((N, OH, w8(OW)), fill, dtype=np.int16)
    by = DevBuf(out0)
    rc = lib().yolo2_execute_conv_layer(bx.addr, by.addr, bw.addr, bb.addr, C_, N, K, stride, W, H, OW, OH, pad,
                                        int(leaky), 0, TM, TN, TR, TC, (mLoops + 1) * TM, mLoops * TM,
                                        (mLoops + 1) * TM, 0, Qw, Qa_in, Qa_out, Qb, 60000)
    check(rc, "yolo2_execute_conv_layer")
    y = by.get(np.int16, (N, OH, w8(OW)))
    for b in (bx, bw, bb, by):
        b.free()
    return y


def conv_layer_f32(x, w_reorg, bias, C_, N, K, stride, W, H, pad, leaky):
    OW = (W - K + 2 * pad) // stride + 1
    OH = (H - K + 2 * pad) // stride + 1
    bx, bw, bb = DevBuf(x), DevBuf(w_reorg), DevBuf(bias)
    by = DevBuf(np.zeros((N, OH, w8(OW)), dtype=np.float32))
    check(lib().yolo2_execute_conv_layer_f32(bx.addr, by.addr, bw.addr, bb.addr, C_, N, K, stride, W, H, OW, OH,
                                             pad, int(leaky), 60000), "yolo2_execute_conv_layer_f32")
    y = by.get(np.float32, (N, OH, w8(OW)))
    for b in (bx, bw, bb, by):
        b.free()
    return y


def maxpool_layer_i16(x, C_, W, H, K=2, stride=2):
    OW, OH = W // stride, H // stride
    TR = min((27 - K) // stride + 1, 13, OH)
    TC = min((27 - K) // stride + 1, 13, OW)
    TM = min(4, C_)
    mLoops = -(-C_ // TM)
    bx = DevBuf(x)
    by = DevBuf(np.zeros((C_, OH, w8(OW)), dtype=np.int16))
    check(lib().yolo2_execute_maxpool_layer(bx.addr, by.addr, C_, K, stride, W, H, OW, OH, 1, TM, TR, TC,
                                            (mLoops + 2) * TM, mLoops * TM, (mLoops + 1) * TM, 60000),
          "yolo2_execute_maxpool_layer")
    y = by.get(np.int16, (C_, OH, w8(OW)))
    bx.free(); by.free()
    return y


# ------------------------------------------------------------------ tier 2: whole network

class Yolo2Hip:
    """One accelerator context on one GPU (yolo2_hip_ctx)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p(0)
        check(lib().yolo2_hip_create(device, C.byref(self._h)), "yolo2_hip_create")
        self.device = device
        self.final_q = None

    def close(self):
        if self._h:
            lib().yolo2_hip_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, weights_i16, bias_i16, weight_q, bias_q, act_q):
        w = np.ascontiguousarray(weights_i16, dtype=np.int16)
        b = np.ascontiguousarray(bias_i16, dtype=np.int16)
        wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (weight_q, bias_q, act_q))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().yolo2_hip_load_weights_int16(self._h, vp(w), w.size, vp(b), b.size, vp(wq), wq.size,
                                                 vp(bq), bq.size, vp(aq), aq.size), "yolo2_hip_load_weights_int16")

    def load_weights_dev(self, w_ptr, n_w, b_ptr, n_b, weight_q, bias_q, act_q):
        wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (weight_q, bias_q, act_q))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().yolo2_hip_load_weights_int16_dev(self._h, w_ptr, n_w, b_ptr, n_b, vp(wq), wq.size, vp(bq),
                                                     bq.size, vp(aq), aq.size), "yolo2_hip_load_weights_int16_dev")

    def load_model(self, model):
        self.load_weights(model.weights_i16(), model.bias_i16(), model.weight_q, model.bias_q, model.act_q)

    def layer_paths(self):
        return [lib().yolo2_hip_layer_path(self._h, o) for o in range(len(net.CONVS))]

    def layer_path_counts(self):
        out = []
        for o in range(len(net.CONVS)):
            v = (C.c_int * 5)()
            check(lib().yolo2_hip_layer_path_counts(self._h, o, v), "yolo2_hip_layer_path_counts")
            out.append(list(v))
        return out

    def set_batch(self, batch: int):
        check(lib().yolo2_hip_set_batch(self._h, batch), "yolo2_hip_set_batch")

    def run_batch_ptr(self, frames_ptr: int, batch: int, region_ptr: int, stream: int = 0) -> int:
        """Device pointers in/out; enqueues on `stream` and returns without synchronising."""
        q = C.c_int(0)
        check(lib().yolo2_hip_run_batch_int16(self._h, frames_ptr, batch, region_ptr, C.byref(q),
                                              C.c_void_p(stream)), "yolo2_hip_run_batch_int16")
        self.final_q = q.value
        return q.value

    def run_batch_host(self, frames: np.ndarray):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        B = frames.shape[0]
        region = np.empty((B, 425, 13, 13), dtype=np.int16)
        q = C.c_int(0)
        check(lib().yolo2_hip_run_batch_int16_host(self._h, frames.ctypes.data_as(C.c_void_p), B,
                                                   region.ctypes.data_as(C.c_void_p), C.byref(q)),
              "yolo2_hip_run_batch_int16_host")
        self.final_q = q.value
        return region, q.value

    def run_images_host(self, images, batch: int = 64):
        """images: list of uint8 arrays [h][w][3] (or [h][w] grey) of arbitrary sizes -> region tensors.
        Bytes cross PCIe, letterboxing runs on the GPU (yolo2_hip_run_images_u8_host)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        ch = 1 if imgs[0].ndim == 2 else imgs[0].shape[2]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        region = np.empty((n, 425, 13, 13), dtype=np.int16)
        q = C.c_int(0)
        check(lib().yolo2_hip_run_images_u8_host(self._h, ptrs, ws, hs, ch, n, batch, region.ctypes.data_as(C.c_void_p), C.byref(q)),
              "yolo2_hip_run_images_u8_host")
        self.final_q = q.value
        return region, q.value

    # ---- fp16 MFMA path
    def load_weights_fp32(self, weights_f32, bias_f32):
        w = np.ascontiguousarray(weights_f32, dtype=np.float32)
        b = np.ascontiguousarray(bias_f32, dtype=np.float32)
        check(lib().yolo2_hip_load_weights_fp32(self._h, w.ctypes.data_as(C.c_void_p), w.size,
                                                b.ctypes.data_as(C.c_void_p), b.size), "yolo2_hip_load_weights_fp32")

    def run_frame_fp32_host(self, frame: np.ndarray) -> np.ndarray:
        """One frame through the exact fp32 network (reference arithmetic); needs load_weights_fp32."""
        f = np.ascontiguousarray(frame, dtype=np.float32).reshape(3, 416, 416)
        region = np.empty((425, 13, 13), dtype=np.float32)
        check(lib().yolo2_hip_run_frame_fp32_host(self._h, f.ctypes.data_as(C.c_void_p), region.ctypes.data_as(C.c_void_p)),
              "yolo2_hip_run_frame_fp32_host")
        return region

    def run_batch_fp32_ptr(self, frames_ptr: int, batch: int, region_ptr: int, stream: int = 0):
        check(lib().yolo2_hip_run_batch_fp32(self._h, frames_ptr, batch, region_ptr, C.c_void_p(stream)), "yolo2_hip_run_batch_fp32")

    def run_batch_fp32_host(self, frames: np.ndarray) -> np.ndarray:
        """The tiled exact fp32 pass (bit-identical to the reference's fp32 path) on a batch of host frames."""
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        B = frames.shape[0]
        region = np.empty((B, 425, 13, 13), dtype=np.float32)
        check(lib().yolo2_hip_run_batch_fp32_host(self._h, frames.ctypes.data_as(C.c_void_p), B, region.ctypes.data_as(C.c_void_p)),
              "yolo2_hip_run_batch_fp32_host")
        return region

    def run_batch_fp16_ptr(self, frames_ptr: int, batch: int, region_ptr: int, stream: int = 0):
        check(lib().yolo2_hip_run_batch_fp16(self._h, frames_ptr, batch, region_ptr, C.c_void_p(stream)),
              "yolo2_hip_run_batch_fp16")

    def run_batch_fp16_host(self, frames: np.ndarray) -> np.ndarray:
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        B = frames.shape[0]
        region = np.empty((B, 425, 13, 13), dtype=np.float32)
        check(lib().yolo2_hip_run_batch_fp16_host(self._h, frames.ctypes.data_as(C.c_void_p), B,
                                                  region.ctypes.data_as(C.c_void_p)), "yolo2_hip_run_batch_fp16_host")
        return region

    def run_frames(self, frames: np.ndarray, batch: int):
        """Streaming entry: any number of host frames, chunks of `batch`, copies overlapped with compute."""
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        n = frames.shape[0]
        region = np.empty((n, 425, 13, 13), dtype=np.int16)
        q = C.c_int(0)
        check(lib().yolo2_hip_run_frames_int16(self._h, frames.ctypes.data_as(C.c_void_p), n, batch,
                                               region.ctypes.data_as(C.c_void_p), C.byref(q)), "yolo2_hip_run_frames_int16")
        return region, q.value

    def debug_layer_output(self, layer_idx: int, frame: int = 0) -> np.ndarray:
        if layer_idx < 0:      # the quantised network input
            shape = (3, 416, 416)
        else:
            l = net.LAYERS[layer_idx]
            shape = (l.out_c, l.out_h, w8(l.out_w))
        out = np.empty(shape, dtype=np.int16)
        n = C.c_size_t(0)
        check(lib().yolo2_hip_debug_layer_output(self._h, layer_idx, frame, out.ctypes.data_as(C.c_void_p),
                                                 out.size, C.byref(n)), "yolo2_hip_debug_layer_output")
        assert n.value == out.size
        return out

    def num_lanes(self) -> int:
        return lib().yolo2_hip_num_lanes(self._h)

    def pool_fused_layers(self):
        """Conv layers that run fused with the max pool after them (their full-resolution tensor is not written)."""
        return [i for i in range(32) if lib().yolo2_hip_layer_pool_fused(self._h, i)]

    def set_profiling(self, on: bool):
        check(lib().yolo2_hip_set_profiling(self._h, int(on)), "yolo2_hip_set_profiling")

    def layer_times_ms(self) -> np.ndarray:
        ms = np.zeros(32, dtype=np.float32)
        check(lib().yolo2_hip_layer_times_ms(self._h, ms.ctypes.data_as(C.c_void_p)), "yolo2_hip_layer_times_ms")
        return ms

    # ---- split-fp16 ("fp32 fast"): fp32 accuracy on the matrix cores
    def run_batch_f32tol_host(self, frames: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(frames, dtype=np.float32)
        batch = f.shape[0]
        region = np.empty((batch, 425, 13, 13), dtype=np.float32)
        check(lib().yolo2_hip_run_batch_f32tol_host(self._h, f.ctypes.data_as(C.c_void_p), batch, region.ctypes.data_as(C.c_void_p)),
              "yolo2_hip_run_batch_f32tol_host")
        return region

    def run_batch_f32tol_ptr(self, frames_ptr: int, batch: int, region_ptr: int, stream: int = 0):
        check(lib().yolo2_hip_run_batch_f32tol(self._h, frames_ptr, batch, region_ptr, C.c_void_p(stream)), "yolo2_hip_run_batch_f32tol")

    def f32tol_layer_kernel(self, layer_idx: int) -> str:
        return lib().yolo2_hip_f32tol_layer_kernel(self._h, layer_idx).decode()

    def num_lanes_f32tol(self) -> int:
        return int(lib().yolo2_hip_num_lanes_f32tol(self._h))

    def set_fp16_lanes(self, lanes: int):
        check(lib().yolo2_hip_set_fp16_lanes(self._h, lanes), "yolo2_hip_set_fp16_lanes")

    def num_lanes_fp16(self) -> int:
        return int(lib().yolo2_hip_num_lanes_fp16(self._h))

    def plan_source(self) -> str:
        return {0: "none", 1: "plan table", 2: "autotuned in this process", 3: "static defaults", 4: "forced by a test hook",
                5: "weight cache"}[int(lib().yolo2_hip_plan_source(self._h))]

    def set_option(self, name: str, value=None):
        """One named option of the context (include/yolo2_hip.h "options"); None restores the default."""
        check(lib().yolo2_hip_set_option(self._h, name.encode(), None if value is None else str(value).encode()), f"yolo2_hip_set_option({name})")

    def options(self) -> str:
        if not hasattr(lib(), "yolo2_hip_options_string"):
            return "(library without option sets)"
        buf = C.create_string_buffer(1024)
        check(lib().yolo2_hip_options_string(self._h, buf, 1024), "yolo2_hip_options_string")
        return buf.value.decode()

    def ks_scratch_bytes(self) -> int:
        return int(lib().yolo2_hip_ks_scratch_bytes(self._h))

    def set_plan_cache(self, path):
        check(lib().yolo2_hip_set_plan_cache(self._h, None if path is None else str(path).encode()), "yolo2_hip_set_plan_cache")

    def plan_cache_info(self):
        h, used, n, nb = C.c_uint64(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().yolo2_hip_plan_cache_info(self._h, C.byref(h), C.byref(used), C.byref(n), C.byref(nb)), "yolo2_hip_plan_cache_info")
        return {"hash": h.value, "bounds_from_file": bool(used.value), "lines": n.value, "batches": nb.value}

    def conv_plan(self, ord_: int) -> str:
        buf = C.create_string_buffer(256)
        check(lib().yolo2_hip_conv_plan_string(self._h, ord_, buf, 256), "yolo2_hip_conv_plan_string")
        return buf.value.decode()

    def conv_launch_info(self, ord_: int):
        v = [C.c_int(0) for _ in range(5)]
        check(lib().yolo2_hip_conv_launch_info(self._h, ord_, *[C.byref(x) for x in v]), "conv_launch_info")
        return dict(zip(("grid_x", "grid_y", "block", "lds_bytes", "pixels_per_lane"), (x.value for x in v)))


# ------------------------------------------------------------------ post-processing on the GPU

DET_DTYPE = np.dtype([("frame", np.int32), ("det", np.int32), ("cls", np.int32), ("prob", np.float32),
                      ("x", np.float32), ("y", np.float32), ("w", np.float32), ("h", np.float32)])


def postprocess(ctx, region_ptr: int, batch: int, im_w, im_h, thresh: float, nms: float, final_q: int = None, cap: int = 256,
                want_rows: bool = False, want_proc: bool = False, stream: int = 0):
    """yolo2_hip_postprocess_int16 (final_q given) / _f32 on a DEVICE region tensor.  Returns a dict with
    dets (structured array of the records kept), counts, and optionally rows [B][845][85], totals, proc."""
    iw = np.ascontiguousarray(im_w, dtype=np.int32)
    ih = np.ascontiguousarray(im_h, dtype=np.int32)
    assert iw.size == batch and ih.size == batch
    dets = np.zeros((batch, cap), dtype=DET_DTYPE)
    counts = np.zeros(batch, dtype=np.int32)
    rows = np.zeros((batch, 845, 85), dtype=np.float32) if want_rows else None
    totals = np.zeros(batch, dtype=np.int32)
    proc = np.zeros((batch, 425 * 169), dtype=np.float32) if want_proc else None
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    if final_q is not None:
        rc = lib().yolo2_hip_postprocess_int16(ctx._h, region_ptr, batch, int(final_q), vp(iw), vp(ih), thresh, nms, vp(dets), cap,
                                               vp(counts), vp(rows), vp(totals), vp(proc), C.c_void_p(stream))
    else:
        rc = lib().yolo2_hip_postprocess_f32(ctx._h, region_ptr, batch, vp(iw), vp(ih), thresh, nms, vp(dets), cap, vp(counts),
                                             vp(rows), vp(totals), vp(proc), C.c_void_p(stream))
    check(rc, "yolo2_hip_postprocess")
    return {"dets": [dets[f, :min(int(counts[f]), cap)] for f in range(batch)], "counts": counts, "rows": rows, "totals": totals, "proc": proc}


DETS_BEST_CLASS = 1


def run_images_dets(handle, images, batch: int, thresh: float, nms: float, cap: int = 845, best_class: bool = True, multi: bool = False):
    """yolo2_hip_run_images_u8_dets / yolo2_hip_multi_run_images_u8_dets: host images (uint8 [h][w][3]) -> per-frame detection
    records; letterbox, network and the tail run on the device(s), the region tensor never leaves HBM."""
    imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
    n = len(imgs)
    ch = 1 if imgs[0].ndim == 2 else imgs[0].shape[2]
    ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
    hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
    dets = np.zeros((n, cap), dtype=DET_DTYPE)
    counts = np.zeros(n, dtype=np.int32)
    q = C.c_int(0)
    fn = lib().yolo2_hip_multi_run_images_u8_dets if multi else lib().yolo2_hip_run_images_u8_dets
    check(fn(handle, ptrs, ws, hs, ch, n, batch, thresh, nms, DETS_BEST_CLASS if best_class else 0, dets.ctypes.data_as(C.c_void_p), cap,
             counts.ctypes.data_as(C.c_void_p), C.byref(q)), "yolo2_hip_run_images_u8_dets")
    return {"dets": [dets[f, :min(int(counts[f]), cap)] for f in range(n)], "counts": counts, "final_q": q.value}


# ------------------------------------------------------------------ more than one GPU

def shard_range(total: int, rank: int, world: int):
    lo, hi = C.c_int(0), C.c_int(0)
    check(lib().yolo2_hip_shard_range(total, rank, world, C.byref(lo), C.byref(hi)), "yolo2_hip_shard_range")
    return lo.value, hi.value


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0): 128 bytes the launcher hands to every rank."""
    buf = (C.c_char * 128)()
    check(lib().yolo2_hip_rccl_unique_id(buf), "yolo2_hip_rccl_unique_id")
    return bytes(buf)


class RcclInfo(C.Structure):
    """yolo2_hip_rccl_info_t"""
    _fields_ = [("nranks", C.c_int), ("rank", C.c_int), ("device", C.c_int), ("version", C.c_int), ("bcasts", C.c_int),
                ("last_bcast_ms", C.c_double), ("last_bcast_bytes", C.c_uint64), ("lib_path", C.c_char * 256)]

    def as_dict(self):
        v = self.version
        return {"nranks": self.nranks, "rank": self.rank, "device": self.device, "version": v,
                "version_str": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else str(v),
                "bcasts": self.bcasts, "bcast_ms": self.last_bcast_ms, "bytes": int(self.last_bcast_bytes),
                "lib_path": self.lib_path.decode()}


class _BcastMixin:
    def rccl_init_rank(self, id128: bytes, nranks: int, rank: int):
        assert len(id128) == 128
        check(lib().yolo2_hip_rccl_init_rank(self._h, self.device, C.c_char_p(id128), nranks, rank), "yolo2_hip_rccl_init_rank")
        self._in_comm = True

    def rccl_info(self) -> dict:
        """What the library's communicator reports about itself (yolo2_hip_rccl_info)."""
        info = RcclInfo()
        check(lib().yolo2_hip_rccl_info(self._h, C.byref(info)), "yolo2_hip_rccl_info")
        return info.as_dict()

    def fp16_layer_kernels(self):
        """Kernel name per layer of the fp16 launch table (after the first run at the current batch)."""
        return {i: lib().yolo2_hip_fp16_layer_kernel(self._h, i).decode() for i in range(32) if lib().yolo2_hip_fp16_layer_kernel(self._h, i)}

    def rccl_finalize(self):
        if getattr(self, "_in_comm", False):
            lib().yolo2_hip_rccl_finalize(self._h)
            self._in_comm = False

    def load_model_bcast(self, model, root: int = 0):
        """All ranks call this; `model` is the SynthModel on the root rank and None elsewhere.  One ncclBroadcast per blob
        (the library's own RCCL communicator), then every rank loads its device copy."""
        if model is not None:
            w = np.ascontiguousarray(model.weights_i16(), dtype=np.int16)
            b = np.ascontiguousarray(model.bias_i16(), dtype=np.int16)
            wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (model.weight_q, model.bias_q, model.act_q))
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            rc = lib().yolo2_hip_load_weights_int16_bcast(self._h, vp(w), w.size, vp(b), b.size, vp(wq), wq.size, vp(bq), bq.size,
                                                          vp(aq), aq.size, root)
        else:
            rc = lib().yolo2_hip_load_weights_int16_bcast(self._h, None, 0, None, 0, None, 0, None, 0, None, 0, root)
        check(rc, "yolo2_hip_load_weights_int16_bcast")

    def load_model_fp32_bcast(self, model, root: int = 0):
        if model is not None:
            w = np.ascontiguousarray(model.weights_f32(), dtype=np.float32)
            b = np.ascontiguousarray(model.bias_f32(), dtype=np.float32)
            rc = lib().yolo2_hip_load_weights_fp32_bcast(self._h, w.ctypes.data_as(C.c_void_p), w.size, b.ctypes.data_as(C.c_void_p),
                                                         b.size, root)
        else:
            rc = lib().yolo2_hip_load_weights_fp32_bcast(self._h, None, 0, None, 0, root)
        check(rc, "yolo2_hip_load_weights_fp32_bcast")


for _name, _fn in list(vars(_BcastMixin).items()):
    if not _name.startswith("__"):
        setattr(Yolo2Hip, _name, _fn)


class Yolo2HipMulti:
    """One process, several devices (yolo2_hip_multi): contiguous frame shards, weights broadcast once."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        self._m = C.c_void_p(0)
        check(lib().yolo2_hip_multi_create(devs, len(devices), C.byref(self._m)), "yolo2_hip_multi_create")
        self.devices = list(devices)

    def close(self):
        if self._m:
            lib().yolo2_hip_multi_destroy(self._m)
            self._m = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def uses_rccl(self) -> bool:
        return bool(lib().yolo2_hip_multi_uses_rccl(self._m))

    def load_model(self, model):
        w = np.ascontiguousarray(model.weights_i16(), dtype=np.int16)
        b = np.ascontiguousarray(model.bias_i16(), dtype=np.int16)
        wq, bq, aq = (np.ascontiguousarray(a, dtype=np.int32) for a in (model.weight_q, model.bias_q, model.act_q))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().yolo2_hip_multi_load_weights_int16(self._m, vp(w), w.size, vp(b), b.size, vp(wq), wq.size, vp(bq), bq.size,
                                                       vp(aq), aq.size), "yolo2_hip_multi_load_weights_int16")

    def run_frames(self, frames: np.ndarray, batch_per_device: int):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        n = frames.shape[0]
        region = np.empty((n, 425, 13, 13), dtype=np.int16)
        q = C.c_int(0)
        check(lib().yolo2_hip_multi_run_frames_int16(self._m, frames.ctypes.data_as(C.c_void_p), n, batch_per_device,
                                                     region.ctypes.data_as(C.c_void_p), C.byref(q)), "yolo2_hip_multi_run_frames_int16")
        return region, q.value

    def run_images(self, images, batch_per_device: int):
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        ch = 1 if imgs[0].ndim == 2 else imgs[0].shape[2]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        region = np.empty((n, 425, 13, 13), dtype=np.int16)
        q = C.c_int(0)
        check(lib().yolo2_hip_multi_run_images_u8_host(self._m, ptrs, ws, hs, ch, n, batch_per_device, region.ctypes.data_as(C.c_void_p),
                                                       C.byref(q)), "yolo2_hip_multi_run_images_u8_host")
        return region, q.value
