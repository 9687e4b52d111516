// yolo2_driver.hip -- tier 1 of include/yolo2_hip.h: the reference's userspace accelerator driver, call for call
// (linux_app/src/yolo2_accel_linux.c:419-575 per-layer calls, :232-258 register file, :266-381 wait-for-idle;
// linux_app/src/dma_buffer_manager.c buffers).  This file holds the driver STATE, argument validation, the shadow register
// file, the lock and the timeout semantics; the device work of a layer call is enqueued by the int16 / fp32 translation units
// (y2_drv_conv_i16, y2_drv_pool_i16, y2_drv_conv_f32 in y2_internal.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "y2_internal.hpp"

namespace {
struct DriverState {
    std::mutex mu;
    int device = 0;
    bool inited = false;
    int qw = 0, qa_in = 0, qa_out = 0, qb = 0;
    int last_path = -1;   // arithmetic form of the most recent yolo2_execute_conv_layer (-1: generic reference-layout kernel)
    int last_rc = YOLO2_SUCCESS;   // status of the most recent layer call, register-level starts included (R_HIP_STATUS)
    struct HostBuf {
        char *host;
        char *dev;
        size_t size;
    };
    std::vector<HostBuf> hostbufs;
    uint32_t regs[1024] = {0};   // shadow of the HLS IP's 4 KiB AXI-Lite register file (yolo2_config.h:36-71 offsets)
    long calls = 0;              // per-layer calls served since yolo2_accel_init
};
DriverState g_drv;

// register offsets of the HLS IP (linux_app/include/yolo2_config.h:36-71)
enum : uint32_t {
    R_AP_CTRL = 0x00, R_INPUT = 0x10, R_OUTPUT = 0x1c, R_WEIGHT = 0x28, R_BETA = 0x34, R_IFM = 0x40, R_OFM = 0x48,
    R_KSIZE = 0x50, R_KSTRIDE = 0x58, R_IN_W = 0x60, R_IN_H = 0x68, R_OUT_W = 0x70, R_OUT_H = 0x78, R_PAD = 0x80,
    R_ISNL = 0x88, R_ISBN = 0x90, R_TM = 0x98, R_TN = 0xa0, R_TR = 0xa8, R_TC = 0xb0, R_OFM_BOUND = 0xb8,
    R_MLOOPS = 0xc0, R_MLOOPS_A1 = 0xc8, R_LTYPE = 0xd0,
    R_HIP_STATUS = 0xf0,   // not an HLS register (the IP's map ends at 0xd0): status code of the last layer call, read-only
    AP_START = 1u << 0, AP_DONE = 1u << 1, AP_IDLE = 1u << 2, AP_READY = 1u << 3,
};
inline void reg_set64(uint32_t off, uint64_t v) { g_drv.regs[off / 4] = (uint32_t)v; g_drv.regs[off / 4 + 1] = (uint32_t)(v >> 32); }
inline uint64_t reg_get64(uint32_t off) { return (uint64_t)g_drv.regs[off / 4] | ((uint64_t)g_drv.regs[off / 4 + 1] << 32); }

int sync_with_timeout(hipStream_t st, uint32_t timeout_ms)
{
    if (timeout_ms == 0) {  // 0 = wait forever (yolo2_accel_linux.h:55-61)
        HIP_TRY(hipStreamSynchronize(st), YOLO2_ERROR);
        return YOLO2_SUCCESS;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return YOLO2_SUCCESS;
        if (e != hipErrorNotReady) return fail(YOLO2_ERROR, "stream error: %s", hipGetErrorString(e));
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (us > (long long)timeout_ms * 1000) return fail(YOLO2_TIMEOUT, "layer did not finish within %u ms", timeout_ms);
        // a layer call takes tens of microseconds: poll hot for the first 200 us, then give the core away between polls
        // (the reference's wait_for_idle sleeps between register reads too, yolo2_accel_linux.c:300-340)
        if (us > 200) std::this_thread::sleep_for(std::chrono::microseconds(us > 5000 ? 200 : 20));
    }
}

// yolo2_accel_linux.c:383-414, which mirrors the HLS asserts (yolo2_accel.cpp:75-87)
bool validate_conv_params(int ifm, int ofm, int k, int s, int iw, int ih, int ow, int oh, int pad, int tm, int tn,
                          int tr, int tc)
{
    if (ifm <= 0 || ifm > 2048) return false;
    if (ofm <= 0 || ofm > 2048) return false;
    if (k <= 0 || k > 3) return false;
    if (s <= 0 || s > 2) return false;
    if (iw <= 0 || iw > 1024 || ih <= 0 || ih > 1024) return false;
    if (ow <= 0 || ow > 1024 || oh <= 0 || oh > 1024) return false;
    if (pad < 0 || pad > 4) return false;
    if (tm <= 0 || tm > 32) return false;
    if (tn < 0 || tn > 4) return false;
    if (tr <= 0 || tr > 13) return false;
    if (tc <= 0 || tc > 13) return false;
    return true;
}
}  // namespace
extern "C" int yolo2_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int yolo2_hip_select_device(int device)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (device < 0 || device >= yolo2_hip_device_count()) return fail(YOLO2_INIT_ERROR, "no HIP device %d", device);
    g_drv.device = device;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_accel_init(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (yolo2_hip_device_count() <= g_drv.device)
        return fail(YOLO2_INIT_ERROR, "no HIP device available (the GPU path has no CPU fallback)");
    HIP_TRY(hipSetDevice(g_drv.device), YOLO2_INIT_ERROR);
    g_drv.inited = true;
    g_drv.calls = 0;
    g_drv.last_rc = YOLO2_SUCCESS;
    memset(g_drv.regs, 0, sizeof(g_drv.regs));
    return YOLO2_SUCCESS;
}

// Device binding of the driver tier: every entry runs on the device chosen at init, whatever thread calls it.
static bool drv_ready_locked()
{
    return g_drv.inited && hipSetDevice(g_drv.device) == hipSuccess;
}

extern "C" void yolo2_accel_cleanup(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!g_drv.inited) return;
    (void)hipSetDevice(g_drv.device);
    (void)hipDeviceSynchronize();
    if (y2_process_options().verbose) fprintf(stderr, "[yolo2_hip] driver served %ld layer calls\n", g_drv.calls);
    y2_drv_release_i16();
    g_drv.inited = false;
}

extern "C" void yolo2_set_q_values(int32_t qw, int32_t qa_in, int32_t qa_out, int32_t qb)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    g_drv.qw = qw; g_drv.qa_in = qa_in; g_drv.qa_out = qa_out; g_drv.qb = qb;
}
// yolo2_accel_linux.c:179-196: before init the reference reports "not busy" / "done"
extern "C" int yolo2_is_busy(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return 0;
    return hipStreamQuery(nullptr) == hipErrorNotReady ? 1 : 0;
}
extern "C" int yolo2_is_done(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return 1;
    return hipStreamQuery(nullptr) == hipSuccess ? 1 : 0;
}
extern "C" int yolo2_wait_for_completion(uint32_t timeout_ms)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return fail(YOLO2_INIT_ERROR, "yolo2_accel_init() has not been called");
    return sync_with_timeout(nullptr, timeout_ms);
}
extern "C" long yolo2_hip_driver_calls(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    return g_drv.calls;
}
// dma_buffer_manager.h:94-139 on mapped pinned host memory: one set of pages, two addresses.
extern "C" int memory_allocate_ddr(size_t size, size_t alignment, memory_buffer_t *buffer)
{
    (void)alignment;  // hipHostMalloc returns page-aligned memory (reference asks for 4 KiB)
    if (!buffer || size == 0) return -1;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, size, hipHostMallocMapped) != hipSuccess) return fail(-1, "hipHostMalloc(%zu) failed", size);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipHostFree(h);
        return fail(-1, "hipHostGetDevicePointer failed");
    }
    memset(h, 0, size);
    buffer->ptr = h;
    buffer->size = size;
    buffer->phys_addr = (uint64_t)(uintptr_t)d;
    std::lock_guard<std::mutex> lk(g_drv.mu);
    g_drv.hostbufs.push_back({(char *)h, (char *)d, size});
    return 0;
}
extern "C" void memory_free_ddr(memory_buffer_t *buffer)
{
    if (!buffer || !buffer->ptr) return;
    bool tracked = false;
    {
        std::lock_guard<std::mutex> lk(g_drv.mu);
        auto &v = g_drv.hostbufs;
        const size_t before = v.size();
        v.erase(std::remove_if(v.begin(), v.end(), [&](const DriverState::HostBuf &b) { return b.host == buffer->ptr; }), v.end());
        tracked = v.size() != before;
    }
    if (tracked) (void)hipHostFree(buffer->ptr);   // (a buffer dma_buffer_cleanup already released is only forgotten)
    buffer->ptr = nullptr;
    buffer->size = 0;
    buffer->phys_addr = 0;
}
extern "C" int memory_allocate_weights(size_t size, memory_buffer_t *b) { return memory_allocate_ddr(size, 4096, b); }
extern "C" int memory_allocate_bias(size_t size, memory_buffer_t *b) { return memory_allocate_ddr(size, 4096, b); }
extern "C" int memory_allocate_inference_buffer(memory_buffer_t *b)
{
    // MEM_LEN int16 words + the reference's 512-element guard bands (yolo2_config.h:99, yolo2_model.cpp:243-244)
    return memory_allocate_ddr((size_t)(6922240 + 1024) * sizeof(int16_t), 4096, b);
}
extern "C" uint64_t memory_get_phys_addr(void *virt_addr)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    for (const auto &b : g_drv.hostbufs)
        if ((char *)virt_addr >= b.host && (char *)virt_addr < b.host + b.size)
            return (uint64_t)(uintptr_t)(b.dev + ((char *)virt_addr - b.host));
    return 0;
}
extern "C" void memory_flush_cache(void *addr, size_t size) { (void)addr; (void)size; __sync_synchronize(); }
extern "C" void memory_invalidate_cache(void *addr, size_t size) { (void)addr; (void)size; (void)hipDeviceSynchronize(); }

// dma_buffer_manager.h:32-92, the udmabuf-level interface.  "udmabuf present" becomes "a HIP device is present";
// a buffer is mapped pinned host memory like memory_allocate_ddr's (fd -1, device name "hip-pinned").
extern "C" int dma_buffer_init(void)
{
    if (yolo2_hip_device_count() < 1) {
        (void)fail(-1, "no HIP device available for DMA buffers (the GPU path has no CPU fallback)");
        return -1;
    }
    return 0;
}
extern "C" void dma_buffer_cleanup(void)
{
    std::vector<DriverState::HostBuf> left;
    {
        std::lock_guard<std::mutex> lk(g_drv.mu);
        left.swap(g_drv.hostbufs);
    }
    if (!left.empty()) (void)hipDeviceSynchronize();
    for (const auto &b : left) (void)hipHostFree(b.host);   // dma_buffer_manager.c:184-192: frees what is still tracked
}
extern "C" int dma_buffer_alloc(size_t size, dma_buffer_t *buffer)
{
    if (!buffer || size == 0) return -1;
    const size_t aligned = (size + 4095) & ~(size_t)4095;   // page multiple (dma_buffer_manager.c:232-234)
    memory_buffer_t mb;
    if (memory_allocate_ddr(aligned, 4096, &mb) != 0) return -1;
    memset(buffer, 0, sizeof(*buffer));
    buffer->virt_addr = mb.ptr;
    buffer->phys_addr = mb.phys_addr;
    buffer->size = aligned;
    buffer->fd = -1;
    snprintf(buffer->device_name, sizeof(buffer->device_name), "hip-pinned");
    return 0;
}
extern "C" void dma_buffer_free(dma_buffer_t *buffer)
{
    if (!buffer || !buffer->virt_addr) return;
    memory_buffer_t mb{buffer->virt_addr, buffer->size, buffer->phys_addr};
    memory_free_ddr(&mb);
    memset(buffer, 0, sizeof(*buffer));
}
extern "C" void dma_buffer_sync_for_device(dma_buffer_t *buffer, size_t offset, size_t size)
{
    (void)buffer; (void)offset; (void)size;
    __sync_synchronize();
}
extern "C" void dma_buffer_sync_for_cpu(dma_buffer_t *buffer, size_t offset, size_t size)
{
    (void)buffer; (void)offset; (void)size;
    (void)hipDeviceSynchronize();
}
extern "C" uint64_t dma_buffer_get_phys(dma_buffer_t *buffer, size_t offset) { return buffer ? buffer->phys_addr + offset : 0; }

// ---- per-layer calls

// The conv call with the driver lock held and the device bound (shared by yolo2_execute_conv_layer and the
// register-level start, yolo2_write_reg(AP_CTRL, ap_start)).
static int drv_conv_locked(uint64_t input_addr, uint64_t output_addr, uint64_t weight_addr, uint64_t beta_addr, int ifm_num,
                           int ofm_num, int ksize, int kstride, int input_w, int input_h, int output_w, int output_h,
                           int padding, int is_nl, int is_bn, int tm, int tn, int tr, int tc, int ofm_num_bound, int mloopsxTM,
                           int mloops_a1xTM, int layer_type, int qw, int qa_in, int qa_out, int qb, uint32_t timeout_ms)
{
    if (layer_type != 0) return fail(YOLO2_ERROR, "yolo2_execute_conv_layer: layer_type %d is not CONV", layer_type);
    if (!input_addr || !output_addr || !weight_addr || !beta_addr) return fail(YOLO2_ERROR, "null buffer address");
    if (!validate_conv_params(ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w, output_h, padding, tm, tn, tr, tc))
        return fail(YOLO2_ERROR, "conv parameters outside the accelerator's limits");
    if (output_w != (input_w - ksize + 2 * padding) / kstride + 1 || output_h != (input_h - ksize + 2 * padding) / kstride + 1)
        return fail(YOLO2_ERROR, "output size does not match input/kernel/stride/padding");
    // yolo2_accel_linux.c:463-466: Q arguments that are all zero leave the latched values in force
    if (qw != 0 || qa_in != 0 || qa_out != 0 || qb != 0) { g_drv.qw = qw; g_drv.qa_in = qa_in; g_drv.qa_out = qa_out; g_drv.qb = qb; }
    else { qw = g_drv.qw; qa_in = g_drv.qa_in; qa_out = g_drv.qa_out; qb = g_drv.qb; }
    // latch the call into the register file like yolo2_accel_linux.c:490-527 writes it
    reg_set64(R_INPUT, input_addr); reg_set64(R_OUTPUT, output_addr); reg_set64(R_WEIGHT, weight_addr); reg_set64(R_BETA, beta_addr);
    {
        const uint32_t offs[19] = {R_IFM, R_OFM, R_KSIZE, R_KSTRIDE, R_IN_W, R_IN_H, R_OUT_W, R_OUT_H, R_PAD, R_ISNL, R_ISBN, R_TM, R_TN,
                                   R_TR, R_TC, R_OFM_BOUND, R_MLOOPS, R_MLOOPS_A1, R_LTYPE};
        const int vals[19] = {ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w, output_h, padding, is_nl, is_bn, tm, tn,
                              tr, tc, ofm_num_bound, mloopsxTM, mloops_a1xTM, layer_type};
        for (int k = 0; k < 19; ++k) g_drv.regs[offs[k] / 4] = (uint32_t)vals[k];
    }
    g_drv.calls++;
    const int rc = y2_drv_conv_i16((const short *)(uintptr_t)input_addr, (short *)(uintptr_t)output_addr, (const short *)(uintptr_t)weight_addr,
                                   (const short *)(uintptr_t)beta_addr, ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w, output_h,
                                   padding, is_nl, qw, qa_in, qa_out, qb, &g_drv.last_path);
    if (rc) return rc;
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return sync_with_timeout(nullptr, timeout_ms);
}

extern "C" int yolo2_execute_conv_layer(uint64_t input_addr, uint64_t output_addr, uint64_t weight_addr,
                                        uint64_t beta_addr, int ifm_num, int ofm_num, int ksize, int kstride,
                                        int input_w, int input_h, int output_w, int output_h, int padding,
                                        int is_nl, int is_bn, int tm, int tn, int tr, int tc, int ofm_num_bound,
                                        int mloopsxTM, int mloops_a1xTM, int layer_type, int qw, int qa_in,
                                        int qa_out, int qb, uint32_t timeout_ms)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return fail(YOLO2_INIT_ERROR, "yolo2_accel_init() has not been called");
    return g_drv.last_rc = drv_conv_locked(input_addr, output_addr, weight_addr, beta_addr, ifm_num, ofm_num, ksize, kstride, input_w, input_h,
                           output_w, output_h, padding, is_nl, is_bn, tm, tn, tr, tc, ofm_num_bound, mloopsxTM, mloops_a1xTM,
                           layer_type, qw, qa_in, qa_out, qb, timeout_ms);
}

extern "C" int yolo2_hip_last_layer_path(void) { return g_drv.last_path; }

static int drv_pool_locked(uint64_t input_addr, uint64_t output_addr, int channels, int ksize, int kstride, int input_w,
                           int input_h, int output_w, int output_h, int padding, int tm, int tr, int tc, int ofm_num_bound,
                           int mloopsxTM, int mloops_a1xTM, uint32_t timeout_ms)
{
    // padding is forced to 0 by the scheduler (core_scheduler.cpp:72-73)
    if (!input_addr || !output_addr) return fail(YOLO2_ERROR, "null buffer address");
    if (!validate_conv_params(channels, channels, ksize, kstride, input_w, input_h, output_w, output_h, 0, tm, 0, tr, tc))
        return fail(YOLO2_ERROR, "maxpool parameters outside the accelerator's limits");
    // yolo2_accel_linux.c:580-655 latches a pool as LayerType 1 with IFM = OFM = channels, TN = 0
    reg_set64(R_INPUT, input_addr); reg_set64(R_OUTPUT, output_addr);
    {
        const uint32_t offs[19] = {R_IFM, R_OFM, R_KSIZE, R_KSTRIDE, R_IN_W, R_IN_H, R_OUT_W, R_OUT_H, R_PAD, R_ISNL, R_ISBN, R_TM, R_TN,
                                   R_TR, R_TC, R_OFM_BOUND, R_MLOOPS, R_MLOOPS_A1, R_LTYPE};
        const int vals[19] = {channels, channels, ksize, kstride, input_w, input_h, output_w, output_h, padding, 0, 0, tm, 0,
                              tr, tc, ofm_num_bound, mloopsxTM, mloops_a1xTM, 1};
        for (int k = 0; k < 19; ++k) g_drv.regs[offs[k] / 4] = (uint32_t)vals[k];
    }
    g_drv.calls++;
    y2_drv_pool_i16((const short *)(uintptr_t)input_addr, (short *)(uintptr_t)output_addr, channels, ksize, kstride, input_w, input_h, output_w,
                    output_h);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return sync_with_timeout(nullptr, timeout_ms);
}

extern "C" int yolo2_execute_maxpool_layer(uint64_t input_addr, uint64_t output_addr, int channels, int ksize,
                                           int kstride, int input_w, int input_h, int output_w, int output_h,
                                           int padding, int tm, int tr, int tc, int ofm_num_bound, int mloopsxTM,
                                           int mloops_a1xTM, uint32_t timeout_ms)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return fail(YOLO2_INIT_ERROR, "yolo2_accel_init() has not been called");
    return g_drv.last_rc = drv_pool_locked(input_addr, output_addr, channels, ksize, kstride, input_w, input_h, output_w, output_h, padding, tm, tr,
                           tc, ofm_num_bound, mloopsxTM, mloops_a1xTM, timeout_ms);
}

// Register file (yolo2_accel_linux.c:232-258).  AP_CTRL is synthesised from the stream state; a write of ap_start
// to it runs the layer the registers describe.
extern "C" uint32_t yolo2_get_status(void)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return 0;
    return hipStreamQuery(nullptr) == hipErrorNotReady ? AP_START : (AP_DONE | AP_IDLE | AP_READY);
}
extern "C" uint32_t yolo2_read_reg(uint32_t offset)
{
    if (offset == R_AP_CTRL) return yolo2_get_status();
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!g_drv.inited || offset >= sizeof(g_drv.regs)) return 0;
    if (offset == R_HIP_STATUS) return (uint32_t)g_drv.last_rc;   // 0, or a YOLO2_* error code as a two's-complement word
    return g_drv.regs[offset / 4];
}
extern "C" void yolo2_write_reg(uint32_t offset, uint32_t value)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked() || offset >= sizeof(g_drv.regs)) return;
    if (offset == R_HIP_STATUS) return;   // read-only
    if (offset != R_AP_CTRL) { g_drv.regs[offset / 4] = value; return; }
    if (!(value & AP_START)) return;
    const uint32_t *r = g_drv.regs;
    auto R = [&](uint32_t off) { return (int)r[off / 4]; };
    // A register-level start has no return value: its status is latched in R_HIP_STATUS (and in yolo2_hip_last_error()), so a
    // client polling ap_done can tell a rejected layer from a finished one.
    if (R(R_LTYPE) == 0)
        g_drv.last_rc = drv_conv_locked(reg_get64(R_INPUT), reg_get64(R_OUTPUT), reg_get64(R_WEIGHT), reg_get64(R_BETA), R(R_IFM), R(R_OFM),
                              R(R_KSIZE), R(R_KSTRIDE), R(R_IN_W), R(R_IN_H), R(R_OUT_W), R(R_OUT_H), R(R_PAD), R(R_ISNL), R(R_ISBN),
                              R(R_TM), R(R_TN), R(R_TR), R(R_TC), R(R_OFM_BOUND), R(R_MLOOPS), R(R_MLOOPS_A1), 0, 0, 0, 0, 0, 0);
    else if (R(R_LTYPE) == 1)
        g_drv.last_rc = drv_pool_locked(reg_get64(R_INPUT), reg_get64(R_OUTPUT), R(R_IFM), R(R_KSIZE), R(R_KSTRIDE), R(R_IN_W), R(R_IN_H),
                              R(R_OUT_W), R(R_OUT_H), R(R_PAD), R(R_TM), R(R_TR), R(R_TC), R(R_OFM_BOUND), R(R_MLOOPS),
                              R(R_MLOOPS_A1), 0);
    else
        g_drv.last_rc = fail(YOLO2_ERROR, "register start: LayerType %d is not served by the accelerator", R(R_LTYPE));
}

extern "C" int yolo2_execute_conv_layer_f32(uint64_t input_addr, uint64_t output_addr, uint64_t weight_addr,
                                            uint64_t beta_addr, int ifm_num, int ofm_num, int ksize, int kstride,
                                            int input_w, int input_h, int output_w, int output_h, int padding,
                                            int is_nl, uint32_t timeout_ms)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked()) return fail(YOLO2_INIT_ERROR, "yolo2_accel_init() has not been called");
    if (!input_addr || !output_addr || !weight_addr || !beta_addr) return fail(YOLO2_ERROR, "null buffer address");
    if (!validate_conv_params(ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w, output_h, padding, 1, 0, 1, 1))
        return fail(YOLO2_ERROR, "conv parameters outside the accelerator's limits");
    y2_drv_conv_f32((const float *)(uintptr_t)input_addr, (float *)(uintptr_t)output_addr, (const float *)(uintptr_t)weight_addr,
                    (const float *)(uintptr_t)beta_addr, ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w, output_h, padding,
                    is_nl ? 1 : 0);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return sync_with_timeout(nullptr, timeout_ms);
}
