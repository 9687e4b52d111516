// yolo2_hip.hip -- the part of libyolo2_hip.so (include/yolo2_hip.h) every tier shares: error slot, model tables, context
// lifecycle, HBM helpers, per-layer profiling, GPU pre-processing and the host-buffer streaming entries.  The tiers themselves:
// yolo2_driver.hip (reference driver interface), yolo2_int16.hip, yolo2_fp16.hip, yolo2_fp32.hip (y2_internal.hpp has the map).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "y2_internal.hpp"
#include "kernels_pre.hpp"

using namespace y2;

// ---------------------------------------------------------------------------- errors

static thread_local char g_err[512] = "";
static const bool g_verbose = y2_process_options().verbose;   // latched once: no getenv on any launch path

int y2_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (g_verbose) fprintf(stderr, "[yolo2_hip] %s\n", g_err);
    return code;
}

extern "C" const char *yolo2_hip_last_error(void) { return g_err; }
// for the library's other translation units (yolo2_multi.hip, yolo2_post.hip): same thread-local message slot
extern "C" int yolo2_hip_set_error(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }

// ---------------------------------------------------------------------------- model tables

extern "C" const int yolo2_weight_len[YOLO2_N_CONV] = {864, 18432, 73728, 8192, 73728, 294912, 32768, 294912,
                                                       1179648, 131072, 1179648, 131072, 1179648, 4718592,
                                                       524288, 4718592, 524288, 4718592, 9437184, 9437184,
                                                       32768, 11796480, 435200};
extern "C" const int yolo2_bias_len[YOLO2_N_CONV] = {32, 64, 128, 64, 128, 256, 128, 256, 512, 256, 512, 256,
                                                     512, 1024, 512, 1024, 512, 1024, 1024, 1024, 64, 1024, 425};

extern "C" long yolo2_strip_int16_layer_pad(const int16_t *file, size_t file_elems, const int *layer_len,
                                            int n_layers, int16_t *dst)
{
    size_t fo = 0, oo = 0;
    for (int l = 0; l < n_layers; ++l) {
        const size_t len = (size_t)layer_len[l];
        if (fo + len > file_elems) return -1;
        memcpy(dst + oo, file + fo, len * sizeof(int16_t));
        fo += len + (len & 1);  // yolo2_model.cpp:215-220
        oo += len;
    }
    return (long)oo;
}

// config/yolov2.cfg as parsed by the reference; the C host re-derives the same
// table from the .cfg file and checks it against this one before using the batched entry.
const LayerDesc kNet[32] = {
    {L_CONV, 3, 416, 416, 32, 3, 1},    {L_MAX, 32, 416, 416, 32, 2, 0},   {L_CONV, 32, 208, 208, 64, 3, 1},
    {L_MAX, 64, 208, 208, 64, 2, 0},    {L_CONV, 64, 104, 104, 128, 3, 1}, {L_CONV, 128, 104, 104, 64, 1, 1},
    {L_CONV, 64, 104, 104, 128, 3, 1},  {L_MAX, 128, 104, 104, 128, 2, 0}, {L_CONV, 128, 52, 52, 256, 3, 1},
    {L_CONV, 256, 52, 52, 128, 1, 1},   {L_CONV, 128, 52, 52, 256, 3, 1},  {L_MAX, 256, 52, 52, 256, 2, 0},
    {L_CONV, 256, 26, 26, 512, 3, 1},   {L_CONV, 512, 26, 26, 256, 1, 1},  {L_CONV, 256, 26, 26, 512, 3, 1},
    {L_CONV, 512, 26, 26, 256, 1, 1},   {L_CONV, 256, 26, 26, 512, 3, 1},  {L_MAX, 512, 26, 26, 512, 2, 0},
    {L_CONV, 512, 13, 13, 1024, 3, 1},  {L_CONV, 1024, 13, 13, 512, 1, 1}, {L_CONV, 512, 13, 13, 1024, 3, 1},
    {L_CONV, 1024, 13, 13, 512, 1, 1},  {L_CONV, 512, 13, 13, 1024, 3, 1}, {L_CONV, 1024, 13, 13, 1024, 3, 1},
    {L_CONV, 1024, 13, 13, 1024, 3, 1}, {L_ROUTE, 0, 0, 0, 0, 0, 0},       {L_CONV, 512, 26, 26, 64, 1, 1},
    {L_REORG, 64, 26, 26, 256, 0, 0},   {L_ROUTE, 0, 0, 0, 0, 0, 0},       {L_CONV, 1280, 13, 13, 1024, 3, 1},
    {L_CONV, 1024, 13, 13, 425, 1, 0},  {L_REGION, 425, 13, 13, 0, 0, 0},
};

extern "C" int yolo2_hip_num_layers(void) { return 32; }
extern "C" int yolo2_hip_layer_desc(int i, int desc[9])
{
    if (i < 0 || i >= 32 || !desc) return YOLO2_ERROR;
    const LayerDesc &l = kNet[i];
    static const int type_code[] = {0, 1, 3, 2, 4};  // yolo2_config.h:118-122 numbering
    const int stride = l.type == L_MAX || l.type == L_REORG ? 2 : (l.type == L_CONV ? 1 : 0);
    const int pad = l.type == L_CONV && l.size == 3 ? 1 : 0;
    const int v[9] = {type_code[l.type], l.c, l.h, l.w, l.n, l.size, stride, pad, l.leaky};
    for (int k = 0; k < 9; ++k) desc[k] = v[k];
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- HBM helpers

int y2_ensure(void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return YOLO2_SUCCESS;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(p, need), YOLO2_MMAP_ERROR);
    *cap = need;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_alloc(size_t bytes, uint64_t *dev_addr)
{
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes), YOLO2_MMAP_ERROR);
    *dev_addr = (uint64_t)(uintptr_t)p;
    return YOLO2_SUCCESS;
}
extern "C" int yolo2_hip_alloc_on(yolo2_hip_ctx *c, size_t bytes, uint64_t *dev_addr)
{
    if (!c || !dev_addr) return fail(YOLO2_ERROR, "null argument");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    return yolo2_hip_alloc(bytes, dev_addr);
}
extern "C" void yolo2_hip_free(uint64_t dev_addr) { (void)hipFree((void *)(uintptr_t)dev_addr); }
extern "C" int yolo2_hip_memcpy_h2d(uint64_t dst, const void *src, size_t bytes)
{
    HIP_TRY(hipMemcpy((void *)(uintptr_t)dst, src, bytes, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    return YOLO2_SUCCESS;
}
extern "C" int yolo2_hip_memcpy_d2h(void *dst, uint64_t src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, (const void *)(uintptr_t)src, bytes, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
    return YOLO2_SUCCESS;
}
extern "C" int yolo2_hip_memset(uint64_t dst, int value, size_t bytes)
{
    HIP_TRY(hipMemset((void *)(uintptr_t)dst, value, bytes), YOLO2_DMA_ERROR);
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- context lifecycle

extern "C" int yolo2_hip_create(int device, yolo2_hip_ctx **out)
{
    if (!out) return fail(YOLO2_ERROR, "null ctx pointer");
    if (device < 0 || device >= yolo2_hip_device_count())
        return fail(YOLO2_INIT_ERROR, "no HIP device %d (the GPU path has no CPU fallback)", device);
    HIP_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    yolo2_hip_ctx *c = new (std::nothrow) yolo2_hip_ctx();
    if (!c) return fail(YOLO2_ERROR, "out of host memory");
    c->device = device;
    c->opt = Y2Options::from_env();   // the ONLY read of the YOLO2_* switches for this context (yolo2_hip_set_option changes them later)
    *out = c;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_ctx_device(yolo2_hip_ctx *c) { return c ? c->device : -1; }

void y2_free_activations(yolo2_hip_ctx *c)
{
    if (c->ks_trip) (void)hipFree(c->ks_trip);
    c->ks_trip = nullptr;
    c->ks_trip_bytes = 0;
    if (c->t_in.d) (void)hipFree(c->t_in.d);
    c->t_in.d = nullptr;
    if (c->t_cat.d) (void)hipFree(c->t_cat.d);
    c->t_cat.d = nullptr;
    for (int i = 0; i < 32; ++i) {
        if (c->t_out[i].d && i != 24 && i != 27) (void)hipFree(c->t_out[i].d);
        c->t_out[i].d = nullptr;
    }
    c->batch = 0;
}

void y2_free_f16_activations(yolo2_hip_ctx *c)
{
    if (c->h_in.d) (void)hipFree(c->h_in.d);
    if (c->h_cat.d) (void)hipFree(c->h_cat.d);
    c->h_in.d = c->h_cat.d = nullptr;
    for (int i = 0; i < 32; ++i) {
        if (c->h_out[i].d && i != 24 && i != 27) (void)hipFree(c->h_out[i].d);
        c->h_out[i].d = nullptr;
    }
    c->f16_batch = 0;
}

void y2_free_f32_activations(yolo2_hip_ctx *c)
{
    if (c->f_in.d) (void)hipFree(c->f_in.d);
    if (c->f_cat.d) (void)hipFree(c->f_cat.d);
    c->f_in.d = c->f_cat.d = nullptr;
    for (int i = 0; i < 32; ++i) {
        if (c->f_out[i].d && i != 24 && i != 27) (void)hipFree(c->f_out[i].d);
        c->f_out[i].d = nullptr;
    }
    c->f32_batch = 0;
}

void y2_destroy_lanes(yolo2_hip_ctx *c)
{
    c->lane_first.clear();
    for (yolo2_hip_ctx *l : c->lanes) yolo2_hip_destroy(l);
    c->lanes.clear();
    c->laned = false;
}

static void pipe_free(PipeBufs &p)
{
    for (int k = 0; k < 2; ++k) {
        y2_post_free(&p.post[k]);
        if (p.hgeom[k]) (void)hipHostFree(p.hgeom[k]);
        if (p.hdets[k]) (void)hipHostFree(p.hdets[k]);
        if (p.hcounts[k]) (void)hipHostFree(p.hcounts[k]);
        if (p.hin[k]) (void)hipHostFree(p.hin[k]);
        if (p.hout[k]) (void)hipHostFree(p.hout[k]);
        (void)hipFree(p.dbytes[k]); (void)hipFree(p.din[k]); (void)hipFree(p.dout[k]);
        if (p.e_in[k]) (void)hipEventDestroy(p.e_in[k]);
        if (p.e_run[k]) (void)hipEventDestroy(p.e_run[k]);
        if (p.e_out[k]) (void)hipEventDestroy(p.e_out[k]);
        for (hipEvent_t &e : p.e_lane[k]) if (e) (void)hipEventDestroy(e);
    }
    if (p.s_in) (void)hipStreamDestroy(p.s_in);
    if (p.s_run) (void)hipStreamDestroy(p.s_run);
    if (p.s_out) (void)hipStreamDestroy(p.s_out);
    p = PipeBufs();
}

// need_hout: the pinned mirror of the region tensor (the entries that return region tensors; the images -> detections entry
// downloads records instead and skips these 2 x 9 MB of pinned memory, which cost ~10 ms each to allocate)
static int pipe_ensure(PipeBufs &p, size_t host_in, size_t dev_in, int batch, bool need_hout = true)
{
    if (p.s_in && p.host_in >= host_in && p.dev_in >= dev_in && p.batch >= batch && (!need_hout || p.hout[0])) return YOLO2_SUCCESS;
    need_hout = need_hout || p.hout[0] != nullptr;
    host_in = std::max(host_in, p.host_in);
    dev_in = std::max(dev_in, p.dev_in);
    batch = std::max(batch, p.batch);
    pipe_free(p);
    const size_t fbytes = (size_t)batch * YOLO2_FRAME_ELEMS * sizeof(float), rbytes = (size_t)batch * YOLO2_REGION_ELEMS * sizeof(int16_t);
    bool ok = true;
    for (int k = 0; k < 2 && ok; ++k) {
        ok = hipHostMalloc((void **)&p.hin[k], host_in, hipHostMallocDefault) == hipSuccess &&
             (!need_hout || hipHostMalloc((void **)&p.hout[k], rbytes, hipHostMallocDefault) == hipSuccess) &&
             (dev_in == 0 || hipMalloc((void **)&p.dbytes[k], dev_in) == hipSuccess) &&
             hipMalloc((void **)&p.din[k], fbytes) == hipSuccess && hipMalloc((void **)&p.dout[k], rbytes) == hipSuccess &&
             hipEventCreateWithFlags(&p.e_in[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&p.e_run[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&p.e_out[k], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&p.s_run, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        pipe_free(p);
        return fail(YOLO2_MMAP_ERROR, "staging buffers for %d frames (+%zu bytes) could not be allocated", batch, host_in);
    }
    p.host_in = host_in; p.dev_in = dev_in; p.batch = batch;
    return YOLO2_SUCCESS;
}

static const bool g_lane_prio = y2_process_options().lane_priority != 0;   // process-wide (the priority pools are): latched once

bool y2_lane0_own_stream() { return g_lane_prio; }

int y2_lane_stream_create(hipStream_t *s)
{
    if (!g_lane_prio) {
        HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking), YOLO2_ERROR);
        return YOLO2_SUCCESS;
    }
    int least = 0, greatest = 0;   // numerically lower = higher priority
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least &&
        hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess)
        return YOLO2_SUCCESS;
    (void)hipGetLastError();       // a runtime without stream priorities: an ordinary stream (correct, possibly sharing a hardware queue)
    HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

extern "C" void yolo2_hip_destroy(yolo2_hip_ctx *c)
{
    if (!c) return;
    yolo2_hip_rccl_finalize(c);   // leaves its communicator, if it joined one (no-op otherwise; never loads librccl)
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    pipe_free(c->pipe);
    y2_f16_plan_free(c);
    y2_destroy_lanes(c);
    y2_free_activations(c);
    if (c->lane_stream) (void)hipStreamDestroy(c->lane_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (yolo2_hip_ctx *l : c->f16_lanes) yolo2_hip_destroy(l);
    c->f16_lanes.clear();
    if (c->tol) yolo2_hip_destroy(c->tol);
    c->tol = nullptr;
    if (c->is_lane) {   // packed weights and biases belong to the parent
        c->wpk = nullptr;
        c->bias_pk = nullptr;
        c->wh = nullptr; c->biasf = nullptr; c->w0f = nullptr; c->wf32 = nullptr; c->bf32 = nullptr;
    }
    if (c->borrows_f32) { c->w0f = nullptr; c->wf32 = nullptr; c->bf32 = nullptr; }   // the split-mode twin: its wh / biasf are its own
    y2_free_f16_activations(c);
    y2_free_f32_activations(c);
    if (c->wpkf) (void)hipFree(c->wpkf);
    if (c->biasf32_pk) (void)hipFree(c->biasf32_pk);
    if (c->wh) (void)hipFree(c->wh);
    if (c->biasf) (void)hipFree(c->biasf);
    if (c->w0f) (void)hipFree(c->w0f);
    if (c->wf32) (void)hipFree(c->wf32);
    if (c->bf32) (void)hipFree(c->bf32);
    if (c->wpk) (void)hipFree(c->wpk);
    if (c->bias_pk) (void)hipFree(c->bias_pk);
    if (c->mb_lists) (void)hipFree(c->mb_lists);
    if (c->ev_made)
        for (auto &slot : c->ev)
            for (auto &e : slot) (void)hipEventDestroy(e);
    delete c;
}

// ---------------------------------------------------------------------------- lanes / profiling accessors

extern "C" int yolo2_hip_num_lanes(yolo2_hip_ctx *c) { return c && c->laned ? (int)c->lanes.size() : 1; }
extern "C" int yolo2_hip_num_lanes_fp16(yolo2_hip_ctx *c) { return c && !c->f16_lanes.empty() ? (int)c->f16_lanes.size() : 1; }

int y2_ensure_prof_events(yolo2_hip_ctx *c)
{
    if (c->ev_made) return YOLO2_SUCCESS;
    for (auto &slot : c->ev)
        for (auto &e : slot) HIP_TRY(hipEventCreate(&e), YOLO2_ERROR);
    c->ev_made = true;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_set_profiling(yolo2_hip_ctx *c, int enable)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    if (c->laned) {   // with lanes the events of lane 0 are reported (its launches overlap lane 1's)
        c->prof = enable != 0;
        c->prof_runs = 0;
        return yolo2_hip_set_profiling(c->lanes[0], enable);
    }
    if (enable) {
        const int rc = y2_ensure_prof_events(c);
        if (rc) return rc;
    }
    c->prof = enable != 0;
    c->prof_runs = 0;  // (re)start the averaging window
    if (c->tol) (void)yolo2_hip_set_profiling(c->tol, enable);                              // the split-mode twin follows its parent
    if (!c->f16_lanes.empty()) return yolo2_hip_set_profiling(c->f16_lanes[0], enable);   // fp16 lanes: lane 0 is reported
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_layer_times_ms(yolo2_hip_ctx *c, float *ms32)
{
    if (!c || !ms32) return fail(YOLO2_ERROR, "null argument");
    if (c->laned) return yolo2_hip_layer_times_ms(c->lanes[0], ms32);
    if (c->prof_runs == 0 && !c->f16_lanes.empty() && c->f16_lanes[0]->prof_runs > 0) return yolo2_hip_layer_times_ms(c->f16_lanes[0], ms32);
    if (c->prof_runs == 0 && c->tol) return yolo2_hip_layer_times_ms(c->tol, ms32);   // only the split-mode twin has run since profiling was switched on
    if (c->prof_runs == 0 && !c->f16_lanes.empty()) return yolo2_hip_layer_times_ms(c->f16_lanes[0], ms32);
    if (c->prof_runs == 0) return fail(YOLO2_ERROR, "no profiled run yet");
    const int n = (int)std::min<long>(c->prof_runs, yolo2_hip_ctx::kProfSlots);
    for (int i = 0; i < 32; ++i) ms32[i] = 0.f;
    for (int sidx = 0; sidx < n; ++sidx) {
        HIP_TRY(hipEventSynchronize(c->ev[sidx][32]), YOLO2_ERROR);
        for (int i = 0; i < 32; ++i) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, c->ev[sidx][i], c->ev[sidx][i + 1]), YOLO2_ERROR);
            ms32[i] += t / n;
        }
    }
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- pre-processing

static int letterbox_args(int w, int h, int channels, int net_w, int net_h, LetterboxArgs &a)
{
    if (w <= 0 || h <= 0 || net_w <= 1 || net_h <= 1 || (channels != 1 && channels != 3))
        return fail(YOLO2_ERROR, "letterbox: bad image geometry %dx%dx%d -> %dx%d", w, h, channels, net_w, net_h);
    if ((long)w * h > (1L << 28)) return fail(YOLO2_ERROR, "letterbox: image too large");
    a.w = w; a.h = h; a.ch = channels; a.net_w = net_w; a.net_h = net_h;
    // letterbox_image, src/core/yolo_image.cpp:148-165
    if (((float)net_w / w) < ((float)net_h / h)) { a.new_w = net_w; a.new_h = (h * net_w) / w; }
    else { a.new_h = net_h; a.new_w = (w * net_h) / h; }
    if (a.new_w < 1 || a.new_h < 1) return fail(YOLO2_ERROR, "letterbox: image aspect too extreme (%dx%d)", w, h);
    a.off_x = (net_w - a.new_w) / 2;
    a.off_y = (net_h - a.new_h) / 2;
    // resize_image, src/core/yolo_image.cpp:89-90 (a one-pixel-wide target divides by zero there too;
    // the value is then never used because its only column takes the c == w-1 branch)
    a.w_scale = (float)(w - 1) / (a.new_w - 1);
    a.h_scale = (float)(h - 1) / (a.new_h - 1);
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_letterbox_u8(uint64_t image_dev, int w, int h, int channels, uint64_t frame_dev, int net_w,
                                      int net_h, void *stream)
{
    if (!image_dev || !frame_dev) return fail(YOLO2_ERROR, "null buffer address");
    LetterboxArgs a;
    int rc = letterbox_args(w, h, channels, net_w, net_h, a);
    if (rc) return rc;
    hipLaunchKernelGGL(k_letterbox_u8, dim3(blocks_for((long)3 * net_w * net_h, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)(uintptr_t)image_dev, (float *)(uintptr_t)frame_dev, a);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

// Camera-style entry: n images of arbitrary sizes as host bytes -> region tensors, in chunks of
// `batch` images.  The bytes (not the 4x larger float frames) cross PCIe; letterboxing runs on the
// GPU into the chunk's frame buffer.  Same three-stream pipeline as yolo2_hip_run_frames_int16:
// upload of chunk k+1 (CPU staging copy + DMA) overlaps the kernels of chunk k and the download of k-1.
extern "C" int yolo2_hip_run_images_u8_host(yolo2_hip_ctx *c, const uint8_t *const *images, const int *widths,
                                            const int *heights, int channels, int n, int batch, int16_t *region,
                                            int *final_q)
{
    if (!c || !images || !widths || !heights || !region) return fail(YOLO2_ERROR, "null argument");
    if (n <= 0 || batch <= 0) return fail(YOLO2_ERROR, "bad image count %d / batch %d", n, batch);
    if (!c->weights_loaded) return fail(YOLO2_ERROR, "weights not loaded");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    batch = std::min(batch, n);
    const int chunks = (n + batch - 1) / batch;
    auto in_chunk = [&](int k) { return std::min(batch, n - k * batch); };
    auto padded = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t table_bytes = padded((size_t)batch * sizeof(LetterboxItem));   // the chunk's letterbox table leads its staging buffer
    size_t cap = 0;   // bytes of the largest chunk
    for (int k = 0; k < chunks; ++k) {
        size_t sum = table_bytes;
        for (int i = k * batch; i < k * batch + in_chunk(k); ++i) {
            LetterboxArgs a;
            if (!images[i]) return fail(YOLO2_ERROR, "null image %d", i);
            const int rc = letterbox_args(widths[i], heights[i], channels, 416, 416, a);
            if (rc) return rc;
            sum += padded((size_t)widths[i] * heights[i] * channels);
        }
        cap = std::max(cap, sum);
    }
    if (batch != c->batch) {
        const int rc = yolo2_hip_set_batch(c, batch);
        if (rc) return rc;
    }
    const size_t rbytes = (size_t)batch * YOLO2_REGION_ELEMS * sizeof(int16_t);
    int rc = pipe_ensure(c->pipe, cap, cap, batch);
    if (rc) return rc;
    PipeBufs &P = c->pipe;
    uint8_t **hin = P.hin, **dbytes = P.dbytes;
    float **din = P.din;
    int16_t **hout = P.hout, **dout = P.dout;
    hipStream_t s_in = P.s_in, s_run = P.s_run, s_out = P.s_out;
    hipEvent_t *e_in = P.e_in, *e_run = P.e_run, *e_out = P.e_out;
    auto cleanup = [&]() { (void)hipDeviceSynchronize(); };
#define Y2_TRY(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(code, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)

    auto drain = [&](int k) {
        const int b = k & 1;
        (void)hipEventSynchronize(e_out[b]);
        memcpy(region + (size_t)k * batch * YOLO2_REGION_ELEMS, hout[b], (size_t)in_chunk(k) * YOLO2_REGION_ELEMS * sizeof(int16_t));
    };
    int q = 0;
    std::vector<size_t> offs((size_t)batch);
    for (int k = 0; k < chunks && rc == YOLO2_SUCCESS; ++k) {
        const int b = k & 1, nf = in_chunk(k), first = k * batch;
        if (k >= 2) drain(k - 2);   // buffer set b is free again once chunk k-2 has left it
        size_t off = table_bytes;
        for (int i = 0; i < nf; ++i) {
            const size_t bytes = (size_t)widths[first + i] * heights[first + i] * channels;
            memcpy(hin[b] + off, images[first + i], bytes);
            offs[(size_t)i] = off;
            off += padded(bytes);
        }
        LetterboxItem *items = reinterpret_cast<LetterboxItem *>(hin[b]);
        for (int f = 0; f < batch && rc == YOLO2_SUCCESS; ++f) {   // a partial last chunk repeats its last image
            const int i = std::min(f, nf - 1);
            items[f].off = offs[(size_t)i];
            rc = letterbox_args(widths[first + i], heights[first + i], channels, 416, 416, items[f].a);
        }
        if (rc) break;
        Y2_TRY(hipMemcpyAsync(dbytes[b], hin[b], off, hipMemcpyHostToDevice, s_in), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(e_in[b], s_in), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(s_run, e_in[b], 0), YOLO2_ERROR);
        hipLaunchKernelGGL(k_letterbox_u8_batch, dim3(blocks_for((long)YOLO2_FRAME_ELEMS, 256), batch), dim3(256), 0, s_run, dbytes[b], din[b],
                           (int)YOLO2_FRAME_ELEMS);   // the whole chunk in one launch
        Y2_TRY(hipGetLastError(), YOLO2_ERROR);
        rc = yolo2_hip_run_batch_int16(c, (uint64_t)(uintptr_t)din[b], batch, (uint64_t)(uintptr_t)dout[b], &q, s_run);
        if (rc) break;
        Y2_TRY(hipEventRecord(e_run[b], s_run), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(s_out, e_run[b], 0), YOLO2_ERROR);
        Y2_TRY(hipMemcpyAsync(hout[b], dout[b], rbytes, hipMemcpyDeviceToHost, s_out), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(e_out[b], s_out), YOLO2_ERROR);
    }
    if (rc == YOLO2_SUCCESS) {
        for (int k = std::max(0, chunks - 2); k < chunks; ++k) drain(k);
        if (final_q) *final_q = q;
    }
    cleanup();
#undef Y2_TRY
    return rc;
}

// ---------------------------------------------------------------------------- streaming host entry

// Camera-to-detections entry (the whole path incl. the step after it, at the path's rate): n images as host bytes -> detection
// records.  Per chunk of `batch` images: bytes H2D, letterbox + network on the device, the region tensor stays in HBM and
// region + boxes + NMS + record compaction run on the SAME device and stream right behind the network; only the records
// (32 bytes each) and the per-frame counts come back.  Upload of chunk k+1, kernels of chunk k and download of chunk k-1 overlap on
// three HIP streams.  Replaces the loop of the reference's streaming app (linux_app/src/main.c:878-1288), which runs
// yolo2_run_inference and the host post-processing frame by frame.
static int pipe_ensure_post(yolo2_hip_ctx *c, int batch, int cap)
{
    PipeBufs &p = c->pipe;
    if (p.post_batch >= batch && p.post_cap >= cap) return YOLO2_SUCCESS;
    batch = std::max(batch, p.post_batch);
    cap = std::max(cap, p.post_cap);
    for (int k = 0; k < 2; ++k) {
        if (p.hgeom[k]) (void)hipHostFree(p.hgeom[k]);
        if (p.hdets[k]) (void)hipHostFree(p.hdets[k]);
        if (p.hcounts[k]) (void)hipHostFree(p.hcounts[k]);
        p.hgeom[k] = nullptr; p.hdets[k] = nullptr; p.hcounts[k] = nullptr;
        p.post_batch = p.post_cap = 0;
        const int rc = y2_post_alloc(c->device, batch, cap, &p.post[k]);
        if (rc) return rc;
        if (hipHostMalloc((void **)&p.hgeom[k], (size_t)batch * y2_post_geom_bytes(), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&p.hdets[k], (size_t)batch * cap * sizeof(yolo2_hip_det), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&p.hcounts[k], (size_t)batch * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(YOLO2_MMAP_ERROR, "pinned buffers for %d frames x %d detection records could not be allocated", batch, cap);
        }
    }
    p.post_batch = batch; p.post_cap = cap;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_run_images_u8_dets(yolo2_hip_ctx *c, const uint8_t *const *images, const int *widths, const int *heights,
                                            int channels, int n, int batch, float thresh, float nms, int flags, yolo2_hip_det *dets,
                                            int cap_per_frame, int *counts, int *final_q)
{
    if (!c || !images || !widths || !heights || !dets || !counts) return fail(YOLO2_ERROR, "null argument");
    if (n <= 0 || batch <= 0 || cap_per_frame <= 0) return fail(YOLO2_ERROR, "bad image count %d / batch %d / capacity %d", n, batch, cap_per_frame);
    if (thresh < 0.f) return fail(YOLO2_ERROR, "negative threshold");
    if (!c->weights_loaded) return fail(YOLO2_ERROR, "weights not loaded");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    batch = std::min(batch, n);
    const int chunks = (n + batch - 1) / batch, cap = cap_per_frame, best_only = (flags & YOLO2_DETS_BEST_CLASS) ? 1 : 0;
    auto in_chunk = [&](int k) { return std::min(batch, n - k * batch); };
    auto padded = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t table_bytes = padded((size_t)batch * sizeof(LetterboxItem));   // the chunk's letterbox table leads its staging buffer
    size_t cap_bytes = 0;   // bytes of the largest chunk
    for (int k = 0; k < chunks; ++k) {
        size_t sum = table_bytes;
        for (int i = k * batch; i < k * batch + in_chunk(k); ++i) {
            LetterboxArgs a;
            if (!images[i]) return fail(YOLO2_ERROR, "null image %d", i);
            const int rc = letterbox_args(widths[i], heights[i], channels, 416, 416, a);
            if (rc) return rc;
            sum += padded((size_t)widths[i] * heights[i] * channels);
        }
        cap_bytes = std::max(cap_bytes, sum);
    }
    if (batch != c->batch) {
        const int rc = yolo2_hip_set_batch(c, batch);
        if (rc) return rc;
    }
    int rc = pipe_ensure(c->pipe, cap_bytes, cap_bytes, batch, false);
    if (rc == YOLO2_SUCCESS) rc = pipe_ensure_post(c, batch, cap);
    if (rc) return rc;
    PipeBufs &P = c->pipe;
    const size_t gbytes = y2_post_geom_bytes();
    auto cleanup = [&]() { (void)hipDeviceSynchronize(); };
#define Y2_TRY(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(code, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
    auto drain = [&](int k) {
        const int b = k & 1, nf = in_chunk(k);
        (void)hipEventSynchronize(P.e_out[b]);
        memcpy(counts + (size_t)k * batch, P.hcounts[b], (size_t)nf * sizeof(int));
        for (int f = 0; f < nf; ++f) {   // frame numbers are global in the caller's records
            const int cnt = std::min(P.hcounts[b][f], cap);
            yolo2_hip_det *dst = dets + ((size_t)k * batch + f) * cap;
            memcpy(dst, P.hdets[b] + (size_t)f * cap, (size_t)cnt * sizeof(yolo2_hip_det));
            for (int r = 0; r < cnt; ++r) dst[r].frame = k * batch + f;
        }
    };
    int q = 0;
    std::vector<size_t> offs((size_t)batch);
    std::vector<int> cw((size_t)batch), chh((size_t)batch);
    for (int k = 0; k < chunks && rc == YOLO2_SUCCESS; ++k) {
        const int b = k & 1, nf = in_chunk(k), first = k * batch;
        if (k >= 2) drain(k - 2);   // buffer set b is free again once chunk k-2 has left it
        size_t off = table_bytes;
        for (int i = 0; i < nf; ++i) {
            const size_t bytes = (size_t)widths[first + i] * heights[first + i] * channels;
            memcpy(P.hin[b] + off, images[first + i], bytes);
            offs[(size_t)i] = off;
            off += padded(bytes);
        }
        LetterboxItem *items = reinterpret_cast<LetterboxItem *>(P.hin[b]);
        for (int f = 0; f < batch; ++f) {   // a partial last chunk repeats its last image
            const int i = std::min(f, nf - 1);
            cw[(size_t)f] = widths[first + i]; chh[(size_t)f] = heights[first + i];
            items[f].off = offs[(size_t)i];
            if ((rc = letterbox_args(widths[first + i], heights[first + i], channels, 416, 416, items[f].a))) break;
        }
        if (rc) break;
        if ((rc = y2_post_fill_geom(P.hgeom[b], cw.data(), chh.data(), batch))) break;
        Y2_TRY(hipMemcpyAsync(P.dbytes[b], P.hin[b], off, hipMemcpyHostToDevice, P.s_in), YOLO2_DMA_ERROR);
        Y2_TRY(hipMemcpyAsync(P.post[b].geom, P.hgeom[b], (size_t)batch * gbytes, hipMemcpyHostToDevice, P.s_in), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(P.e_in[b], P.s_in), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(P.s_run, P.e_in[b], 0), YOLO2_ERROR);
        hipLaunchKernelGGL(k_letterbox_u8_batch, dim3(blocks_for((long)YOLO2_FRAME_ELEMS, 256), batch), dim3(256), 0, P.s_run, P.dbytes[b], P.din[b],
                           (int)YOLO2_FRAME_ELEMS);
        Y2_TRY(hipGetLastError(), YOLO2_ERROR);
        // The network.  A chunk's lanes are NOT joined back into one stream here: consecutive chunks are independent, every lane
        // owns its activations and its stream keeps its chunks in order, so lane i starts chunk k + 1 the moment it has finished
        // chunk k (and the chunk's letterbox is done) instead of waiting for the slowest lane - the fork / join per 64 frames and the
        // lanes' end spread (0.3 - 0.7 ms of a 20 ms step in the bench loop) do not exist in a stream of chunks.  Only the tail,
        // which needs the whole region tensor, waits for all of them, on the download stream.
        bool own_streams = c->laned && (int)c->lanes.size() <= 8;
        if (own_streams) for (yolo2_hip_ctx *l : c->lanes) own_streams = own_streams && l->lane_stream;
        Y2_TRY(hipEventRecord(P.e_run[b], P.s_run), YOLO2_ERROR);            // (here: the letterboxed frames are ready)
        if (own_streams) {
            for (size_t i = 0; i < c->lanes.size() && rc == YOLO2_SUCCESS; ++i) {
                yolo2_hip_ctx *l = c->lanes[i];
                const uint64_t first = (uint64_t)c->lane_first[i];
                if (!P.e_lane[b][i]) Y2_TRY(hipEventCreateWithFlags(&P.e_lane[b][i], hipEventDisableTiming), YOLO2_ERROR);
                Y2_TRY(hipStreamWaitEvent(l->lane_stream, P.e_run[b], 0), YOLO2_ERROR);
                rc = yolo2_hip_run_batch_int16(l, (uint64_t)(uintptr_t)(P.din[b] + first * YOLO2_FRAME_ELEMS), l->batch,
                                               (uint64_t)(uintptr_t)(P.dout[b] + first * YOLO2_REGION_ELEMS), &q, l->lane_stream);
                if (rc) break;
                Y2_TRY(hipEventRecord(P.e_lane[b][i], l->lane_stream), YOLO2_ERROR);
                Y2_TRY(hipStreamWaitEvent(P.s_out, P.e_lane[b][i], 0), YOLO2_ERROR);
            }
        } else {
            rc = yolo2_hip_run_batch_int16(c, (uint64_t)(uintptr_t)P.din[b], batch, (uint64_t)(uintptr_t)P.dout[b], &q, P.s_run);
            if (rc) break;
            Y2_TRY(hipEventRecord(P.e_run[b], P.s_run), YOLO2_ERROR);
            Y2_TRY(hipStreamWaitEvent(P.s_out, P.e_run[b], 0), YOLO2_ERROR);
        }
        if (rc) break;
        // the step after the path, on the device that produced the tensor, straight from HBM - on the download stream, so that the
        // next chunk's letterbox and network need not wait for it (the tail is 64 workgroups for 0.7 ms: latency, not work)
        rc = y2_post_enqueue_int16(c->device, P.dout[b], batch, q, thresh, nms, cap, best_only, &P.post[b], P.s_out);
        if (rc) break;
        Y2_TRY(hipMemcpyAsync(P.hcounts[b], P.post[b].counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, P.s_out), YOLO2_DMA_ERROR);
        Y2_TRY(hipMemcpyAsync(P.hdets[b], P.post[b].dets, (size_t)batch * cap * sizeof(yolo2_hip_det), hipMemcpyDeviceToHost, P.s_out), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(P.e_out[b], P.s_out), YOLO2_ERROR);
    }
    if (rc == YOLO2_SUCCESS) {
        for (int k = std::max(0, chunks - 2); k < chunks; ++k) drain(k);
        if (final_q) *final_q = q;
    }
    cleanup();
#undef Y2_TRY
    return rc;
}

extern "C" int yolo2_hip_run_frames_int16(yolo2_hip_ctx *c, const float *frames, int n_frames, int batch, int16_t *region,
                                          int *final_q)
{
    if (!c || !frames || !region) return fail(YOLO2_ERROR, "null argument");
    if (n_frames <= 0 || batch <= 0) return fail(YOLO2_ERROR, "bad frame count / batch");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    if (batch != c->batch) {
        const int rc = yolo2_hip_set_batch(c, batch);
        if (rc) return rc;
    }
    const size_t fbytes = (size_t)batch * YOLO2_FRAME_ELEMS * sizeof(float), rbytes = (size_t)batch * YOLO2_REGION_ELEMS * sizeof(int16_t);
    int rc = pipe_ensure(c->pipe, fbytes, 0, batch);
    if (rc) return rc;
    PipeBufs &P = c->pipe;
    float *hin[2] = {(float *)P.hin[0], (float *)P.hin[1]};
    float **din = P.din;
    int16_t **hout = P.hout, **dout = P.dout;
    hipStream_t s_in = P.s_in, s_run = P.s_run, s_out = P.s_out;
    hipEvent_t *e_in = P.e_in, *e_run = P.e_run, *e_out = P.e_out;
    auto cleanup = [&]() { (void)hipDeviceSynchronize(); };
#define Y2_TRY(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(code, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)

    const int chunks = (n_frames + batch - 1) / batch;
    auto frames_in_chunk = [&](int k) { return std::min(batch, n_frames - k * batch); };
    auto drain = [&](int k) {   // copy chunk k's results from its pinned buffer to the caller's memory
        const int b = k & 1;
        (void)hipEventSynchronize(e_out[b]);
        memcpy(region + (size_t)k * batch * YOLO2_REGION_ELEMS, hout[b], (size_t)frames_in_chunk(k) * YOLO2_REGION_ELEMS * sizeof(int16_t));
    };
    int q = 0;
    for (int k = 0; k < chunks && rc == YOLO2_SUCCESS; ++k) {
        const int b = k & 1, nf = frames_in_chunk(k);
        if (k >= 2) drain(k - 2);   // buffer set b is free again once chunk k-2 has left it
        // stage: pageable -> pinned (CPU), pad a partial last chunk with its last frame
        memcpy(hin[b], frames + (size_t)k * batch * YOLO2_FRAME_ELEMS, (size_t)nf * YOLO2_FRAME_ELEMS * sizeof(float));
        for (int f = nf; f < batch; ++f)
            memcpy(hin[b] + (size_t)f * YOLO2_FRAME_ELEMS, hin[b] + (size_t)(nf - 1) * YOLO2_FRAME_ELEMS, YOLO2_FRAME_ELEMS * sizeof(float));
        Y2_TRY(hipMemcpyAsync(din[b], hin[b], fbytes, hipMemcpyHostToDevice, s_in), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(e_in[b], s_in), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(s_run, e_in[b], 0), YOLO2_ERROR);
        rc = yolo2_hip_run_batch_int16(c, (uint64_t)(uintptr_t)din[b], batch, (uint64_t)(uintptr_t)dout[b], &q, s_run);
        if (rc) break;
        Y2_TRY(hipEventRecord(e_run[b], s_run), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(s_out, e_run[b], 0), YOLO2_ERROR);
        Y2_TRY(hipMemcpyAsync(hout[b], dout[b], rbytes, hipMemcpyDeviceToHost, s_out), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(e_out[b], s_out), YOLO2_ERROR);
    }
    if (rc == YOLO2_SUCCESS) {
        for (int k = std::max(0, chunks - 2); k < chunks; ++k) drain(k);
        if (final_q) *final_q = q;
    }
    cleanup();
#undef Y2_TRY
    return rc;
}
