// yolo2_fp32.hip -- the exact fp32 path of libyolo2_hip.so (configs[0]'s precision): the reference's fp32 arithmetic in the
// reference's operation order, (a) one thread per output in the reference's [C][H][W8] layout (yolo2_execute_conv_layer_f32,
// yolo2_hip_run_frame_fp32_host) and (b) tiled and batched (csrc/kernels_f32.hpp, yolo2_hip_run_batch_fp32).  Both are
// bit-identical to the compiled reference's region tensor.  The fp32 blobs are loaded by yolo2_hip_load_weights_fp32 (yolo2_fp16.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "y2_internal.hpp"
#include "kernels_f32.hpp"

using namespace y2;

void y2_drv_conv_f32(const float *in, float *out, const float *w, const float *beta, int ifm, int ofm, int ksize, int kstride, int iw,
                     int ih, int ow, int oh, int pad, int is_nl)
{
    const int n = ofm * oh * ow;
    hipLaunchKernelGGL(k_conv_ref_f32, dim3(blocks_for(n, 256)), dim3(256), 0, nullptr, in, out, w, beta, ifm, ofm, ksize, kstride, iw, ih,
                       ow, oh, pad, is_nl ? 1 : 0);
}

// fp32 whole network, reference arithmetic: every layer in the reference's [C][H][W8] layout through the
// one-thread-per-output kernels (k_conv_ref_f32: reference operation order, no FMA contraction; k_pool_ref;
// the legacy reorg indexing of yolo2_model.cpp:112-129,358-376), i.e. what yolov2_hls_ps does at
// Precision::FP32 (yolo2_model.cpp:229-449).  Bit-identical to the reference's fp32 region tensor; a
// correctness path (about 0.2 s per frame), not a fast one - the fast floating-point path is run_batch_fp16.
extern "C" int yolo2_hip_run_frame_fp32_host(yolo2_hip_ctx *c, const float *frame, float *region)
{
    if (!c || !frame || !region) return fail(YOLO2_ERROR, "null argument");
    if (!c->f16_loaded || !c->wf32) return fail(YOLO2_ERROR, "fp32 weights not loaded (yolo2_hip_load_weights_fp32)");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    auto w8 = [](int w) { return (w + 7) & ~7; };
    float *bufs[32] = {nullptr};
    float *in0 = nullptr, *cat = nullptr;
    int rc = YOLO2_SUCCESS;
    auto release = [&]() {
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 32; ++i)
            if (bufs[i] && i != 24 && i != 27) (void)hipFree(bufs[i]);
        (void)hipFree(in0); (void)hipFree(cat);
    };
    auto dalloc = [&](float **p, size_t elems) -> bool {
        if (hipMalloc((void **)p, elems * sizeof(float)) != hipSuccess || hipMemsetAsync(*p, 0, elems * sizeof(float), nullptr) != hipSuccess) {
            rc = fail(YOLO2_MMAP_ERROR, "fp32 pass: activation buffer allocation failed");
            return false;
        }
        return true;
    };
    if (!dalloc(&in0, (size_t)3 * 416 * 416) || !dalloc(&cat, (size_t)1280 * 13 * 16)) { release(); return rc; }
    if (hipMemcpyAsync(in0, frame, (size_t)3 * 416 * 416 * sizeof(float), hipMemcpyHostToDevice, nullptr) != hipSuccess) {
        release();
        return fail(YOLO2_DMA_ERROR, "H2D of the frame failed");
    }
    const float *cur = in0;
    long woff = 0, boff = 0;
    int ord = 0;
    for (int i = 0; i < 32 && rc == YOLO2_SUCCESS; ++i) {
        const LayerDesc &l = kNet[i];
        const int pad = l.type == L_CONV ? (l.size == 3 ? 1 : 0) : 0;
        const int ow = l.type == L_CONV ? (l.w - l.size + 2 * pad) + 1 : l.w / 2, oh = l.type == L_CONV ? (l.h - l.size + 2 * pad) + 1 : l.h / 2;
        switch (l.type) {
        case L_CONV: {
            const float *src = i == 26 ? bufs[16] : (i == 29 ? cat : cur);
            float *dst = nullptr;
            if (i == 24) dst = cat + (size_t)256 * 13 * 16;
            else if (!dalloc(&dst, (size_t)l.n * oh * w8(ow))) break;
            hipLaunchKernelGGL(k_conv_ref_f32, dim3(blocks_for((long)l.n * oh * ow, 256)), dim3(256), 0, nullptr, src, dst,
                               (const float *)(c->wf32 + woff), (const float *)(c->bf32 + boff), l.c, l.n, l.size, 1, l.w, l.h, ow, oh,
                               pad, l.leaky);
            woff += yolo2_weight_len[ord];
            boff += yolo2_bias_len[ord];
            ord++;
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_MAX: {
            float *dst = nullptr;
            if (!dalloc(&dst, (size_t)l.c * oh * w8(ow))) break;
            hipLaunchKernelGGL((k_pool_ref<float>), dim3(blocks_for((long)l.c * oh * ow, 256)), dim3(256), 0, nullptr, cur, dst, l.c, 2, 2,
                               l.w, l.h, ow, oh, -1024.f * 1024.f);   // pad value of core_compute.cpp:291, core_io.cpp:101
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_REORG:
            hipLaunchKernelGGL(k_reorg_ref_f32, dim3(blocks_for(256 * 13 * 13, 256)), dim3(256), 0, nullptr, cur, cat);
            bufs[i] = cat;
            cur = cat;
            break;
        default:
            break;   // route: concat by placement; region: gathered below
        }
    }
    if (rc == YOLO2_SUCCESS && hipGetLastError() != hipSuccess) rc = fail(YOLO2_ERROR, "fp32 pass: kernel launch failed");
    if (rc == YOLO2_SUCCESS) {   // yolo2_model.cpp:406-414: 13 of 16 columns
        std::vector<float> padded((size_t)425 * 13 * 16);
        if (hipMemcpy(padded.data(), cur, padded.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(YOLO2_DMA_ERROR, "D2H of the region tensor failed");
        else
            for (int k = 0; k < 425 * 13; ++k) memcpy(region + (size_t)k * 13, padded.data() + (size_t)k * 16, 13 * sizeof(float));
    }
    release();
    return rc;
}

// ---------------------------------------------------------------------------- exact fp32, tiled (kernels_f32.hpp)

static int alloc_ftensor(yolo2_hip_ctx::FTensor &t, int C, int H, int W, int B)
{
    t.g = make_geom(C, H, W, B);
    HIP_TRY(hipMalloc((void **)&t.d, (size_t)t.g.items * 16), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(t.d, 0, (size_t)t.g.items * 16), YOLO2_DMA_ERROR);   // +0.0f: the conv padding and the 4th lane of the input
    return YOLO2_SUCCESS;
}

template <int KS, int P>
static void launch_conv_f32_n(const ConvPlan &p, const float4 *in, float4 *out, const float4 *w, const float *b, hipStream_t st)
{
    const int nst = (p.args.lt_max + 255) / 256;
    if (nst <= 2) hipLaunchKernelGGL((k_conv_f32<KS, P, 2>), p.grid, dim3(256), p.lds_bytes, st, in, out, w, b, p.args);
    else if (nst <= 4) hipLaunchKernelGGL((k_conv_f32<KS, P, 4>), p.grid, dim3(256), p.lds_bytes, st, in, out, w, b, p.args);
    else hipLaunchKernelGGL((k_conv_f32<KS, P, 8>), p.grid, dim3(256), p.lds_bytes, st, in, out, w, b, p.args);
}
static void launch_conv_f32(const ConvPlan &p, const float4 *in, float4 *out, const float4 *w, const float *b, hipStream_t st)
{
    if (p.K == 3) {
        if (p.P == 4) launch_conv_f32_n<3, 4>(p, in, out, w, b, st);
        else if (p.P == 2) launch_conv_f32_n<3, 2>(p, in, out, w, b, st);
        else launch_conv_f32_n<3, 1>(p, in, out, w, b, st);
    } else {
        if (p.P == 4) launch_conv_f32_n<1, 4>(p, in, out, w, b, st);
        else if (p.P == 2) launch_conv_f32_n<1, 2>(p, in, out, w, b, st);
        else launch_conv_f32_n<1, 1>(p, in, out, w, b, st);
    }
}

static void plan_conv_f32(ConvPlan &p, const LayerDesc &l, const ActGeom &gin, long out_cg_stride, long out_base, int P)
{
    p = ConvPlan();
    p.C = l.c; p.N = l.n; p.K = l.size; p.H = l.h; p.W = l.w; p.leaky = l.leaky;
    const int halo = l.size == 3 ? gin.Wp + 1 : 0;
    while (P > 1 && tile_items_bound(gin, 64 * P, halo) > kMaxTileItems) P >>= 1;
    p.P = P;
    ConvArgs &a = p.args;
    memset(&a, 0, sizeof(a));
    a.B = gin.B; a.H = gin.H; a.W = gin.W; a.Wp = gin.Wp; a.PL = gin.PL;
    a.CGin = gin.CG;
    a.CGout = (l.n + 3) / 4;
    a.npix = gin.B * gin.H * gin.W;
    set_conv_div(a);
    a.in_cg_stride = gin.cg_stride;
    a.out_cg_stride = out_cg_stride;
    a.out_base = out_base;
    a.leaky = l.leaky;
    a.lt_max = tile_items_bound(gin, 64 * P, halo);
    a.mb_list = nullptr;
    p.lds_bytes = (a.lt_max + l.size * l.size * 32) * 16 * 2;   // two buffers of {input tile, the group's 32-channel weight slice}
    p.grid = dim3((a.npix + 64 * P - 1) / (64 * P), (l.n + 31) / 32, 1);
    // same XCD grid rule as the int16 kernel (items and weights are twice as large: same ratio)
    const double in_bytes = (double)gin.B * gin.CG * gin.PL * 16, w_mb = (double)gin.CG * l.size * l.size * 32 * 16;
    const int gy = (int)p.grid.y, gx = (int)p.grid.x;
    double best = 0;
    for (int lg = 0; lg < 4; ++lg) {
        const int Xm = 1 << lg, Xt = 8 >> lg;
        if (Xm > gy || Xt > gx) continue;
        const int own = (gy + Xm - 1) / Xm;
        double G = 1;
        if (own * w_mb > 3.0e6) G = std::max(1.0, ((double)gx / Xt) / std::max(1, 128 / own));
        const double cost = in_bytes * Xm + w_mb * gy * Xt * G;
        if (!a.xcd_remap || cost < best) { best = cost; a.xcd_remap = 1 + lg; }
    }
}

static int ensure_f32_path(yolo2_hip_ctx *c, int B)
{
    if (!c->wpkf) {   // pack the resident fp32 blobs: partial tiles zero-padded, like the int16 weights
        long wtot = 0, btot = 0;
        int ord = 0;
        for (int i = 0; i < 32; ++i)
            if (kNet[i].type == L_CONV) {
                c->wpkf_off[ord] = wtot;
                c->biasf32_off[ord] = btot;
                wtot += packed_weight_elems(kNet[i].c, kNet[i].n, kNet[i].size);
                btot += (long)((kNet[i].n + 31) / 32) * 32;
                ord++;
            }
        HIP_TRY(hipMalloc((void **)&c->wpkf, (size_t)wtot * 4), YOLO2_MMAP_ERROR);
        HIP_TRY(hipMalloc((void **)&c->biasf32_pk, (size_t)btot * 4), YOLO2_MMAP_ERROR);
        HIP_TRY(hipMemset(c->biasf32_pk, 0, (size_t)btot * 4), YOLO2_DMA_ERROR);
        long woff = 0, boff = 0;
        ord = 0;
        for (int i = 0; i < 32; ++i) {
            const LayerDesc &l = kNet[i];
            if (l.type != L_CONV) continue;
            const long n = packed_weight_elems(l.c, l.n, l.size);
            hipLaunchKernelGGL((k_repack_weights<float>), dim3(blocks_for(n, 256)), dim3(256), 0, nullptr, (const float *)(c->wf32 + woff),
                               c->wpkf + c->wpkf_off[ord], l.c, l.n, l.size * l.size);
            HIP_TRY(hipMemcpyAsync(c->biasf32_pk + c->biasf32_off[ord], c->bf32 + boff, (size_t)l.n * 4, hipMemcpyDeviceToDevice, nullptr),
                    YOLO2_DMA_ERROR);
            woff += yolo2_weight_len[ord];
            boff += yolo2_bias_len[ord];
            ord++;
        }
        HIP_TRY(hipGetLastError(), YOLO2_ERROR);
        HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
    }
    if (c->f32_batch == B) return YOLO2_SUCCESS;
    y2_free_f32_activations(c);
    int rc;
    if ((rc = alloc_ftensor(c->f_in, 3, 416, 416, B))) return rc;
    if ((rc = alloc_ftensor(c->f_cat, 1280, 13, 13, B))) return rc;
    for (int i = 0; i < 31; ++i) {
        const LayerDesc &l = kNet[i];
        if (l.type == L_CONV && i != 24) {
            if ((rc = alloc_ftensor(c->f_out[i], l.n, l.h, l.w, B))) return rc;
        } else if (l.type == L_MAX) {
            if ((rc = alloc_ftensor(c->f_out[i], l.c, l.h / 2, l.w / 2, B))) return rc;
        }
    }
    c->f_out[24] = c->f_cat;
    c->f_out[27] = c->f_cat;
    c->f32_batch = B;
    // pixels per lane: timed once per layer (the arithmetic does not depend on it)
    const int fp = c->opt.f32_p;   // test hook: 1 / 2 / 4 for every layer (0: timed)
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), YOLO2_ERROR);
    HIP_TRY(hipEventCreate(&e1), YOLO2_ERROR);
    int ord = 0;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        if (l.type != L_CONV) continue;
        const auto &tin = i == 0 ? c->f_in : (i == 26 ? c->f_out[16] : (i == 29 ? c->f_cat : c->f_out[i - 1]));
        const auto &tout = c->f_out[i];
        const long out_base = kLead + (i == 24 ? (long)64 * tout.g.cg_stride : 0);
        float best = 1e30f;
        int bestP = 2;
        for (int P = 1; P <= 4; P <<= 1) {
            if (fp && fp != P) continue;
            ConvPlan cand;
            plan_conv_f32(cand, l, tin.g, tout.g.cg_stride, out_base, P);
            if (cand.P != P) continue;
            (void)hipEventRecord(e0, nullptr);
            launch_conv_f32(cand, tin.d, tout.d, (const float4 *)(c->wpkf + c->wpkf_off[ord]), c->biasf32_pk + c->biasf32_off[ord], nullptr);
            (void)hipEventRecord(e1, nullptr);
            HIP_TRY(hipEventSynchronize(e1), YOLO2_ERROR);
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, e0, e1), YOLO2_ERROR);
            if (t < best) { best = t; bestP = P; }
        }
        plan_conv_f32(c->fp32_plan[i], l, tin.g, tout.g.cg_stride, out_base, bestP);
        ord++;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_run_batch_fp32(yolo2_hip_ctx *c, uint64_t frames_dev, int batch, uint64_t region_dev, void *stream)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (!c->f16_loaded || !c->wf32) return fail(YOLO2_ERROR, "fp32 weights not loaded (yolo2_hip_load_weights_fp32)");
    if (!frames_dev || !region_dev) return fail(YOLO2_ERROR, "null buffer address");
    if (batch <= 0 || batch > 1024) return fail(YOLO2_ERROR, "batch %d out of range", batch);
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    int rc = ensure_f32_path(c, batch);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch;
    {
        const ActGeom &g = c->f_in.g;
        hipLaunchKernelGGL(k_pack_input_f32, dim3(blocks_for((long)B * g.H * g.W, 256)), dim3(256), 0, st, (const float *)(uintptr_t)frames_dev,
                           c->f_in.d, B, g.H, g.W, g.Wp, g.PL);
    }
    int ord = 0;
    const yolo2_hip_ctx::FTensor *cur = &c->f_in;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        switch (l.type) {
        case L_CONV: {
            const auto *tin = i == 26 ? &c->f_out[16] : (i == 29 ? &c->f_cat : cur);
            launch_conv_f32(c->fp32_plan[i], tin->d, c->f_out[i].d, (const float4 *)(c->wpkf + c->wpkf_off[ord]),
                            c->biasf32_pk + c->biasf32_off[ord], st);
            cur = &c->f_out[i];
            ord++;
            break;
        }
        case L_MAX: {
            const ActGeom &gi = cur->g, &go = c->f_out[i].g;
            const long n = (long)go.CG * B * go.H * go.W;
            hipLaunchKernelGGL(k_maxpool2_f32, dim3(blocks_for(n, 256)), dim3(256), 0, st, cur->d, c->f_out[i].d, go.CG, B, go.H, go.W, gi.Wp,
                               gi.PL, go.Wp, go.PL);
            cur = &c->f_out[i];
            break;
        }
        case L_REORG: {
            const ActGeom &gi = cur->g, &go = c->f_cat.g;
            hipLaunchKernelGGL(k_reorg_f32, dim3(blocks_for((long)B * 256 * 169, 256)), dim3(256), 0, st, (const float *)cur->d,
                               (float *)c->f_cat.d, B, gi.Wp, gi.PL, gi.cg_stride, go.Wp, go.PL, go.cg_stride);
            cur = &c->f_cat;
            break;
        }
        case L_ROUTE:
            break;
        case L_REGION: {
            const ActGeom &g = cur->g;
            hipLaunchKernelGGL(k_unpack_dense_f32, dim3(blocks_for((long)B * 425 * 169, 256)), dim3(256), 0, st, (const float *)cur->d,
                               (float *)(uintptr_t)region_dev, B, 425, 13, 13, g.Wp, g.PL, g.cg_stride);
            break;
        }
        }
    }
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_run_batch_fp32_host(yolo2_hip_ctx *c, const float *frames, int batch, float *region)
{
    if (!c || !frames || !region) return fail(YOLO2_ERROR, "null argument");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    float *fd = nullptr, *rd = nullptr;
    HIP_TRY(hipMalloc((void **)&fd, (size_t)batch * YOLO2_FRAME_ELEMS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&rd, (size_t)batch * YOLO2_REGION_ELEMS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemcpy(fd, frames, (size_t)batch * YOLO2_FRAME_ELEMS * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    int rc = yolo2_hip_run_batch_fp32(c, (uint64_t)(uintptr_t)fd, batch, (uint64_t)(uintptr_t)rd, nullptr);
    if (rc == YOLO2_SUCCESS) {
        hipError_t e = hipMemcpy(region, rd, (size_t)batch * YOLO2_REGION_ELEMS * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(YOLO2_DMA_ERROR, "D2H of region tensor failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(fd);
    (void)hipFree(rd);
    return rc;
}
