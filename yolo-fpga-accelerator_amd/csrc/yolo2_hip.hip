// yolo2_hip.hip -- C ABI of libyolo2_hip.so (include/yolo2_hip.h) and the launch logic.
//
// Tier 1 mirrors the reference's userspace driver (linux_app/src/yolo2_accel_linux.c:419-575,
// dma_buffer_manager.c) call for call; tier 2 restates the layer loop of yolov2_hls_ps
// (hls/models/yolov2/yolo2_model.cpp:229-449) as 28+3 kernel launches per batch on one stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "kernels_f16.hpp"
#include "kernels_pre.hpp"
#include "kernels_int16.hpp"
#include "kernels_f32.hpp"
#include "layout.hpp"

using namespace y2;

// ---------------------------------------------------------------------------- errors

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (getenv("YOLO2_VERBOSE")) fprintf(stderr, "[yolo2_hip] %s\n", g_err);
    return code;
}

#define HIP_TRY(expr, code)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

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

enum LType { L_CONV, L_MAX, L_ROUTE, L_REORG, L_REGION };
struct LayerDesc {
    LType type;
    int c, h, w, n, size, leaky;
};
// config/yolov2.cfg as parsed by the reference; the C host re-derives the same
// table from the .cfg file and checks it against this one before using the batched entry.
static const LayerDesc kNet[32] = {
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

// ---------------------------------------------------------------------------- launch helpers

struct ShiftSpec {
    int right, left, mag;
};
static ShiftSpec make_shift(int s)  // core_compute.cpp:48-63: magnitude capped at 30
{
    ShiftSpec sh;
    sh.right = s > 0;
    sh.left = s < 0;
    int a = sh.right ? s : (sh.left ? -s : 0);
    sh.mag = a > 30 ? 30 : a;
    return sh;
}

// 32-bit exactness: no intermediate of the fast kernel may leave int32.
//   |p + round| <= maxsum*32768 + round ;  |acc + scaled| <= max(32768,|bias0|) + |p + round|
// Form B keeps acc*2^s + round in the register:  max(32768,|bias0|)*2^s + round + |p| must fit.
// Form C packs two int16 accumulators per register and needs every increment to fit int16.
// Form D is form C with the shift folded into the weights (w * 2^(16-s) must still be int16 and the
// scaled dot product + 2^15 must fit int32); only offered when the caller passes the block's max |w|.
// Returns 0 (form A), 1 (form B), 3 (form C), 4 (form D) or 2 (64-bit); the narrowest legal form wins.
static int choose_path(int so, int sb, int maxsum, int max_abs_bias, int max_abs_w = -1)
{
    if (so < 0) return 2;
    const ShiftSpec o = make_shift(so), b = make_shift(sb);
    const long long round = o.mag > 0 ? (1LL << (o.mag - 1)) : 0;
    long long bias0 = max_abs_bias;
    if (b.right) bias0 = ((bias0 + (b.mag > 0 ? (1LL << (b.mag - 1)) : 0)) >> b.mag) + 1;
    else if (b.left) bias0 = bias0 << b.mag;
    const long long accmax = std::max<long long>(32768, bias0);
    const long long pmax = (long long)maxsum * 32768;
    if (bias0 > 2147483647LL) return 2;
    int path = 2;
    const bool okA = pmax + round + accmax <= 2147483647LL;
    const bool okB = (accmax << o.mag) + round + pmax <= 2147483647LL;
    // form C: every t = (p + round) >> s and the shifted bias must fit int16 (and p + round int32)
    const bool okC = okA && bias0 <= 32767 && ((pmax + round) >> o.mag) <= 32767;
    const int k = 16 - so;
    const bool okD = okC && max_abs_w >= 0 && k >= 0 && k <= 15 && ((long long)max_abs_w << k) <= 32767 &&
                     (((long long)maxsum << k) * 32768 + 32768) <= 2147483647LL;
    if (okA) path = 0;
    if (okB) path = 1;
    if (okC) path = 3;
    if (okD) path = 4;
    const char *force = getenv("YOLO2_FORCE_PATH");  // test hook: a narrower path only when it is legal, 2 always
    if (force) {
        const int f = atoi(force);
        if (f == 2 || (f == 0 && okA) || (f == 1 && okB) || (f == 3 && okC) || (f == 4 && okD)) path = f;
    }
    return path;
}

constexpr int kMaxTileItems = 2048;  // 8 staging registers x 256 threads (k_conv_i16 NST <= 8)

struct ConvPlan {
    int C = 0, N = 0, K = 0, H = 0, W = 0, leaky = 0;
    int Qw = 0, Qa_in = 0, Qa_out = 0, Qb = 0;
    int path = 0;  // 0 = form A, 1 = form B (pre-shifted accumulator), 3 = form C (packed int16), 4 = form D (C, shift-free), 2 = 64-bit
    int P = 8;
    int mb_count = 0;          // output-channel blocks this launch covers (0 = all of the layer)
    int splitk_ok = 0;         // the loader proved the split-K bounds for these blocks (|t| < 2^29, sums < 2^30)
    int splitk = 0;            // small batches: S K-splits x 64/S pixels per wavefront, shuffle-combined (k_conv_i16_splitk); 0 or S
    int splitk_pp = 1;         // pixels per lane of the split-K kernel (2: two pixel tiles share the staged weight slices)
    int grp = 1;               // 1x1 convs: channel groups per barrier (8 when it divides CGin and fits LDS staging)
    int pool_fused = 0;        // conv + leaky + 2x2 pool in one kernel (k_conv_i16_pool): 1 = pooled tensor only, 2 = + full tensor
    int lds_pad = 0;           // extra dynamic LDS requested only to cap workgroups per CU (autotuned):
                               // fewer co-resident workgroups finish sooner each, which shortens the
                               // idle tail of layers that are only a few workgroup-generations long
    dim3 grid;
    int lds_bytes = 0;
    ConvArgs args;
};

static void plan_conv(ConvPlan &p, const ActGeom &gin, long out_cg_stride, long out_base, int CGout, int forceP = 0)
{
    const ShiftSpec so = make_shift(p.Qa_in + p.Qw - p.Qa_out), sb = make_shift(p.Qb - p.Qa_out);
    const int npix = gin.B * gin.H * gin.W;
    const int maxP = (p.path == 1 || p.path == 3 || p.path == 4) ? 8 : 4;  // form A / 64-bit: 8 pixels per lane overflows the register file
    if (forceP) {
        p.P = std::min(forceP, maxP);
    } else {
        p.P = maxP;
        // small problems (single frame): fewer pixels per lane -> more workgroups
        while (p.P > 1 && (long)((npix + 64 * p.P - 1) / (64 * p.P)) * (p.mb_count ? p.mb_count : (p.N + 31) / 32) < 1024) p.P >>= 1;
    }
    const int halo = p.K == 3 ? gin.Wp + 1 : 0;
    while (p.P > 1 && tile_items_bound(gin, 64 * p.P, halo) > kMaxTileItems) p.P >>= 1;
    if (p.splitk) {   // splitk = number of K-splits S (4 or 8); 64/S pixels per wavefront
        if (p.splitk_pp > 1 && !(p.splitk == 4 && p.K == 3 && p.path == 4 && !getenv("YOLO2_SPLITK_NO_PACK"))) p.splitk_pp = 1;   // only built for 3x3 form D layers
        const int S = p.splitk, lt = tile_items_bound(gin, 64 / S * p.splitk_pp, halo);
        // (8 splits only for 1x1 layers: on the 3x3 layers the kernel is bound by re-staging the weight slices per
        //  pixel tile, and halving the tile to 8 pixels measured 2x slower)
        if ((S != 4 && !(S == 8 && p.K == 1)) || gin.CG % S != 0 || gin.CG < 4 * S || S * (lt + p.K * p.K * 32) > kMaxTileItems) p.splitk = 0;
    }
    if (p.splitk) p.P = 1;
    const int T = p.splitk ? 64 / p.splitk * p.splitk_pp : 64 * p.P;
    ConvArgs &a = p.args;
    // (a.mb_list is owned by the caller: nullptr unless the layer is split by arithmetic form)
    a.B = gin.B; a.H = gin.H; a.W = gin.W; a.Wp = gin.Wp; a.PL = gin.PL;
    a.CGin = gin.CG;
    a.CGout = CGout;
    a.npix = npix;
    set_conv_div(a);
    a.in_cg_stride = gin.cg_stride;
    a.out_cg_stride = out_cg_stride;
    a.out_base = out_base;
    a.shift = so.mag;
    a.round = (so.right && so.mag > 0) ? (1 << (so.mag - 1)) : 0;
    if (p.path == 4) { a.shift = 16; a.round = 32768; }   // form D: these blocks' weights are stored as w * 2^(16-s)
    a.sh_right = so.right; a.sh_left = so.left;
    a.bs_right = sb.right; a.bs_left = sb.left; a.bs_mag = sb.mag;
    a.leaky = p.leaky;
    a.lt_max = tile_items_bound(gin, T, halo);
    p.grp = (!p.splitk && p.K == 1 && p.path != 2 && gin.CG % 8 == 0 && a.lt_max * 8 <= kMaxTileItems && !getenv("YOLO2_NO_GRP")) ? 8 : 1;
    // (two channel groups per barrier for the 3x3 forms C/D - one barrier per 18 taps - was measured: -1 to -2 %)
    p.lds_bytes = p.splitk ? p.splitk * (a.lt_max + p.K * p.K * 32 + 4) * 8 * 2 : std::max(a.lt_max * p.grp * 8 * 2, p.lds_pad);  // double-buffered input tile (or the occupancy cap)
    p.grid = dim3((npix + T - 1) / T, p.mb_count ? p.mb_count : (p.N + 31) / 32, 1);
    // XCD grid over (tiles, blocks): bytes crossing the fabric = input x Xm + weights x Xt x G, where
    // G > 1 only if the blocks one XCD owns do not keep their weights in its 4 MiB L2 (then every
    // generation of co-resident tiles fetches them again).  See xcd_partition in kernels_int16.hpp.
    a.xcd_remap = 0;
    if (!getenv("YOLO2_NO_XCD_REMAP")) {
        const double in_bytes = (double)gin.B * gin.CG * gin.PL * 8;
        const double w_mb = (double)gin.CG * p.K * p.K * 32 * 8;
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
}

// Upper bound of the LDS tile (items) of k_conv_i16_pool: 64 consecutive pool windows in raster order.  Between the
// top-left pixels of two consecutive windows the flat offset grows by 2 (same row pair), W + 4 (next row pair) or
// 2W + 5 (next frame); the tile adds a halo of Wp + 1 on either side and the bottom-right pixel of its last window.
static int pool_tile_items_bound(const ActGeom &g)
{
    const int OW = g.W / 2, OHW = (g.H / 2) * OW;
    const int wraps = std::min(63, (63 + OW - 1) / OW), frames = std::min(63, (63 + OHW - 1) / OHW);
    return 126 + wraps * (g.W + 2) + frames * (g.W + 1) + 3 * g.Wp + 4;
}

// Re-plans a conv launch (already planned by plan_conv for this input geometry) as the fused conv + pool kernel.
static bool plan_conv_pool(ConvPlan &p, const ActGeom &gin, const ActGeom &gpool, int full)
{
    if (p.K != 3 || (p.path != 3 && p.path != 4) || (gin.H & 1) || (gin.W & 1)) return false;
    const int lt = pool_tile_items_bound(gin);
    if (lt > 12 * 256) return false;
    ConvArgs &a = p.args;
    p.splitk = 0; p.grp = 1; p.P = 4; p.lds_pad = 0;
    p.pool_fused = full ? 2 : 1;
    a.lt_max = lt;
    a.nwin = gin.B * (gin.H / 2) * (gin.W / 2);
    a.oWp = gpool.Wp; a.oPL = gpool.PL;
    a.pool_cg_stride = gpool.cg_stride;
    a.pool_base = kLead;
    p.lds_bytes = lt * 8 * (gin.CG > 1 ? 2 : 1);
    p.grid = dim3((a.nwin + 63) / 64, p.mb_count ? p.mb_count : (p.N + 31) / 32, 1);
    return true;
}

template <int MODE, bool FULL>
static void launch_conv_pool_n(const ConvPlan &p, const int2 *in, int2 *out, int2 *out_pool, const int2 *wpk, const short *bias,
                               hipStream_t st)
{
    const int nst = (p.args.lt_max + 255) / 256;
#define Y2_POOL(NSTV, SINGLEV) hipLaunchKernelGGL((k_conv_i16_pool<MODE, NSTV, FULL, SINGLEV>), p.grid, dim3(256), p.lds_bytes, st, in, out, out_pool, wpk, bias, p.args)
    if (p.args.CGin == 1) {   // layer 0: the whole input tile is staged once, up front
        if (nst <= 4) Y2_POOL(4, true);
        else Y2_POOL(12, true);
    } else if (nst <= 2) Y2_POOL(2, false);
    else if (nst <= 3) Y2_POOL(3, false);
    else if (nst <= 5) Y2_POOL(5, false);
    else if (nst <= 8) Y2_POOL(8, false);
    else Y2_POOL(12, false);
#undef Y2_POOL
}

template <int KS, int MODE, int P>
static void launch_conv_n(const ConvPlan &p, const int2 *in, int2 *out, const int2 *wpk, const short *bias, hipStream_t st)
{
    const int nst = (p.args.lt_max * p.grp + 255) / 256;
    if (KS == 1 && p.grp == 8 && MODE != 2) {
        if (nst <= 4) hipLaunchKernelGGL((k_conv_i16<1, P, MODE == 2 ? 1 : MODE, 4, KS == 1 ? 8 : 1>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
        else hipLaunchKernelGGL((k_conv_i16<1, P, MODE == 2 ? 1 : MODE, 8, KS == 1 ? 8 : 1>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
        return;
    }
    if (nst <= 2) hipLaunchKernelGGL((k_conv_i16<KS, P, MODE, 2>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
    else if (nst <= 4) hipLaunchKernelGGL((k_conv_i16<KS, P, MODE, 4>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
    else hipLaunchKernelGGL((k_conv_i16<KS, P, MODE, 8>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
}

template <int KS, int MODE>
static void launch_conv_p(const ConvPlan &p, const int2 *in, int2 *out, const int2 *wpk, const short *bias,
                          hipStream_t st)
{
    switch (p.P) {
    case 8: launch_conv_n<KS, MODE, 8>(p, in, out, wpk, bias, st); break;
    case 4: launch_conv_n<KS, MODE, 4>(p, in, out, wpk, bias, st); break;
    case 2: launch_conv_n<KS, MODE, 2>(p, in, out, wpk, bias, st); break;
    default: launch_conv_n<KS, MODE, 1>(p, in, out, wpk, bias, st); break;
    }
}

static void launch_conv(const ConvPlan &p, const int2 *in, int2 *out, const int2 *wpk, const short *bias, hipStream_t st,
                        int2 *out_pool = nullptr)
{
    if (p.pool_fused) {   // out_pool: the pooled tensor (the layer after this conv)
        if (p.path == 4) {
            if (p.pool_fused == 2) launch_conv_pool_n<4, true>(p, in, out, out_pool, wpk, bias, st);
            else launch_conv_pool_n<4, false>(p, in, out, out_pool, wpk, bias, st);
        } else {
            if (p.pool_fused == 2) launch_conv_pool_n<3, true>(p, in, out, out_pool, wpk, bias, st);
            else launch_conv_pool_n<3, false>(p, in, out, out_pool, wpk, bias, st);
        }
        return;
    }
    if (p.splitk) {
        const int nst = (p.splitk * (p.args.lt_max + p.K * p.K * 32) + 255) / 256;
        const bool pack = p.path == 4 && !getenv("YOLO2_SPLITK_NO_PACK");   // form D layers: packed int16 triples
#define Y2_SPLITK(KSV, NSTV, PACKV, SV) \
    hipLaunchKernelGGL((k_conv_i16_splitk<KSV, NSTV, PACKV, SV>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args)
#define Y2_SPLITK_S(KSV, NSTV, PACKV) do { if (p.splitk == 8) Y2_SPLITK(KSV, NSTV, PACKV, 8); else Y2_SPLITK(KSV, NSTV, PACKV, 4); } while (0)
        if (pack) {
            if (p.K == 3 && p.splitk_pp == 4) hipLaunchKernelGGL((k_conv_i16_splitk<3, 8, true, 4, 4>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
            else if (p.K == 3 && p.splitk_pp == 2) hipLaunchKernelGGL((k_conv_i16_splitk<3, 8, true, 4, 2>), p.grid, dim3(256), p.lds_bytes, st, in, out, wpk, bias, p.args);
            else if (p.K == 3) Y2_SPLITK(3, 8, true, 4);
            else if (nst <= 2) Y2_SPLITK_S(1, 2, true);
            else Y2_SPLITK_S(1, 8, true);
        } else {
            if (p.K == 3) Y2_SPLITK(3, 8, false, 4);
            else if (nst <= 2) Y2_SPLITK_S(1, 2, false);
            else Y2_SPLITK_S(1, 8, false);
        }
#undef Y2_SPLITK_S
#undef Y2_SPLITK
        return;
    }
    if (p.K == 3) {
        if (p.path == 2) launch_conv_p<3, 2>(p, in, out, wpk, bias, st);
        else if (p.path == 4) launch_conv_p<3, 4>(p, in, out, wpk, bias, st);
        else if (p.path == 3) launch_conv_p<3, 3>(p, in, out, wpk, bias, st);
        else if (p.path == 1) launch_conv_p<3, 1>(p, in, out, wpk, bias, st);
        else launch_conv_p<3, 0>(p, in, out, wpk, bias, st);
    } else {
        if (p.path == 2) launch_conv_p<1, 2>(p, in, out, wpk, bias, st);
        else if (p.path == 4) launch_conv_p<1, 4>(p, in, out, wpk, bias, st);
        else if (p.path == 3) launch_conv_p<1, 3>(p, in, out, wpk, bias, st);
        else if (p.path == 1) launch_conv_p<1, 1>(p, in, out, wpk, bias, st);
        else launch_conv_p<1, 0>(p, in, out, wpk, bias, st);
    }
}

static inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

static long packed_weight_elems(int C, int N, int K)
{
    return (long)((N + kTm - 1) / kTm) * ((C + kTn - 1) / kTn) * K * K * 128;
}

// ---------------------------------------------------------------------------- tier 1: driver state

namespace {
struct DriverState {
    std::mutex mu;
    int device = 0;
    bool inited = false;
    int qw = 0, qa_in = 0, qa_out = 0, qb = 0;
    // grow-only scratch for the per-layer calls
    void *in_items = nullptr, *out_items = nullptr, *wpk = nullptr, *bias_pk = nullptr;
    int last_path = -1;   // arithmetic form of the most recent yolo2_execute_conv_layer (-1: generic reference-layout kernel)
    size_t in_cap = 0, out_cap = 0, wpk_cap = 0, bias_cap = 0;
    int *bound = nullptr;
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
    AP_START = 1u << 0, AP_DONE = 1u << 1, AP_IDLE = 1u << 2, AP_READY = 1u << 3,
};
inline void reg_set64(uint32_t off, uint64_t v) { g_drv.regs[off / 4] = (uint32_t)v; g_drv.regs[off / 4 + 1] = (uint32_t)(v >> 32); }
inline uint64_t reg_get64(uint32_t off) { return (uint64_t)g_drv.regs[off / 4] | ((uint64_t)g_drv.regs[off / 4 + 1] << 32); }

int ensure(void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return YOLO2_SUCCESS;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(p, need), YOLO2_MMAP_ERROR);
    *cap = need;
    return YOLO2_SUCCESS;
}

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
    if (!g_drv.bound) HIP_TRY(hipMalloc((void **)&g_drv.bound, 4 * sizeof(int)), YOLO2_MMAP_ERROR);   // [max sum, max sum (1 block), max |w|, scale byte]
    g_drv.inited = true;
    g_drv.calls = 0;
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
    if (getenv("YOLO2_VERBOSE")) fprintf(stderr, "[yolo2_hip] driver served %ld layer calls\n", g_drv.calls);
    for (void *p : {g_drv.in_items, g_drv.out_items, g_drv.wpk, g_drv.bias_pk, (void *)g_drv.bound})
        if (p) (void)hipFree(p);
    g_drv.in_items = g_drv.out_items = g_drv.wpk = g_drv.bias_pk = nullptr;
    g_drv.bound = nullptr;
    g_drv.in_cap = g_drv.out_cap = g_drv.wpk_cap = g_drv.bias_cap = 0;
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

extern "C" int yolo2_hip_alloc(size_t bytes, uint64_t *dev_addr)
{
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes), YOLO2_MMAP_ERROR);
    *dev_addr = (uint64_t)(uintptr_t)p;
    return YOLO2_SUCCESS;
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

static int max_abs_i16_dev(const short *dev, int n, int *out)
{
    std::vector<short> h(n);
    HIP_TRY(hipMemcpy(h.data(), dev, (size_t)n * 2, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
    int m = 0;
    for (short v : h) m = std::max(m, std::abs((int)v));
    *out = m;
    return YOLO2_SUCCESS;
}

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
    hipStream_t st = nullptr;
    const short *in = (const short *)(uintptr_t)input_addr;
    short *out = (short *)(uintptr_t)output_addr;
    const short *w = (const short *)(uintptr_t)weight_addr;
    const short *beta = (const short *)(uintptr_t)beta_addr;
    const int so = qa_in + qw - qa_out, sb = qb - qa_out;

    bool tiled = kstride == 1 && ((ksize == 3 && padding == 1) || (ksize == 1 && padding == 0));
    if (tiled) {  // very wide images: the halo of a 64-pixel tile must fit the LDS staging scheme
        const ActGeom g = make_geom(ifm_num, input_h, input_w, 1);
        if (tile_items_bound(g, 64, ksize == 3 ? g.Wp + 1 : 0) > kMaxTileItems) tiled = false;
    }
    if (!tiled) {
        g_drv.last_path = -1;
        const int n = ofm_num * output_h * output_w;
        hipLaunchKernelGGL(k_conv_ref_i16, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, w, beta, ifm_num, ofm_num,
                           ksize, kstride, input_w, input_h, output_w, output_h, padding, is_nl ? 1 : 0, so, sb);
        HIP_TRY(hipGetLastError(), YOLO2_ERROR);
        return sync_with_timeout(st, timeout_ms);
    }

    const ActGeom gi = make_geom(ifm_num, input_h, input_w, 1), go = make_geom(ofm_num, output_h, output_w, 1);
    const long wpk_elems = packed_weight_elems(ifm_num, ofm_num, ksize);
    const int MB = (ofm_num + 31) / 32;
    int rc;
    if ((rc = ensure(&g_drv.in_items, &g_drv.in_cap, (size_t)gi.items * 8))) return rc;
    if ((rc = ensure(&g_drv.out_items, &g_drv.out_cap, (size_t)go.items * 8))) return rc;
    if ((rc = ensure(&g_drv.wpk, &g_drv.wpk_cap, (size_t)wpk_elems * 2))) return rc;
    if ((rc = ensure(&g_drv.bias_pk, &g_drv.bias_cap, (size_t)MB * 32 * 2))) return rc;
    HIP_TRY(hipMemsetAsync(g_drv.in_items, 0, (size_t)gi.items * 8, st), YOLO2_DMA_ERROR);
    HIP_TRY(hipMemsetAsync(g_drv.bias_pk, 0, (size_t)MB * 32 * 2, st), YOLO2_DMA_ERROR);
    HIP_TRY(hipMemsetAsync(g_drv.bound, 0, sizeof(int), st), YOLO2_DMA_ERROR);
    HIP_TRY(hipMemcpyAsync(g_drv.bias_pk, beta, (size_t)ofm_num * 2, hipMemcpyDeviceToDevice, st), YOLO2_DMA_ERROR);
    hipLaunchKernelGGL(k_ref_to_items, dim3(blocks_for((long)ifm_num * input_h * input_w, 256)), dim3(256), 0, st, in,
                       (short *)g_drv.in_items, ifm_num, input_h, input_w, (input_w + 7) & ~7, gi.Wp, gi.cg_stride);
    hipLaunchKernelGGL((k_repack_weights<short>), dim3(blocks_for(wpk_elems, 256)), dim3(256), 0, st, w, (short *)g_drv.wpk,
                       ifm_num, ofm_num, ksize * ksize);
    hipLaunchKernelGGL(k_weight_bound, dim3(std::min<unsigned>(blocks_for(wpk_elems / 4, 256), 1024)), dim3(256), 0, st,
                       (const short *)g_drv.wpk, wpk_elems / 4, g_drv.bound);
    // whole layer as one "block": its largest |w| decides whether the shift can be folded into the weights (form D)
    hipLaunchKernelGGL(k_weight_bound_mb, dim3(1), dim3(256), 0, st, (const short *)g_drv.wpk, wpk_elems / 4, g_drv.bound + 1,
                       g_drv.bound + 2);
    int hbound[3] = {0, 0, 0}, maxb = 0;
    HIP_TRY(hipMemcpy(hbound, g_drv.bound, sizeof(hbound), hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
    const int maxsum = hbound[0], maxabs = hbound[2];
    if ((rc = max_abs_i16_dev(beta, ofm_num, &maxb))) return rc;

    ConvPlan p;
    p.args.mb_list = nullptr;
    p.C = ifm_num; p.N = ofm_num; p.K = ksize; p.H = input_h; p.W = input_w; p.leaky = is_nl ? 1 : 0;
    p.Qw = qw; p.Qa_in = qa_in; p.Qa_out = qa_out; p.Qb = qb;
    p.path = choose_path(so, sb, maxsum, maxb, maxabs);
    if (p.path == 4) {   // this call's packed copy carries w * 2^(16-s)
        HIP_TRY(hipMemsetAsync(g_drv.bound + 3, 16 - so, 1, st), YOLO2_DMA_ERROR);
        hipLaunchKernelGGL(k_scale_weight_blocks, dim3(std::min<unsigned>(blocks_for(wpk_elems, 256), 256), 1), dim3(256), 0, st,
                           (short *)g_drv.wpk, wpk_elems, (const signed char *)(g_drv.bound + 3));
    }
    plan_conv(p, gi, go.cg_stride, kLead, go.CG);
    g_drv.last_path = p.path;
    launch_conv(p, (const int2 *)g_drv.in_items, (int2 *)g_drv.out_items, (const int2 *)g_drv.wpk, (const short *)g_drv.bias_pk, st);
    hipLaunchKernelGGL(k_items_to_ref, dim3(blocks_for((long)ofm_num * output_h * output_w, 256)), dim3(256), 0, st,
                       (const short *)g_drv.out_items, out, ofm_num, output_h, output_w, (output_w + 7) & ~7, go.Wp, go.PL,
                       go.cg_stride, 0);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return sync_with_timeout(st, timeout_ms);
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
    return drv_conv_locked(input_addr, output_addr, weight_addr, beta_addr, ifm_num, ofm_num, ksize, kstride, input_w, input_h,
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
    const int n = channels * output_h * output_w;
    hipLaunchKernelGGL((k_pool_ref<short>), dim3(blocks_for(n, 256)), dim3(256), 0, nullptr, (const short *)(uintptr_t)input_addr,
                       (short *)(uintptr_t)output_addr, channels, ksize, kstride, input_w, input_h, output_w, output_h,
                       (short)-32768);
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
    return drv_pool_locked(input_addr, output_addr, channels, ksize, kstride, input_w, input_h, output_w, output_h, padding, tm, tr,
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
    return g_drv.regs[offset / 4];
}
extern "C" void yolo2_write_reg(uint32_t offset, uint32_t value)
{
    std::lock_guard<std::mutex> lk(g_drv.mu);
    if (!drv_ready_locked() || offset >= sizeof(g_drv.regs)) return;
    if (offset != R_AP_CTRL) { g_drv.regs[offset / 4] = value; return; }
    if (!(value & AP_START)) return;
    const uint32_t *r = g_drv.regs;
    auto R = [&](uint32_t off) { return (int)r[off / 4]; };
    if (R(R_LTYPE) == 0)
        (void)drv_conv_locked(reg_get64(R_INPUT), reg_get64(R_OUTPUT), reg_get64(R_WEIGHT), reg_get64(R_BETA), R(R_IFM), R(R_OFM),
                              R(R_KSIZE), R(R_KSTRIDE), R(R_IN_W), R(R_IN_H), R(R_OUT_W), R(R_OUT_H), R(R_PAD), R(R_ISNL), R(R_ISBN),
                              R(R_TM), R(R_TN), R(R_TR), R(R_TC), R(R_OFM_BOUND), R(R_MLOOPS), R(R_MLOOPS_A1), 0, 0, 0, 0, 0, 0);
    else if (R(R_LTYPE) == 1)
        (void)drv_pool_locked(reg_get64(R_INPUT), reg_get64(R_OUTPUT), R(R_IFM), R(R_KSIZE), R(R_KSTRIDE), R(R_IN_W), R(R_IN_H),
                              R(R_OUT_W), R(R_OUT_H), R(R_PAD), R(R_TM), R(R_TR), R(R_TC), R(R_OFM_BOUND), R(R_MLOOPS),
                              R(R_MLOOPS_A1), 0);
    else
        (void)fail(YOLO2_ERROR, "register start: LayerType %d is not served by the accelerator", R(R_LTYPE));
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
    const int n = ofm_num * output_h * output_w;
    hipLaunchKernelGGL(k_conv_ref_f32, dim3(blocks_for(n, 256)), dim3(256), 0, nullptr, (const float *)(uintptr_t)input_addr,
                       (float *)(uintptr_t)output_addr, (const float *)(uintptr_t)weight_addr,
                       (const float *)(uintptr_t)beta_addr, ifm_num, ofm_num, ksize, kstride, input_w, input_h, output_w,
                       output_h, padding, is_nl ? 1 : 0);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return sync_with_timeout(nullptr, timeout_ms);
}

// ---------------------------------------------------------------------------- tier 2: whole network

struct Tensor {
    ActGeom g;
    int2 *d = nullptr;
};

// Staging for the host-buffer entries (run_frames / run_images): two buffer sets and three streams
// (upload, kernels, download), kept with the context and grown on demand so that a caller streaming
// chunk after chunk does not pay pinned-memory allocation per call.
struct PipeBufs {
    size_t host_in = 0, dev_in = 0;   // capacities in bytes (dev_in: raw image bytes, 0 for float frames)
    int batch = 0;
    uint8_t *hin[2] = {nullptr, nullptr}, *dbytes[2] = {nullptr, nullptr};
    float *din[2] = {nullptr, nullptr};
    int16_t *hout[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t e_in[2] = {nullptr, nullptr}, e_run[2] = {nullptr, nullptr}, e_out[2] = {nullptr, nullptr};
};

struct yolo2_hip_ctx {
    PipeBufs pipe;
    int device = 0;
    bool weights_loaded = false;
    short *wpk = nullptr;      // all layers, packed
    short *bias_pk = nullptr;  // all layers, padded to 32
    long wpk_off[YOLO2_N_CONV], bias_off[YOLO2_N_CONV];
    int maxsum[YOLO2_N_CONV], maxbias[YOLO2_N_CONV];
    std::vector<int> weight_q, bias_q, act_q;
    ConvPlan plan[32];                 // per conv layer: the launch covering most output-channel blocks
    std::vector<ConvPlan> extra[32];   // further launches for blocks that need another arithmetic form
    std::vector<int> maxsum_mb[YOLO2_N_CONV], maxbias_mb[YOLO2_N_CONV], maxabs_mb[YOLO2_N_CONV];
    std::vector<signed char> wscale_mb[YOLO2_N_CONV];   // log2 of the factor each block's packed weights currently carry (form D)
    int *mb_lists = nullptr;           // device: block index lists of all split layers
    // Lanes: a batch is run as two half-batches on two internal streams (forked from / joined to the
    // caller's stream with events).  Every layer is then two concurrent launches, and the idle tail of
    // one (a layer is only a few workgroup-generations long at batch 64) is filled by the other:
    // +4 % frames/s at batch 64.  A lane is a child context that shares the parent's weights.
    std::vector<yolo2_hip_ctx *> lanes;
    std::vector<int> lane_first;       // first frame of each lane within the batch
    std::vector<yolo2_hip_ctx *> f16_lanes;   // fp16 path: two half-batch lanes (share wh / biasf / w0f)
    bool is_lane = false, laned = false;
    hipStream_t lane_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fuse_pool[32] = {false};      // conv layer i stores the pooled tensor of layer i+1 itself (k_conv_i16_pool)
    ConvPlan fplan[32];                // the fused launches of those layers (plan / extra keep the unfused ones)
    std::vector<ConvPlan> fextra[32];
    int path_counts[YOLO2_N_CONV][5];
    int reorg_shift = 0, final_q = 0;
    int batch = 0;
    Tensor t_in, t_out[32], t_cat;
    // ---- fp16 MFMA path
    struct HalfTensor {
        int C = 0, Cp = 0, H = 0, W = 0, Wp = 0, PL = 0, B = 0;
        size_t items = 0;
        _Float16 *d = nullptr;
    };
    bool f16_loaded = false;
    _Float16 *wh = nullptr;
    float *biasf = nullptr;
    float *w0f = nullptr;  // layer 0: [27][32] fp32 weights + [32] bias for the fused conv0+pool kernel
    float *wf32 = nullptr, *bf32 = nullptr;   // the fp32 blobs as loaded (reference stream order), for the exact fp32 pass
    // ---- tiled exact fp32 path (kernels_f32.hpp): packed weights [mb][cg][tap][32][4] floats, items of 4 floats
    float *wpkf = nullptr, *biasf32_pk = nullptr;
    long wpkf_off[YOLO2_N_CONV], biasf32_off[YOLO2_N_CONV];
    struct FTensor {
        ActGeom g;
        float4 *d = nullptr;
    };
    int f32_batch = 0;
    FTensor f_in, f_out[32], f_cat;
    ConvPlan fp32_plan[32];
    long wh_off[YOLO2_N_CONV], biasf_off[YOLO2_N_CONV];
    int f16_batch = 0;
    HalfTensor h_in, h_out[32], h_cat;
    // per-layer device timing: a ring of event sets, one per profiled run (hipEvents on the
    // stream the kernels are launched on); the analogue of yolo2_inference.c:75-142
    static constexpr int kProfSlots = 32;
    bool prof = false;
    hipEvent_t ev[kProfSlots][33];
    bool ev_made = false;
    long prof_runs = 0;
};

extern "C" int yolo2_hip_create(int device, yolo2_hip_ctx **out)
{
    if (!out) return fail(YOLO2_ERROR, "null ctx pointer");
    if (device < 0 || device >= yolo2_hip_device_count())
        return fail(YOLO2_INIT_ERROR, "no HIP device %d (the GPU path has no CPU fallback)", device);
    HIP_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    yolo2_hip_ctx *c = new (std::nothrow) yolo2_hip_ctx();
    if (!c) return fail(YOLO2_ERROR, "out of host memory");
    c->device = device;
    *out = c;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_ctx_device(yolo2_hip_ctx *c) { return c ? c->device : -1; }

static void free_activations(yolo2_hip_ctx *c)
{
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

static void free_f16_activations(yolo2_hip_ctx *c)
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

static void free_f32_activations(yolo2_hip_ctx *c)
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

static void destroy_lanes(yolo2_hip_ctx *c)
{
    c->lane_first.clear();
    for (yolo2_hip_ctx *l : c->lanes) yolo2_hip_destroy(l);
    c->lanes.clear();
    c->laned = false;
}

static void pipe_free(PipeBufs &p)
{
    for (int k = 0; k < 2; ++k) {
        if (p.hin[k]) (void)hipHostFree(p.hin[k]);
        if (p.hout[k]) (void)hipHostFree(p.hout[k]);
        (void)hipFree(p.dbytes[k]); (void)hipFree(p.din[k]); (void)hipFree(p.dout[k]);
        if (p.e_in[k]) (void)hipEventDestroy(p.e_in[k]);
        if (p.e_run[k]) (void)hipEventDestroy(p.e_run[k]);
        if (p.e_out[k]) (void)hipEventDestroy(p.e_out[k]);
    }
    if (p.s_in) (void)hipStreamDestroy(p.s_in);
    if (p.s_run) (void)hipStreamDestroy(p.s_run);
    if (p.s_out) (void)hipStreamDestroy(p.s_out);
    p = PipeBufs();
}

static int pipe_ensure(PipeBufs &p, size_t host_in, size_t dev_in, int batch)
{
    if (p.s_in && p.host_in >= host_in && p.dev_in >= dev_in && p.batch >= batch) return YOLO2_SUCCESS;
    host_in = std::max(host_in, p.host_in);
    dev_in = std::max(dev_in, p.dev_in);
    batch = std::max(batch, p.batch);
    pipe_free(p);
    const size_t fbytes = (size_t)batch * YOLO2_FRAME_ELEMS * sizeof(float), rbytes = (size_t)batch * YOLO2_REGION_ELEMS * sizeof(int16_t);
    bool ok = true;
    for (int k = 0; k < 2 && ok; ++k) {
        ok = hipHostMalloc((void **)&p.hin[k], host_in, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&p.hout[k], rbytes, hipHostMallocDefault) == hipSuccess &&
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

extern "C" void yolo2_hip_destroy(yolo2_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    pipe_free(c->pipe);
    destroy_lanes(c);
    free_activations(c);
    if (c->lane_stream) (void)hipStreamDestroy(c->lane_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (yolo2_hip_ctx *l : c->f16_lanes) yolo2_hip_destroy(l);
    c->f16_lanes.clear();
    if (c->is_lane) {   // packed weights and biases belong to the parent
        c->wpk = nullptr;
        c->bias_pk = nullptr;
        c->wh = nullptr; c->biasf = nullptr; c->w0f = nullptr; c->wf32 = nullptr; c->bf32 = nullptr;
    }
    free_f16_activations(c);
    free_f32_activations(c);
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

// Resolve the per-layer Q values exactly like the layer loop does (yolo2_model.cpp:290-340, 379-399).
static int resolve_q(yolo2_hip_ctx *c)
{
    std::vector<int> lists;   // concatenated block lists of split layers
    const int na = (int)c->act_q.size();
    int current_Qa = na ? c->act_q[0] : 0, route24_q = 0, pending = -1, ord = 0;
    c->reorg_shift = 0;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        if (l.type == L_CONV) {
            ConvPlan &p = c->plan[i];
            p.C = l.c; p.N = l.n; p.K = l.size; p.H = l.h; p.W = l.w; p.leaky = l.leaky;
            p.Qa_in = ord < na ? c->act_q[ord] : current_Qa;
            p.Qa_out = ord + 1 < na ? c->act_q[ord + 1] : p.Qa_in;
            p.Qw = ord < (int)c->weight_q.size() ? c->weight_q[ord] : 0;
            p.Qb = ord < (int)c->bias_q.size() ? c->bias_q[ord] : 0;
            if (pending >= 0) p.Qa_in = pending;
            current_Qa = p.Qa_out;
            if (i == 24) route24_q = current_Qa;
            pending = -1;
            // arithmetic form per block of 32 output channels; launches are grouped by form
            const int MB = (l.n + 31) / 32, so = p.Qa_in + p.Qw - p.Qa_out, sb = p.Qb - p.Qa_out;
            std::vector<int> groups[5];
            std::vector<signed char> delta((size_t)MB, 0);
            bool rescale = false;
            if (c->wscale_mb[ord].size() != (size_t)MB) c->wscale_mb[ord].assign((size_t)MB, 0);
            for (int mb = 0; mb < MB; ++mb) {
                const int path = choose_path(so, sb, c->maxsum_mb[ord][mb], c->maxbias_mb[ord][mb],
                                             c->maxabs_mb[ord].empty() ? -1 : c->maxabs_mb[ord][mb]);
                groups[path].push_back(mb);
                const int want = path == 4 ? 16 - so : 0;   // form D blocks keep w * 2^(16-s) in the packed buffer
                delta[(size_t)mb] = (signed char)(want - c->wscale_mb[ord][(size_t)mb]);
                rescale |= delta[(size_t)mb] != 0;
                c->wscale_mb[ord][(size_t)mb] = (signed char)want;
            }
            if (rescale && !c->is_lane) {   // lanes share the parent's packed weights (and reach the same decisions)
                signed char *dd = nullptr;
                HIP_TRY(hipMalloc((void **)&dd, (size_t)MB), YOLO2_MMAP_ERROR);
                HIP_TRY(hipMemcpy(dd, delta.data(), (size_t)MB, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
                const long per_mb = packed_weight_elems(l.c, l.n, l.size) / MB;
                hipLaunchKernelGGL(k_scale_weight_blocks, dim3(std::min<unsigned>(blocks_for(per_mb, 256), 64), MB), dim3(256), 0,
                                   nullptr, c->wpk + c->wpk_off[ord], per_mb, dd);
                HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
                (void)hipFree(dd);
            }
            int dom = 0;
            for (int k = 0; k < 5; ++k) {
                c->path_counts[ord][k] = (int)groups[k].size();
                if (groups[k].size() > groups[dom].size()) dom = k;
            }
            c->extra[i].clear();
            p.args.mb_list = nullptr;
            p.mb_count = 0;
            p.path = dom;
            {   // split-K bounds with the layer-wide maxima: form A arithmetic (no int32 overflow), every
                // increment below 2^29 and the unclamped sum of one split below 2^30
                const ShiftSpec o = make_shift(so);
                const long long rnd = o.mag > 0 ? (1LL << (o.mag - 1)) : 0;
                const long long tmax = so >= 0 ? (((long long)c->maxsum[ord] * 32768 + rnd) >> o.mag) : (1LL << 40);
                const long long steps = (long long)((l.c + 3) / 4 / 4 + 1) * l.size * l.size;
                const bool okA = choose_path(so, sb, c->maxsum[ord], c->maxbias[ord]) != 2;
                p.splitk_ok = okA && tmax < (1LL << 29) && tmax * steps < (1LL << 30);
            }
            if ((int)groups[dom].size() != MB) {
                p.mb_count = (int)groups[dom].size();
                p.args.mb_list = (const int *)(uintptr_t)lists.size();   // offset for now, pointer once uploaded
                lists.insert(lists.end(), groups[dom].begin(), groups[dom].end());
                for (int k = 0; k < 5; ++k) {
                    if (k == dom || groups[k].empty()) continue;
                    ConvPlan e = p;
                    e.path = k;
                    e.mb_count = (int)groups[k].size();
                    e.args.mb_list = (const int *)(uintptr_t)lists.size();
                    lists.insert(lists.end(), groups[k].begin(), groups[k].end());
                    c->extra[i].push_back(e);
                }
            }
            ord++;
        } else if (l.type == L_REORG) {
            if (route24_q > 0) {
                const int target = std::min(route24_q, current_Qa);
                c->reorg_shift = current_Qa - target;
                if (c->reorg_shift != 0) current_Qa = target;
                pending = current_Qa;
            }
        }
    }
    c->final_q = current_Qa;
    if (c->mb_lists) (void)hipFree(c->mb_lists);
    c->mb_lists = nullptr;
    if (!lists.empty()) {
        HIP_TRY(hipMalloc((void **)&c->mb_lists, lists.size() * sizeof(int)), YOLO2_MMAP_ERROR);
        HIP_TRY(hipMemcpy(c->mb_lists, lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
        for (int i = 0; i < 32; ++i) {
            if (kNet[i].type != L_CONV || !c->plan[i].mb_count) continue;
            c->plan[i].args.mb_list = c->mb_lists + (uintptr_t)c->plan[i].args.mb_list;
            for (auto &e : c->extra[i]) e.args.mb_list = c->mb_lists + (uintptr_t)e.args.mb_list;
        }
    }
    return YOLO2_SUCCESS;
}

static int load_common(yolo2_hip_ctx *c, const short *w_dev, size_t n_weights, const short *b_dev, size_t n_bias,
                       const int32_t *weight_q, int n_wq, const int32_t *bias_q, int n_bq, const int32_t *act_q, int n_aq)
{
    if (n_weights < YOLO2_N_WEIGHTS) return fail(YOLO2_ERROR, "weights blob too small (%zu < %d)", n_weights, YOLO2_N_WEIGHTS);
    if (n_bias < YOLO2_N_BIAS) return fail(YOLO2_ERROR, "bias blob too small (%zu < %d)", n_bias, YOLO2_N_BIAS);
    if (n_wq < YOLO2_N_CONV || n_bq < YOLO2_N_CONV) return fail(YOLO2_ERROR, "Q tables too small for conv layers");
    if (n_aq < 1) return fail(YOLO2_ERROR, "Activation Q table (iofm_Q.bin) is required for int16 inference.");
    destroy_lanes(c);   // they alias the weight buffers that are about to be replaced
    c->weight_q.assign(weight_q, weight_q + n_wq);
    c->bias_q.assign(bias_q, bias_q + n_bq);
    c->act_q.assign(act_q, act_q + n_aq);

    long wtot = 0, btot = 0;
    int ord = 0;
    for (int i = 0; i < 32; ++i)
        if (kNet[i].type == L_CONV) {
            c->wpk_off[ord] = wtot;
            c->bias_off[ord] = btot;
            wtot += packed_weight_elems(kNet[i].c, kNet[i].n, kNet[i].size);
            btot += (long)((kNet[i].n + 31) / 32) * 32;
            ord++;
        }
    if (c->wpk) (void)hipFree(c->wpk);
    if (c->bias_pk) (void)hipFree(c->bias_pk);
    c->wpk = c->bias_pk = nullptr;
    HIP_TRY(hipMalloc((void **)&c->wpk, (size_t)wtot * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&c->bias_pk, (size_t)btot * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(c->bias_pk, 0, (size_t)btot * 2), YOLO2_DMA_ERROR);
    int *bound = nullptr, *bound_mb = nullptr, *bound_abs = nullptr;
    int mb_total = 0;
    for (int i = 0; i < 32; ++i)
        if (kNet[i].type == L_CONV) mb_total += (kNet[i].n + 31) / 32;
    HIP_TRY(hipMalloc((void **)&bound, sizeof(int) * YOLO2_N_CONV), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&bound_mb, sizeof(int) * mb_total), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&bound_abs, sizeof(int) * mb_total), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(bound, 0, sizeof(int) * YOLO2_N_CONV), YOLO2_DMA_ERROR);
    int mb_off = 0;
    std::vector<int> mb_offs;
    std::vector<short> hb(YOLO2_N_BIAS);
    HIP_TRY(hipMemcpy(hb.data(), b_dev, (size_t)YOLO2_N_BIAS * 2, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
    long woff = 0, boff = 0;
    ord = 0;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        if (l.type != L_CONV) continue;
        const long n = packed_weight_elems(l.c, l.n, l.size);
        hipLaunchKernelGGL((k_repack_weights<short>), dim3(blocks_for(n, 256)), dim3(256), 0, nullptr, w_dev + woff,
                           c->wpk + c->wpk_off[ord], l.c, l.n, l.size * l.size);
        hipLaunchKernelGGL(k_weight_bound, dim3(std::min<unsigned>(blocks_for(n / 4, 256), 1024)), dim3(256), 0, nullptr,
                           (const short *)(c->wpk + c->wpk_off[ord]), n / 4, bound + ord);
        const int MB = (l.n + 31) / 32;
        hipLaunchKernelGGL(k_weight_bound_mb, dim3(MB), dim3(256), 0, nullptr, (const short *)(c->wpk + c->wpk_off[ord]),
                           n / 4 / MB, bound_mb + mb_off, bound_abs + mb_off);
        c->wscale_mb[ord].assign((size_t)MB, 0);   // freshly packed: unscaled
        mb_offs.push_back(mb_off);
        mb_off += MB;
        HIP_TRY(hipMemcpyAsync(c->bias_pk + c->bias_off[ord], b_dev + boff, (size_t)l.n * 2, hipMemcpyDeviceToDevice, nullptr),
                YOLO2_DMA_ERROR);
        int mb = 0;
        c->maxbias_mb[ord].assign(MB, 0);
        for (int k = 0; k < l.n; ++k) {
            const int v = std::abs((int)hb[boff + k]);
            mb = std::max(mb, v);
            c->maxbias_mb[ord][k / 32] = std::max(c->maxbias_mb[ord][k / 32], v);
        }
        c->maxbias[ord] = mb;
        woff += yolo2_weight_len[ord];
        boff += yolo2_bias_len[ord];
        ord++;
    }
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    HIP_TRY(hipMemcpy(c->maxsum, bound, sizeof(int) * YOLO2_N_CONV, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
    {
        std::vector<int> hm(mb_total), ha(mb_total);
        HIP_TRY(hipMemcpy(hm.data(), bound_mb, sizeof(int) * mb_total, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
        HIP_TRY(hipMemcpy(ha.data(), bound_abs, sizeof(int) * mb_total, hipMemcpyDeviceToHost), YOLO2_DMA_ERROR);
        int o = 0;
        for (int i = 0; i < 32; ++i)
            if (kNet[i].type == L_CONV) {
                const int MB = (kNet[i].n + 31) / 32;
                c->maxsum_mb[o].assign(hm.begin() + mb_offs[o], hm.begin() + mb_offs[o] + MB);
                c->maxabs_mb[o].assign(ha.begin() + mb_offs[o], ha.begin() + mb_offs[o] + MB);
                o++;
            }
    }
    (void)hipFree(bound);
    (void)hipFree(bound_mb);
    (void)hipFree(bound_abs);
    {
        const int rq = resolve_q(c);
        if (rq) return rq;
    }
    c->weights_loaded = true;
    if (c->batch) {  // re-plan for the new Q values
        const int b = c->batch;
        c->batch = 0;
        return yolo2_hip_set_batch(c, b);
    }
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_load_weights_int16_dev(yolo2_hip_ctx *c, uint64_t weights_reorg_dev, size_t n_weights,
                                                uint64_t bias_dev, size_t n_bias, const int32_t *weight_q, int n_weight_q,
                                                const int32_t *bias_q, int n_bias_q, const int32_t *act_q, int n_act_q)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (!weights_reorg_dev || !bias_dev || !weight_q || !bias_q || !act_q) return fail(YOLO2_ERROR, "null argument");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    return load_common(c, (const short *)(uintptr_t)weights_reorg_dev, n_weights, (const short *)(uintptr_t)bias_dev, n_bias,
                       weight_q, n_weight_q, bias_q, n_bias_q, act_q, n_act_q);
}

extern "C" int yolo2_hip_load_weights_int16(yolo2_hip_ctx *c, const int16_t *weights_reorg, size_t n_weights,
                                            const int16_t *bias, size_t n_bias, const int32_t *weight_q, int n_weight_q,
                                            const int32_t *bias_q, int n_bias_q, const int32_t *act_q, int n_act_q)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (!weights_reorg || !bias || !weight_q || !bias_q || !act_q) return fail(YOLO2_ERROR, "null argument");
    if (n_weights < YOLO2_N_WEIGHTS) return fail(YOLO2_ERROR, "weights file too small");
    if (n_bias < YOLO2_N_BIAS) return fail(YOLO2_ERROR, "bias file too small");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    short *wd = nullptr, *bd = nullptr;
    HIP_TRY(hipMalloc((void **)&wd, (size_t)YOLO2_N_WEIGHTS * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&bd, (size_t)YOLO2_N_BIAS * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemcpy(wd, weights_reorg, (size_t)YOLO2_N_WEIGHTS * 2, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    HIP_TRY(hipMemcpy(bd, bias, (size_t)YOLO2_N_BIAS * 2, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    const int rc = load_common(c, wd, YOLO2_N_WEIGHTS, bd, YOLO2_N_BIAS, weight_q, n_weight_q, bias_q, n_bias_q, act_q, n_act_q);
    (void)hipDeviceSynchronize();
    (void)hipFree(wd);
    (void)hipFree(bd);
    return rc;
}

extern "C" int yolo2_hip_layer_path(yolo2_hip_ctx *c, int ord)
{
    if (!c || !c->weights_loaded || ord < 0 || ord >= YOLO2_N_CONV) return -1;
    int o = 0;
    for (int i = 0; i < 32; ++i)
        if (kNet[i].type == L_CONV) {
            if (o == ord) return c->plan[i].path;
            o++;
        }
    return -1;
}

extern "C" int yolo2_hip_layer_path_counts(yolo2_hip_ctx *c, int ord, int counts[5])
{
    if (!c || !c->weights_loaded || ord < 0 || ord >= YOLO2_N_CONV || !counts) return YOLO2_ERROR;
    for (int k = 0; k < 5; ++k) counts[k] = c->path_counts[ord][k];
    return YOLO2_SUCCESS;
}

static int alloc_tensor(Tensor &t, int C, int H, int W, int B)
{
    t.g = make_geom(C, H, W, B);
    HIP_TRY(hipMalloc((void **)&t.d, (size_t)t.g.items * 8), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(t.d, 0, (size_t)t.g.items * 8), YOLO2_DMA_ERROR);  // the zeros ARE the conv padding
    return YOLO2_SUCCESS;
}

// Pixels-per-lane (P) decides tile count, occupancy and how evenly a layer's workgroups divide
// over the 256 CUs; the best value depends on layer shape and batch.  Time each candidate once
// per layer on the layer's own buffers (integer kernels: timing does not depend on the data) and
// keep the fastest.  ~0.2 s at batch 64; disable with YOLO2_AUTOTUNE=0.
static int autotune(yolo2_hip_ctx *c)
{
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), YOLO2_ERROR);
    HIP_TRY(hipEventCreate(&e1), YOLO2_ERROR);
    // In a real pass every layer meets its weights cold in L2 (the other layers' 100 MB went through
    // since), so each timed launch is preceded by a 64 MB fill that evicts the L2s.  Without it a
    // repeated launch finds its weights in L2 and, at small batch, the latency of the per-tap
    // scalar weight loads - exactly what the split-K kernel avoids - is not seen.
    const size_t flush_bytes = (size_t)64 << 20;
    void *flush = nullptr;
    HIP_TRY(hipMalloc(&flush, flush_bytes), YOLO2_MMAP_ERROR);
    int ord = 0;
    for (int i = 0; i < 32; ++i) {
        if (kNet[i].type != L_CONV) continue;
        const Tensor &tin = i == 0 ? c->t_in : (i == 26 ? c->t_out[16] : (i == 29 ? c->t_cat : c->t_out[i - 1]));
        const Tensor &tout = c->t_out[i];
        const long out_base = kLead + (i == 24 ? (long)64 * tout.g.cg_stride : 0);
        const int CGout = (kNet[i].n + 3) / 4;
        std::vector<ConvPlan *> subs{&c->plan[i]};
        for (auto &e : c->extra[i]) subs.push_back(&e);
        for (ConvPlan *sp : subs) {
            float best = 1e30f;
            int bestP = sp->P, bestPad = 0;
            int bestSplit = 0, bestPP = 1;
            const char *fs = getenv("YOLO2_SPLITK");   // test hook: 0 = never, 1 = wherever legal, unset = tuned
            // candidates: pixels per lane x workgroups-per-CU cap (160 KiB LDS / cap), and the split-K kernel
            for (int cfg = 0; cfg < 16; ++cfg) {   // 0..11: tile shapes; 12, 13: split-K with 4 / 8 splits; 14: 4 splits, 2 pixels per lane
                const int P = cfg >= 12 ? 1 : 8 >> (cfg & 3);
                const int pad = cfg >= 12 ? 0 : ((cfg >> 2) == 0 ? 0 : ((cfg >> 2) == 1 ? 160 * 1024 / 6 : 160 * 1024 / 4));
                if (pad && P > 2) continue;   // the cap only matters for the small-tile, 8-waves/SIMD shapes
                ConvPlan cand = *sp;
                cand.lds_pad = pad;
                cand.splitk = 0;
                cand.splitk_pp = 1;
                if (cfg >= 12) {
                    if (!sp->splitk_ok || !c->extra[i].empty() || (fs && atoi(fs) == 0)) continue;
                    cand.splitk = cfg == 13 ? 8 : 4;
                    cand.splitk_pp = cfg == 14 ? 2 : (cfg == 15 ? 4 : 1);
                } else if (fs && atoi(fs) == 1 && sp->splitk_ok && c->extra[i].empty()) {
                    ConvPlan probe = *sp;
                    probe.splitk = 4;
                    plan_conv(probe, tin.g, tout.g.cg_stride, out_base, CGout, 1);
                    if (probe.splitk) continue;   // forced: skip the ordinary candidates where split-K is available
                }
                plan_conv(cand, tin.g, tout.g.cg_stride, out_base, CGout, P);
                if (cfg >= 12 && !cand.splitk) continue;
                if (cfg == 14 && cand.splitk_pp != 2) continue;
                if (cfg == 15 && cand.splitk_pp != 4) continue;
                if (cand.P != P) continue;  // not available for this path / shape
                float tmin = 1e30f;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipMemsetAsync(flush, rep, flush_bytes, nullptr);
                    (void)hipEventRecord(e0, nullptr);
                    launch_conv(cand, tin.d, tout.d, (const int2 *)(c->wpk + c->wpk_off[ord]), c->bias_pk + c->bias_off[ord], nullptr);
                    (void)hipEventRecord(e1, nullptr);
                    HIP_TRY(hipEventSynchronize(e1), YOLO2_ERROR);
                    float t = 0;
                    HIP_TRY(hipEventElapsedTime(&t, e0, e1), YOLO2_ERROR);
                    tmin = std::min(tmin, t);
                }
                if (getenv("YOLO2_VERBOSE"))
                    fprintf(stderr, "[yolo2_hip] tune L%d path %d: P=%d pad=%d splitk=%d grid=(%u,%u) %.1f us\n", i, cand.path, P, pad,
                            cand.splitk, cand.grid.x, cand.grid.y, tmin * 1e3);
                if (tmin < best) { best = tmin; bestP = P; bestPad = pad; bestSplit = cand.splitk; bestPP = cand.splitk_pp; }
            }
            sp->lds_pad = bestPad;
            sp->splitk = bestSplit;
            sp->splitk_pp = bestPP;
            plan_conv(*sp, tin.g, tout.g.cg_stride, out_base, CGout, bestP);
        }
        ord++;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(flush);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

static int set_batch_single(yolo2_hip_ctx *c, int batch);

static void launch_maxpool(const Tensor &tin, const Tensor &tout, int B, hipStream_t st)
{
    const ActGeom &gi = tin.g, &go = tout.g;
    const long n = (long)go.CG * B * go.H * go.W;
    hipLaunchKernelGGL(k_maxpool2, dim3(blocks_for(n, 256)), dim3(256), 0, st, tin.d, tout.d, go.CG, B, go.H, go.W, gi.Wp, gi.PL,
                       go.Wp, go.PL);
}

// Conv layers followed by a 2x2 pool (0, 2, 6, 10, 16) may run as ONE kernel that stores the pooled tensor
// (k_conv_i16_pool); layer 16 also feeds the route to layer 26, so it stores the full-resolution tensor too.
// Legal when every launch of the layer runs a packed-accumulator form (C / D).  timed: keep whichever of
// {conv launches + k_maxpool2, fused launches} is faster on this batch (cold L2, like autotune); otherwise fuse
// wherever legal.  YOLO2_NO_POOLFUSE=1 disables, YOLO2_POOLFUSE=1 forces it wherever legal.
static int setup_pool_fusion(yolo2_hip_ctx *c, bool timed, bool default_on)
{
    for (bool &f : c->fuse_pool) f = false;
    if (getenv("YOLO2_NO_POOLFUSE")) return YOLO2_SUCCESS;
    const char *fe = getenv("YOLO2_POOLFUSE");
    const bool force = fe && atoi(fe) == 1;
    if (!default_on && !force) return YOLO2_SUCCESS;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void *flush = nullptr;
    const size_t flush_bytes = (size_t)64 << 20;
    if (timed && !force) {
        HIP_TRY(hipEventCreate(&e0), YOLO2_ERROR);
        HIP_TRY(hipEventCreate(&e1), YOLO2_ERROR);
        HIP_TRY(hipMalloc(&flush, flush_bytes), YOLO2_MMAP_ERROR);
    }
    int ord = 0;
    for (int i = 0; i < 31; ++i) {
        if (kNet[i].type != L_CONV) continue;
        const int o = ord++;
        if (kNet[i + 1].type != L_MAX) continue;
        const Tensor &tin = i == 0 ? c->t_in : c->t_out[i - 1];
        const Tensor &tout = c->t_out[i], &tpool = c->t_out[i + 1];
        const int full = i == 16 ? 1 : 0;
        ConvPlan fp = c->plan[i];
        bool ok = plan_conv_pool(fp, tin.g, tpool.g, full);
        std::vector<ConvPlan> fx;
        for (const auto &e : c->extra[i]) {
            ConvPlan fe2 = e;
            ok = ok && plan_conv_pool(fe2, tin.g, tpool.g, full);
            fx.push_back(fe2);
        }
        if (!ok) continue;
        c->fplan[i] = fp;
        c->fextra[i] = fx;
        if (!timed || force) { c->fuse_pool[i] = true; continue; }
        const int2 *wp = (const int2 *)(c->wpk + c->wpk_off[o]);
        const short *bp = c->bias_pk + c->bias_off[o];
        float best[2] = {1e30f, 1e30f};
        for (int variant = 0; variant < 2; ++variant)
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemsetAsync(flush, rep, flush_bytes, nullptr);
                (void)hipEventRecord(e0, nullptr);
                if (variant == 0) {
                    launch_conv(c->plan[i], tin.d, tout.d, wp, bp, nullptr);
                    for (const auto &e : c->extra[i]) launch_conv(e, tin.d, tout.d, wp, bp, nullptr);
                    launch_maxpool(tout, tpool, c->batch, nullptr);
                } else {
                    launch_conv(fp, tin.d, tout.d, wp, bp, nullptr, tpool.d);
                    for (const auto &e : fx) launch_conv(e, tin.d, tout.d, wp, bp, nullptr, tpool.d);
                }
                (void)hipEventRecord(e1, nullptr);
                HIP_TRY(hipEventSynchronize(e1), YOLO2_ERROR);
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, e0, e1), YOLO2_ERROR);
                best[variant] = std::min(best[variant], t);
            }
        // Fused unless the separate kernels are clearly faster: within timing noise the fused form wins on traffic, and a choice that
        // flips from run to run changes which layers the bench's per-kernel objects describe.
        c->fuse_pool[i] = best[1] < best[0] * 1.05f;
        if (getenv("YOLO2_VERBOSE"))
            fprintf(stderr, "[yolo2_hip] L%d conv+pool: separate %.1f us, fused %.1f us -> %s\n", i, best[0] * 1e3, best[1] * 1e3,
                    c->fuse_pool[i] ? "fused" : "separate");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (flush) (void)hipFree(flush);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

static int make_lane(yolo2_hip_ctx *p, yolo2_hip_ctx **out)
{
    yolo2_hip_ctx *l = new (std::nothrow) yolo2_hip_ctx();
    if (!l) return fail(YOLO2_ERROR, "out of host memory");
    l->device = p->device;
    l->is_lane = true;
    l->wpk = p->wpk;
    l->bias_pk = p->bias_pk;
    memcpy(l->wpk_off, p->wpk_off, sizeof(p->wpk_off));
    memcpy(l->bias_off, p->bias_off, sizeof(p->bias_off));
    memcpy(l->maxsum, p->maxsum, sizeof(p->maxsum));
    memcpy(l->maxbias, p->maxbias, sizeof(p->maxbias));
    for (int o = 0; o < YOLO2_N_CONV; ++o) {
        l->maxsum_mb[o] = p->maxsum_mb[o];
        l->maxbias_mb[o] = p->maxbias_mb[o];
        l->maxabs_mb[o] = p->maxabs_mb[o];
    }
    l->weight_q = p->weight_q;
    l->bias_q = p->bias_q;
    l->act_q = p->act_q;
    int rc = resolve_q(l);
    if (rc == YOLO2_SUCCESS && hipStreamCreateWithFlags(&l->lane_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(YOLO2_ERROR, "hipStreamCreate failed");
    if (rc == YOLO2_SUCCESS && hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming) != hipSuccess) rc = fail(YOLO2_ERROR, "hipEventCreate failed");
    if (rc) { yolo2_hip_destroy(l); return rc; }
    l->weights_loaded = true;
    *out = l;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_set_batch(yolo2_hip_ctx *c, int batch)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (batch <= 0 || batch > 4096) return fail(YOLO2_ERROR, "batch %d out of range", batch);
    if (!c->weights_loaded) return fail(YOLO2_ERROR, "load weights before set_batch");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    // Lanes: three for batches 48..127 (measured +2 % over two at batch 64: one more launch to fill each tail; at
    // batch 256 two are 1 % better), two otherwise from batch 16; YOLO2_LANES=n overrides.  Sizes differ by at
    // most one frame (64 = 22 + 21 + 21).
    int nl = (batch >= 48 && batch < 128) ? 3 : 2;
    if (const char *e = getenv("YOLO2_LANES")) nl = std::max(1, atoi(e));
    const bool want_lanes = !c->is_lane && nl > 1 && batch >= 8 * nl && !getenv("YOLO2_NO_LANES");
    if (!want_lanes) {
        if (c->laned) c->batch = 0;   // a laned parent owns no activation tensors: force set_batch_single to allocate
        destroy_lanes(c);
        return set_batch_single(c, batch);
    }
    if (c->laned && c->batch == batch && (int)c->lanes.size() == nl) return YOLO2_SUCCESS;
    destroy_lanes(c);
    free_activations(c);
    if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), YOLO2_ERROR);
    c->lane_first.clear();
    int first = 0;
    for (int i = 0; i < nl; ++i) {
        yolo2_hip_ctx *l = nullptr;
        // The remainder goes to the LAST lanes: the first lane's launches are enqueued first in every step and it is the one that
        // finishes last (kernel trace: by 0.6-3 ms of a 20 ms step at batch 64), so it gets the smaller share.
        int frames = batch / nl + (i >= nl - batch % nl ? 1 : 0);
        if (const char *sp = getenv("YOLO2_LANE_SPLIT")) {   // diagnostic: "20,22,22" (must sum to the batch)
            std::vector<int> v;
            for (const char *q = sp; *q;) { v.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
            int sum = 0;
            for (int x : v) sum += x;
            if ((int)v.size() == nl && sum == batch) frames = v[i];
        }
        int rc = make_lane(c, &l);
        if (rc == YOLO2_SUCCESS) {
            c->lanes.push_back(l);
            c->lane_first.push_back(first);
            rc = set_batch_single(l, frames);
        }
        if (rc) { destroy_lanes(c); return rc; }
        first += frames;
    }
    if (c->prof) (void)yolo2_hip_set_profiling(c->lanes[0], 1);
    c->batch = batch;
    c->laned = true;
    return YOLO2_SUCCESS;
}

static int set_batch_single(yolo2_hip_ctx *c, int batch)
{
    if (c->batch != batch) {
        free_activations(c);
        int rc;
        if ((rc = alloc_tensor(c->t_in, 3, 416, 416, batch))) return rc;
        if ((rc = alloc_tensor(c->t_cat, 1280, 13, 13, batch))) return rc;
        for (int i = 0; i < 31; ++i) {
            const LayerDesc &l = kNet[i];
            if (l.type == L_CONV && i != 24) {
                if ((rc = alloc_tensor(c->t_out[i], l.n, l.h, l.w, batch))) return rc;
            } else if (l.type == L_MAX) {
                if ((rc = alloc_tensor(c->t_out[i], l.c, l.h / 2, l.w / 2, batch))) return rc;
            }
        }
        c->t_out[24] = c->t_cat;  // conv-24 output and the reorg output live in the concat tensor
        c->t_out[27] = c->t_cat;  // (yolo2_model.cpp:97-104 does the same by arena placement)
        c->batch = batch;
    }
    for (int i = 0; i < 32; ++i) {
        if (kNet[i].type != L_CONV) continue;
        const Tensor &tin = i == 0 ? c->t_in : (i == 26 ? c->t_out[16] : (i == 29 ? c->t_cat : c->t_out[i - 1]));
        const Tensor &tout = c->t_out[i];
        const long out_base = kLead + (i == 24 ? (long)64 * tout.g.cg_stride : 0);
        const int CGout = (kNet[i].n + 3) / 4;
        plan_conv(c->plan[i], tin.g, tout.g.cg_stride, out_base, CGout);
        for (auto &e : c->extra[i]) plan_conv(e, tin.g, tout.g.cg_stride, out_base, CGout);
    }
    const char *fp = getenv("YOLO2_FORCE_P");  // test hook: one pixels-per-lane value for every layer
    if (fp && atoi(fp) > 0) {
        for (int i = 0; i < 32; ++i) {
            if (kNet[i].type != L_CONV) continue;
            const Tensor &tin = i == 0 ? c->t_in : (i == 26 ? c->t_out[16] : (i == 29 ? c->t_cat : c->t_out[i - 1]));
            const Tensor &tout = c->t_out[i];
            plan_conv(c->plan[i], tin.g, tout.g.cg_stride, kLead + (i == 24 ? (long)64 * tout.g.cg_stride : 0),
                      (kNet[i].n + 3) / 4, atoi(fp));
            for (auto &e : c->extra[i])
                plan_conv(e, tin.g, tout.g.cg_stride, kLead + (i == 24 ? (long)64 * tout.g.cg_stride : 0), (kNet[i].n + 3) / 4, atoi(fp));
        }
        HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);   // (the tensors' zero fills ran on the null stream)
        return setup_pool_fusion(c, false, false);      // fixed tile shapes: fusion only on request (YOLO2_POOLFUSE=1)
    }
    const char *at = getenv("YOLO2_AUTOTUNE");
    if (!(at && at[0] == '0')) {
        const int rc = autotune(c);
        return rc ? rc : setup_pool_fusion(c, true, true);
    }
    HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
    return setup_pool_fusion(c, false, true);
}

extern "C" int yolo2_hip_num_lanes(yolo2_hip_ctx *c) { return c && c->laned ? (int)c->lanes.size() : 1; }
extern "C" int yolo2_hip_num_lanes_fp16(yolo2_hip_ctx *c) { return c && !c->f16_lanes.empty() ? (int)c->f16_lanes.size() : 1; }

static int ensure_prof_events(yolo2_hip_ctx *c)
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
        const int rc = ensure_prof_events(c);
        if (rc) return rc;
    }
    c->prof = enable != 0;
    c->prof_runs = 0;  // (re)start the averaging window
    if (!c->f16_lanes.empty()) return yolo2_hip_set_profiling(c->f16_lanes[0], enable);   // fp16 lanes: lane 0 is reported
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_layer_times_ms(yolo2_hip_ctx *c, float *ms32)
{
    if (!c || !ms32) return fail(YOLO2_ERROR, "null argument");
    if (c->laned) return yolo2_hip_layer_times_ms(c->lanes[0], ms32);
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

extern "C" int yolo2_hip_layer_pool_fused(yolo2_hip_ctx *c, int layer_idx)
{
    if (!c || layer_idx < 0 || layer_idx > 31 || !c->batch) return 0;
    if (c->laned) return yolo2_hip_layer_pool_fused(c->lanes[0], layer_idx);
    return c->fuse_pool[layer_idx] ? 1 : 0;
}

extern "C" int yolo2_hip_conv_launch_info(yolo2_hip_ctx *c, int ord, int *grid_x, int *grid_y, int *block, int *lds_bytes,
                                          int *ppl)
{
    if (!c || !c->batch) return fail(YOLO2_ERROR, "set_batch first");
    if (c->laned) return yolo2_hip_conv_launch_info(c->lanes[0], ord, grid_x, grid_y, block, lds_bytes, ppl);
    int o = 0;
    for (int i = 0; i < 32; ++i)
        if (kNet[i].type == L_CONV) {
            if (o == ord) {
                const ConvPlan &pl = c->fuse_pool[i] ? c->fplan[i] : c->plan[i];
                if (grid_x) *grid_x = pl.grid.x;
                if (grid_y) *grid_y = pl.grid.y;
                if (block) *block = 256;
                if (lds_bytes) *lds_bytes = pl.lds_bytes;
                if (ppl) *ppl = pl.splitk ? 0 : pl.P;
                return YOLO2_SUCCESS;
            }
            o++;
        }
    return fail(YOLO2_ERROR, "bad conv ordinal %d", ord);
}

extern "C" int yolo2_hip_run_batch_int16(yolo2_hip_ctx *c, uint64_t frames_dev, int batch, uint64_t region_dev,
                                         int *final_q, void *stream)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (!c->weights_loaded) return fail(YOLO2_ERROR, "weights not loaded");
    if (!frames_dev || !region_dev) return fail(YOLO2_ERROR, "null buffer address");
    if (batch != c->batch) {
        const int rc = yolo2_hip_set_batch(c, batch);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    if (c->laned) {   // fork the two half-batches onto the lane streams, join back into the caller's stream
        const int nl = (int)c->lanes.size();
        HIP_TRY(hipEventRecord(c->ev_fork, st), YOLO2_ERROR);
        for (int i = 0; i < nl; ++i) {
            yolo2_hip_ctx *l = c->lanes[i];
            const uint64_t first = (uint64_t)c->lane_first[i];
            HIP_TRY(hipStreamWaitEvent(l->lane_stream, c->ev_fork, 0), YOLO2_ERROR);
            const int rc = yolo2_hip_run_batch_int16(l, frames_dev + first * YOLO2_FRAME_ELEMS * sizeof(float), l->batch,
                                                     region_dev + first * YOLO2_REGION_ELEMS * sizeof(int16_t), final_q,
                                                     l->lane_stream);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(l->ev_join, l->lane_stream), YOLO2_ERROR);
            HIP_TRY(hipStreamWaitEvent(st, l->ev_join, 0), YOLO2_ERROR);
        }
        c->final_q = c->lanes[0]->final_q;
        return YOLO2_SUCCESS;
    }
    const float *frames = (const float *)(uintptr_t)frames_dev;
    short *region = (short *)(uintptr_t)region_dev;
    const int B = batch;
    const float scale = ldexpf(1.0f, c->act_q[0]);

    if (c->prof) {
        const int rc = ensure_prof_events(c);
        if (rc) return rc;
    }
    hipEvent_t *ev = c->prof ? c->ev[c->prof_runs % yolo2_hip_ctx::kProfSlots] : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[0], st), YOLO2_ERROR);
    {  // input quantise + pack (yolo2_model.cpp:257-278); its time is booked to layer 0
        const ActGeom &g = c->t_in.g;
        hipLaunchKernelGGL(k_pack_input, dim3(blocks_for((long)B * g.H * g.W, 256)), dim3(256), 0, st, frames, c->t_in.d, B,
                           g.H, g.W, g.Wp, g.PL, scale);
    }
    int ord = 0;
    const Tensor *cur = &c->t_in;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        switch (l.type) {
        case L_CONV: {
            const Tensor *tin = i == 26 ? &c->t_out[16] : (i == 29 ? &c->t_cat : cur);
            const int2 *wp = (const int2 *)(c->wpk + c->wpk_off[ord]);
            const short *bp = c->bias_pk + c->bias_off[ord];
            if (c->fuse_pool[i]) {   // conv + leaky + pool in one kernel: stores layer i+1's tensor (and layer 16's own)
                launch_conv(c->fplan[i], tin->d, c->t_out[i].d, wp, bp, st, c->t_out[i + 1].d);
                for (const auto &e : c->fextra[i]) launch_conv(e, tin->d, c->t_out[i].d, wp, bp, st, c->t_out[i + 1].d);
            } else {
                launch_conv(c->plan[i], tin->d, c->t_out[i].d, wp, bp, st);
                for (const auto &e : c->extra[i])   // blocks of this layer that need another arithmetic form
                    launch_conv(e, tin->d, c->t_out[i].d, wp, bp, st);
            }
            cur = &c->t_out[i];
            ord++;
            break;
        }
        case L_MAX: {
            if (!c->fuse_pool[i - 1]) launch_maxpool(*cur, c->t_out[i], B, st);
            cur = &c->t_out[i];
            break;
        }
        case L_REORG: {
            const ActGeom &gi = cur->g, &go = c->t_cat.g;
            hipLaunchKernelGGL(k_reorg, dim3(blocks_for((long)B * 256 * 169, 256)), dim3(256), 0, st, (const short *)cur->d,
                               (short *)c->t_cat.d, B, gi.Wp, gi.PL, gi.cg_stride, go.Wp, go.PL, go.cg_stride, c->reorg_shift);
            cur = &c->t_cat;
            break;
        }
        case L_ROUTE:
            break;  // concat by placement (yolo2_model.cpp:404-405)
        case L_REGION: {
            const ActGeom &g = cur->g;
            hipLaunchKernelGGL(k_unpack_dense, dim3(blocks_for((long)B * 425 * 169, 256)), dim3(256), 0, st,
                               (const short *)cur->d, region, B, 425, 13, 13, g.Wp, g.PL, g.cg_stride);
            break;
        }
        }
        if (ev) (void)hipEventRecord(ev[i + 1], st);
    }
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    if (ev) c->prof_runs++;
    if (final_q) *final_q = c->final_q;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_run_batch_int16_host(yolo2_hip_ctx *c, const float *frames, int batch, int16_t *region,
                                              int *final_q)
{
    if (!c || !frames || !region) return fail(YOLO2_ERROR, "null argument");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    float *fd = nullptr;
    short *rd = nullptr;
    HIP_TRY(hipMalloc((void **)&fd, (size_t)batch * YOLO2_FRAME_ELEMS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&rd, (size_t)batch * YOLO2_REGION_ELEMS * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemcpy(fd, frames, (size_t)batch * YOLO2_FRAME_ELEMS * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    int rc = yolo2_hip_run_batch_int16(c, (uint64_t)(uintptr_t)fd, batch, (uint64_t)(uintptr_t)rd, final_q, nullptr);
    if (rc == YOLO2_SUCCESS) {
        hipError_t e = hipMemcpy(region, rd, (size_t)batch * YOLO2_REGION_ELEMS * 2, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(YOLO2_DMA_ERROR, "D2H of region tensor failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(fd);
    (void)hipFree(rd);
    return rc;
}

extern "C" int yolo2_hip_debug_layer_output(yolo2_hip_ctx *c, int layer_idx, int frame, int16_t *out, size_t cap,
                                            size_t *out_elems)
{
    if (!c || !out) return fail(YOLO2_ERROR, "null argument");
    if (layer_idx < -1 || layer_idx > 30 || !c->batch || frame < 0 || frame >= c->batch) return fail(YOLO2_ERROR, "bad layer/frame");
    if (c->laned) {
        int li = (int)c->lanes.size() - 1;
        while (li > 0 && frame < c->lane_first[li]) --li;
        return yolo2_hip_debug_layer_output(c->lanes[li], layer_idx, frame - c->lane_first[li], out, cap, out_elems);
    }
    if (layer_idx >= 0 && c->fuse_pool[layer_idx] && layer_idx != 16)
        return fail(YOLO2_ERROR, "layer %d's tensor is not materialised: conv + pool run fused (YOLO2_NO_POOLFUSE=1 keeps it)", layer_idx);
    // layer -1 = the quantised network input (yolo2_model.cpp:257-273), 3 x 416 x 416
    const LayerDesc &l = kNet[layer_idx < 0 ? 0 : layer_idx];
    if (layer_idx >= 0 && l.type == L_ROUTE) return fail(YOLO2_ERROR, "route layers have no tensor of their own");
    const Tensor &t = layer_idx < 0 ? c->t_in : c->t_out[layer_idx];
    int C = layer_idx < 0 ? 3 : (l.type == L_MAX ? l.c : l.n), H = t.g.H, W = t.g.W;
    const short *base = (const short *)t.d;
    if (layer_idx == 24) base += (long)64 * t.g.cg_stride * 4;  // channels 256.. of the concat tensor
    const int W8 = (W + 7) & ~7;
    const size_t n = (size_t)C * H * W8;
    if (out_elems) *out_elems = n;
    if (cap < n) return fail(YOLO2_ERROR, "output buffer too small (%zu < %zu)", cap, n);
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    short *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, n * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(tmp, 0, n * 2), YOLO2_DMA_ERROR);
    hipLaunchKernelGGL(k_items_to_ref, dim3(blocks_for((long)C * H * W, 256)), dim3(256), 0, nullptr, base, tmp, C, H, W, W8,
                       t.g.Wp, t.g.PL, t.g.cg_stride, frame);
    hipError_t e = hipMemcpy(out, tmp, n * 2, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(YOLO2_DMA_ERROR, "D2H failed: %s", hipGetErrorString(e));
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
    size_t cap = 0;   // bytes of the largest chunk
    for (int k = 0; k < chunks; ++k) {
        size_t sum = 0;
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
        size_t off = 0;
        for (int i = 0; i < nf; ++i) {
            const size_t bytes = (size_t)widths[first + i] * heights[first + i] * channels;
            memcpy(hin[b] + off, images[first + i], bytes);
            offs[(size_t)i] = off;
            off += padded(bytes);
        }
        Y2_TRY(hipMemcpyAsync(dbytes[b], hin[b], off, hipMemcpyHostToDevice, s_in), YOLO2_DMA_ERROR);
        Y2_TRY(hipEventRecord(e_in[b], s_in), YOLO2_ERROR);
        Y2_TRY(hipStreamWaitEvent(s_run, e_in[b], 0), YOLO2_ERROR);
        for (int f = 0; f < batch && rc == YOLO2_SUCCESS; ++f) {   // a partial last chunk repeats its last image
            const int i = std::min(f, nf - 1);
            rc = yolo2_hip_letterbox_u8((uint64_t)(uintptr_t)(dbytes[b] + offs[(size_t)i]), widths[first + i], heights[first + i],
                                        channels, (uint64_t)(uintptr_t)(din[b] + (size_t)f * YOLO2_FRAME_ELEMS), 416, 416, s_run);
        }
        if (rc == YOLO2_SUCCESS) rc = yolo2_hip_run_batch_int16(c, (uint64_t)(uintptr_t)din[b], batch, (uint64_t)(uintptr_t)dout[b], &q, s_run);
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

// ---------------------------------------------------------------------------- fp16 MFMA path

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static int load_fp32_common(yolo2_hip_ctx *c, const void *weights_reorg, size_t n_weights, const void *bias, size_t n_bias, hipMemcpyKind kind);

extern "C" int yolo2_hip_load_weights_fp32(yolo2_hip_ctx *c, const float *weights_reorg, size_t n_weights,
                                           const float *bias, size_t n_bias)
{
    return load_fp32_common(c, weights_reorg, n_weights, bias, n_bias, hipMemcpyHostToDevice);
}

extern "C" int yolo2_hip_load_weights_fp32_dev(yolo2_hip_ctx *c, uint64_t weights_reorg_dev, size_t n_weights, uint64_t bias_dev,
                                               size_t n_bias)
{
    return load_fp32_common(c, (const void *)(uintptr_t)weights_reorg_dev, n_weights, (const void *)(uintptr_t)bias_dev, n_bias,
                            hipMemcpyDeviceToDevice);
}

static int load_fp32_common(yolo2_hip_ctx *c, const void *weights_reorg, size_t n_weights, const void *bias, size_t n_bias, hipMemcpyKind kind)
{
    if (!c || !weights_reorg || !bias) return fail(YOLO2_ERROR, "null argument");
    if (n_weights < YOLO2_N_WEIGHTS) return fail(YOLO2_ERROR, "weights file too small");
    if (n_bias < YOLO2_N_BIAS) return fail(YOLO2_ERROR, "bias file too small");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    long wtot = 0, btot = 0;
    int ord = 0;
    for (int i = 0; i < 32; ++i)
        if (kNet[i].type == L_CONV) {
            const LayerDesc &l = kNet[i];
            const int npad = round_up(l.n, l.n <= 64 ? 64 : kBN);
            c->wh_off[ord] = wtot;
            c->biasf_off[ord] = btot;
            wtot += i == 0 ? (long)npad * 32 : (long)npad * l.size * l.size * round_up(l.c, 32);
            btot += npad;
            ord++;
        }
    for (yolo2_hip_ctx *l : c->f16_lanes) yolo2_hip_destroy(l);   // they alias the buffers that are about to be replaced
    c->f16_lanes.clear();
    if (c->wh) (void)hipFree(c->wh);
    if (c->biasf) (void)hipFree(c->biasf);
    c->wh = nullptr;
    c->biasf = nullptr;
    float *wd = nullptr, *bd = nullptr;
    HIP_TRY(hipMalloc((void **)&c->wh, (size_t)wtot * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&c->biasf, (size_t)btot * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&wd, (size_t)YOLO2_N_WEIGHTS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&bd, (size_t)YOLO2_N_BIAS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemcpy(wd, weights_reorg, (size_t)YOLO2_N_WEIGHTS * 4, kind), YOLO2_DMA_ERROR);
    HIP_TRY(hipMemcpy(bd, bias, (size_t)YOLO2_N_BIAS * 4, kind), YOLO2_DMA_ERROR);
    long woff = 0, boff = 0;
    ord = 0;
    for (int i = 0; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        if (l.type != L_CONV) continue;
        const int npad = round_up(l.n, l.n <= 64 ? 64 : kBN), KK = l.size * l.size;
        const int Cp = i == 0 ? 32 : round_up(l.c, 32);
        const long n = (long)npad * (i == 0 ? 1 : KK) * Cp;
        hipLaunchKernelGGL(k_pack_weights_f16, dim3(blocks_for(std::max<long>(n, npad), 256)), dim3(256), 0, nullptr, wd + woff,
                           c->wh + c->wh_off[ord], c->biasf + c->biasf_off[ord], bd + boff, l.c, l.n, KK, Cp, npad, i == 0 ? 1 : 0);
        woff += yolo2_weight_len[ord];
        boff += yolo2_bias_len[ord];
        ord++;
    }
    // the halo-tile kernels use up to the whole 160 KiB of LDS: raise their dynamic-LDS limit on THIS device
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo<256, 2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo<256, 2, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo_p<256, 16, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo_p<256, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo_p<128, 8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_conv_f16_halo_p<128, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_gemm1_f16_p<256, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    HIP_TRY(hipFuncSetAttribute((const void *)k_gemm1_f16_p<256, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), YOLO2_ERROR);
    if (!c->w0f) HIP_TRY(hipMalloc((void **)&c->w0f, (27 * 32 + 32) * sizeof(float)), YOLO2_MMAP_ERROR);
    hipLaunchKernelGGL(k_pack_w0_f32, dim3(4), dim3(256), 0, nullptr, wd, bd, c->w0f, c->w0f + 27 * 32);
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
    // the fp32 blobs stay resident (204 MB of 288 GB): yolo2_hip_run_frame_fp32_host consumes them as they are
    if (c->wpkf) (void)hipFree(c->wpkf);         // the tiled fp32 path re-packs from the new blobs at its next run
    if (c->biasf32_pk) (void)hipFree(c->biasf32_pk);
    c->wpkf = c->biasf32_pk = nullptr;
    if (c->wf32) (void)hipFree(c->wf32);
    if (c->bf32) (void)hipFree(c->bf32);
    c->wf32 = wd;
    c->bf32 = bd;
    c->f16_loaded = true;
    return YOLO2_SUCCESS;
}

// fp32 whole network, reference arithmetic: every layer in the reference's [C][H][W8] layout through the
// one-thread-per-output kernels (k_conv_ref_f32: reference operation order, no FMA contraction; k_pool_ref;
// the legacy reorg indexing of yolo2_model.cpp:112-129,358-376), i.e. what yolov2_hls_ps does at
// Precision::FP32 (yolo2_model.cpp:229-449).  Bit-identical to the reference's fp32 region tensor; a
// correctness path (about 0.2 s per frame), not a fast one - the fast floating-point path is run_batch_fp16.
__global__ void k_reorg_ref_f32(const float *__restrict__ in, float *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // over 256*13 rows x 13 columns
    if (t >= 256 * 13 * 13) return;
    const int kr = t / 13, cc = t - kr * 13;
    const int p = kr * 13 + cc;                            // index into the permuted dense tensor
    const int i = p % 26, rest = p / 26, j = rest % 416, k = rest / 416;
    const int d = (2 * i + k % 2) + 52 * (2 * j + k / 2);  // index into the dense 64 x 26 x 26 input
    out[(size_t)kr * 16 + cc] = in[(size_t)(d / 26) * 32 + d % 26];
}

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
    free_f32_activations(c);
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
    const char *fp = getenv("YOLO2_F32_P");   // test hook: 1 / 2 / 4 for every layer
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
            if (fp && atoi(fp) != P) continue;
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

static int alloc_half(yolo2_hip_ctx::HalfTensor &t, int C, int Cp, int H, int W, int B)
{
    t.C = C; t.Cp = Cp; t.H = H; t.W = W; t.Wp = W + 1; t.PL = (H + 1) * t.Wp; t.B = B;
    t.items = (size_t)kLead + (size_t)B * t.PL + kTail;
    HIP_TRY(hipMalloc((void **)&t.d, t.items * Cp * 2), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemset(t.d, 0, t.items * Cp * 2), YOLO2_DMA_ERROR);  // zeros = conv padding and channel padding
    return YOLO2_SUCCESS;
}

static int ensure_f16_batch(yolo2_hip_ctx *c, int B)
{
    if (c->f16_batch == B) return YOLO2_SUCCESS;
    free_f16_activations(c);
    int rc;
    if ((rc = alloc_half(c->h_cat, 1280, 1280, 13, 13, B))) return rc;
    for (int i = 1; i < 30; ++i) {   // layer 0's 416x416x32 tensor never exists: conv0+pool are fused
        const LayerDesc &l = kNet[i];
        if (l.type == L_CONV && i != 24) {
            if ((rc = alloc_half(c->h_out[i], l.n, round_up(l.n, 32), l.h, l.w, B))) return rc;
        } else if (l.type == L_MAX) {
            if ((rc = alloc_half(c->h_out[i], l.c, round_up(l.c, 32), l.h / 2, l.w / 2, B))) return rc;
        }
    }
    c->h_out[24] = c->h_cat;
    c->h_out[27] = c->h_cat;
    c->f16_batch = B;
    // the zero fills above run on the null stream; the pass may be enqueued on a non-blocking stream (the lanes'
    // are), which does not order itself behind it
    HIP_TRY(hipDeviceSynchronize(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_run_batch_fp16(yolo2_hip_ctx *c, uint64_t frames_dev, int batch, uint64_t region_dev, void *stream)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (!c->f16_loaded) return fail(YOLO2_ERROR, "fp32 weights not loaded (yolo2_hip_load_weights_fp32)");
    if (!frames_dev || !region_dev) return fail(YOLO2_ERROR, "null buffer address");
    if (batch <= 0 || batch > 4096) return fail(YOLO2_ERROR, "batch %d out of range", batch);
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    hipStream_t st = (hipStream_t)stream;
    // Two half-batch lanes like the int16 path: the big-tile kernels run one workgroup per CU and a layer is only
    // 2-3 generations of workgroups, so a second stream's launches fill the last, partly empty generation.
    const int want_lanes = getenv("YOLO2_F16_LANES") ? std::max(1, std::min(8, atoi(getenv("YOLO2_F16_LANES")))) : 2;
    if (!c->is_lane && batch >= 64 && want_lanes > 1 && batch % want_lanes == 0 && !getenv("YOLO2_F16_NO_LANES")) {
        if ((int)c->f16_lanes.size() != want_lanes) {
            for (yolo2_hip_ctx *l : c->f16_lanes) yolo2_hip_destroy(l);
            c->f16_lanes.clear();
            if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), YOLO2_ERROR);
            std::vector<yolo2_hip_ctx *> made;   // committed only when both lanes are complete
            bool ok = true;
            for (int i = 0; i < want_lanes && ok; ++i) {
                yolo2_hip_ctx *l = new (std::nothrow) yolo2_hip_ctx();
                if (!l) { ok = false; break; }
                made.push_back(l);
                l->device = c->device;
                l->is_lane = true;
                l->wh = c->wh; l->biasf = c->biasf; l->w0f = c->w0f;
                memcpy(l->wh_off, c->wh_off, sizeof(c->wh_off));
                memcpy(l->biasf_off, c->biasf_off, sizeof(c->biasf_off));
                l->f16_loaded = true;
                ok = hipStreamCreateWithFlags(&l->lane_stream, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming) == hipSuccess;
            }
            if (!ok) {
                for (yolo2_hip_ctx *l : made) yolo2_hip_destroy(l);
                return fail(YOLO2_ERROR, "fp16 lanes: context / stream / event creation failed");
            }
            c->f16_lanes = made;
            if (c->prof) (void)yolo2_hip_set_profiling(c->f16_lanes[0], 1);
        }
        const int half = batch / want_lanes;
        HIP_TRY(hipEventRecord(c->ev_fork, st), YOLO2_ERROR);
        for (int i = 0; i < want_lanes; ++i) {
            yolo2_hip_ctx *l = c->f16_lanes[i];
            HIP_TRY(hipStreamWaitEvent(l->lane_stream, c->ev_fork, 0), YOLO2_ERROR);
            const int rc = yolo2_hip_run_batch_fp16(l, frames_dev + (uint64_t)i * half * YOLO2_FRAME_ELEMS * sizeof(float), half,
                                                    region_dev + (uint64_t)i * half * YOLO2_REGION_ELEMS * sizeof(float), l->lane_stream);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(l->ev_join, l->lane_stream), YOLO2_ERROR);
            HIP_TRY(hipStreamWaitEvent(st, l->ev_join, 0), YOLO2_ERROR);
        }
        return YOLO2_SUCCESS;
    }
    int rc = ensure_f16_batch(c, batch);
    if (rc) return rc;
    const int B = batch;
    if (c->prof) {
        const int prc = ensure_prof_events(c);
        if (prc) return prc;
    }
    hipEvent_t *ev = c->prof ? c->ev[c->prof_runs % yolo2_hip_ctx::kProfSlots] : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[0], st), YOLO2_ERROR);
    {   // layers 0+1 fused: conv 3->32 + leaky + 2x2 pool straight from the float frames
        const auto &g = c->h_out[1];
        if (!getenv("YOLO2_F16_NO_MFMA0"))   // 416 = 26 x 16 = 13 x 32: the tile grid is exact
            hipLaunchKernelGGL(k_conv0_pool_mfma, dim3((unsigned)B * (416 / 16) * (416 / 32)), dim3(256), 0, st,
                               (const float *)(uintptr_t)frames_dev, (const float *)c->w0f, (const float *)(c->w0f + 27 * 32), g.d,
                               416, 416, g.Wp, g.PL);
        else
            hipLaunchKernelGGL(k_conv0_pool_f16, dim3(blocks_for((long)B * g.H * g.W, 256), 2), dim3(256), 0, st,
                               (const float *)(uintptr_t)frames_dev, (const float *)c->w0f, (const float *)(c->w0f + 27 * 32), g.d, B,
                               416, 416, g.Wp, g.PL);
        if (ev) { (void)hipEventRecord(ev[1], st); (void)hipEventRecord(ev[2], st); }
    }
    int ord = 1, skip_pool = -1;
    const yolo2_hip_ctx::HalfTensor *cur = &c->h_out[1];
    for (int i = 2; i < 32; ++i) {
        const LayerDesc &l = kNet[i];
        switch (l.type) {
        case L_CONV: {
            const auto *tin = i == 26 ? &c->h_out[16] : (i == 29 ? &c->h_cat : cur);
            const auto &tout = c->h_out[i];
            ConvF16Args a;
            a.B = B; a.H = l.h; a.W = l.w; a.Wp = l.w + 1; a.PL = (l.h + 1) * (l.w + 1);
            a.Cp_in = tin->Cp;
            a.Cp_out = i == 30 ? 0 : tout.Cp;
            a.N = l.n;
            a.out_ch_off = i == 24 ? 256 : 0;
            a.n_store = i == 30 ? l.n : round_up(l.n, 32);
            a.npix = B * l.h * l.w;
            a.leaky = l.leaky;
            a.KS = l.size;
            a.pool = 0; a.oWp = a.oPL = a.npool = 0;
            set_fast_div(a);
#ifdef Y2_STAMPS
            a.stamp = getenv("YOLO2_STAMP_LAYER") && atoi(getenv("YOLO2_STAMP_LAYER")) == i;
#else
            a.stamp = 0;
#endif
            const _Float16 *wp = (const _Float16 *)(c->wh + c->wh_off[ord]);
            const float *bp = (const float *)(c->biasf + c->biasf_off[ord]);
            _Float16 *op = i == 30 ? (_Float16 *)nullptr : tout.d;
            float *of = i == 30 ? (float *)(uintptr_t)region_dev : (float *)nullptr;
            const bool bk64 = a.Cp_in % 64 == 0;   // K-step of 64 channels wherever the item size allows it
            const _Float16 *ip = (const _Float16 *)tin->d;
            const bool glds = bk64 && !getenv("YOLO2_F16_NO_GLDS");   // LDS-DMA staging wherever the K-step is 64
            // Conv layers whose only consumer is the 2x2 pool after them (2 and 6; 10 runs the halo kernel, 16 also
            // feeds the route) store the pooled tensor directly: MFMA rows ordered by pool window, max in the epilogue.
            // (layer 6 only where the persistent halo kernel, which stores the full-resolution tensor, does not take it)
            // Measured (tools/stamps.py, profiles/r02_f16_halo_wg_timeline.txt): at <= 52x52 the persistent kernel's tile costs what the
            // one-tile-per-workgroup kernel's does (its epilogue is 6k cycles of VALU work that nothing overlaps either way, and the
            // next tile's staging slows the taps it runs beside), and its few long workgroups pack worse next to the other lane's
            // (-4 % at batch 256).  It is used where the halo kernel does not fit: the 104x104 layers (-15..18 % vs k_conv_f16_glds).
            // Layer 6 stays on the pool-fused per-tap kernel: the persistent kernel stores the full-resolution tensor, and the pool
            // kernel that then has to follow (0.08 ms at batch 128) costs more than the conv gains (0.31 -> 0.26 ms).
            const bool pool_next = kNet[i + 1].type == L_MAX && !getenv("YOLO2_F16_NO_POOLFUSE");
            const bool persist_ok = !getenv("YOLO2_F16_NO_GLDS") && !getenv("YOLO2_F16_NO_HALO") && !getenv("YOLO2_F16_NO_PERSIST") &&
                                    ((l.w > 52 && !(i == 6 && pool_next)) || getenv("YOLO2_F16_PERSIST_ALL"));
            const bool fuse_pool = (i == 2 || (i == 6 && !persist_ok)) && pool_next;
            if (fuse_pool) {
                const auto &tp = c->h_out[i + 1];
                a.pool = 1; a.oWp = tp.Wp; a.oPL = tp.PL; a.npool = B * tp.H * tp.W;
                a.Cp_out = tp.Cp;
                op = tp.d;
                skip_pool = i + 1;
            }
            // 1x1 layers: persistent workgroups over a ring of staged K-steps (k_gemm1_f16_p)
            if (l.size == 1 && bk64 && (i == 30 || getenv("YOLO2_F16_RING_ALL")) && ((size_t)kLead + (size_t)B * a.PL) * a.Cp_in * 2 < (1ull << 32) && !getenv("YOLO2_F16_NO_RING")) {
                const int bn = l.n <= 64 ? 64 : 128;
                a.n_tiles = round_up(l.n, bn) / bn;
                const int T = ((a.npix + 255) / 256) * a.n_tiles;
                const int rounds = (T + 255) / 256;                                    // tiles per workgroup
                const int G = std::min(256, std::max(8, round_up((T + rounds - 1) / rounds, 8)));
                if (bn == 64)
                    hipLaunchKernelGGL((k_gemm1_f16_p<256, 64, 3>), dim3(G), dim3(256), 3 * (256 + 64) * 128, st, ip, wp, bp, op, of, a, T);
                else
                    hipLaunchKernelGGL((k_gemm1_f16_p<256, 128, 3>), dim3(G), dim3(512), 3 * (256 + 128) * 128, st, ip, wp, bp, op, of, a, T);
                if (i != 30) cur = &c->h_out[i];
                ord++;
                break;
            }
            // the 32-channel layer + its pool: 16 x 16 tiles, patch and all nine taps' weights resident in LDS (k_conv_f16_c32_pool)
            if (fuse_pool && a.Cp_in == 32 && l.size == 3 && l.n == 64 && l.h % 16 == 0 && l.w % 16 == 0 && a.Cp_out >= 64 &&
                ((size_t)kLead + (size_t)B * a.PL) * 64 < (1ull << 32) && !getenv("YOLO2_F16_NO_C32")) {
                set_fast_div(a);
                const int T = B * (l.h / 16) * (l.w / 16);
                hipLaunchKernelGGL(k_conv_f16_c32_pool, dim3((unsigned)std::min(T, 512)), dim3(256), (9 * 64 + 336) * 64, st, ip, wp, bp, op, a, T);   // two workgroups per CU, weights staged once each
                cur = &c->h_out[i + 1];
                ord++;
                break;
            }
            const int m_tiles = fuse_pool ? (a.npool + 31) / 32 : (a.npix + 127) / 128;   // 128-row tiles (32 pool windows)
            if (l.n <= 64) {
                a.n_tiles = round_up(l.n, 64) / 64;
                const dim3 grid(m_tiles * a.n_tiles);
                if (glds) hipLaunchKernelGGL((k_conv_f16_glds<64>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
                else if (bk64) hipLaunchKernelGGL((k_conv_f16<128, 64, 64>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
                else hipLaunchKernelGGL((k_conv_f16<128, 64, 32>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
            } else {
                a.n_tiles = round_up(l.n, kBN) / kBN;
                // 3x3 layers: halo-tile kernel (input tile staged once per 64-channel chunk, nine taps read it
                // shifted) wherever its LDS arena fits: 2 x lt_rows x 128 B (A) + 3 x 128 x 128 B (B) + fo table
                // persistent halo-tile kernel: the workgroup walks its tiles, the next tile's staging overlaps this one's tail
                if (glds && l.size == 3 && l.n % kBN == 0 && persist_ok && !fuse_pool) {
                    const int lt_rows = round_up(256 + 2 * (l.w + 1), 8) + 8;
                    const size_t a_bytes = (size_t)2 * lt_rows * 128, cap = 160 * 1024;
                    const bool off32 = ((size_t)kLead + (size_t)B * a.PL) * std::max(a.Cp_in, a.Cp_out) * 2 < (1ull << 32);
                    const bool wide = l.n % 256 == 0 && a_bytes + (size_t)2 * 256 * 128 <= cap;
                    const int bn = wide ? 256 : 128;
                    const size_t lds = a_bytes + (size_t)2 * bn * 128;
                    if (off32 && lds <= cap && lt_rows - 8 <= 8 * 8 * 8) {
                        a.n_tiles = l.n / bn;
                        const int T = ((a.npix + 255) / 256) * a.n_tiles;
                        const int rounds = (T + 255) / 256;
                        // (Launched with one tile per workgroup - the same kernel, only the LDS-free epilogue and the operand order
                        //  differ from k_conv_f16_halo - it measured 2.8 % slower over the pass at batch 256.)
                        const int G = std::min(256, std::max(8, round_up((T + rounds - 1) / rounds, 8)));
                        const bool m16 = getenv("YOLO2_F16_M16") != nullptr;
                        if (wide && m16) hipLaunchKernelGGL((k_conv_f16_halo_p<256, 16, 16>), dim3(G), dim3(1024), lds, st, ip, wp, bp, op, a, lt_rows, T);
                        else if (wide) hipLaunchKernelGGL((k_conv_f16_halo_p<256, 16, 32>), dim3(G), dim3(1024), lds, st, ip, wp, bp, op, a, lt_rows, T);
                        else if (m16) hipLaunchKernelGGL((k_conv_f16_halo_p<128, 8, 16>), dim3(G), dim3(512), lds, st, ip, wp, bp, op, a, lt_rows, T);
                        else hipLaunchKernelGGL((k_conv_f16_halo_p<128, 8, 32>), dim3(G), dim3(512), lds, st, ip, wp, bp, op, a, lt_rows, T);
                        cur = &c->h_out[i];
                        ord++;
                        break;
                    }
                }
                if (glds && l.size == 3 && l.n % kBN == 0 && !getenv("YOLO2_F16_NO_HALO")) {
                    // dense tile: 256 pixels + W+1 on either side, rounded to 8-row groups, + 8 zero rows
                    const int lt_rows = round_up(256 + 2 * (l.w + 1), 8) + 8;
                    const size_t a_bytes = (size_t)2 * lt_rows * 128, fo_bytes = 256 * sizeof(int), cap = 160 * 1024;
                    const size_t lds256 = a_bytes + (size_t)2 * 256 * 128 + fo_bytes, lds128 = a_bytes + (size_t)3 * 128 * 128 + fo_bytes;
                    const bool wide = l.n % 256 == 0 && lds256 <= cap && a_bytes + (size_t)2 * 256 * 128 >= (size_t)256 * 264 * 2 &&
                                      !getenv("YOLO2_F16_NO_WIDE");
                    const bool three = lds128 <= cap;
                    // (the two-buffer 256x128 form that would fit the 104x104 layers runs one workgroup per CU and measured
                    //  6 % slower there than the 128x128 kernel with two: only the shapes below are used)
                    // (the kernel addresses its tensors with 32-bit byte offsets from a uniform base)
                    const bool off32 = ((size_t)kLead + (size_t)B * a.PL) * a.Cp_in * 2 < (1ull << 32);
                    const bool fits = lt_rows - 8 <= 8 * 8 * 8 && a_bytes >= (size_t)256 * kCtRow * 2 && (wide || three) && off32;
                    if (fits) {   // (the kernels' dynamic-LDS limit was raised for this device in load_weights_fp32)
                        if (wide) {
                            a.n_tiles = l.n / 256;
                            const dim3 hgrid(((a.npix + 255) / 256) * a.n_tiles);
                            if (getenv("YOLO2_F16_M16"))
                                hipLaunchKernelGGL((k_conv_f16_halo<256, 2, 16, 16>), hgrid, dim3(1024), lds256, st, ip, wp, bp, op, a, lt_rows);
                            else if (!getenv("YOLO2_F16_W8"))   // 16 wavefronts of 64x64 (4 per SIMD, +4 %) instead of 8 of 128x64
                                hipLaunchKernelGGL((k_conv_f16_halo<256, 2, 16>), hgrid, dim3(1024), lds256, st, ip, wp, bp, op, a, lt_rows);
                            else
                                hipLaunchKernelGGL((k_conv_f16_halo<256, 2>), hgrid, dim3(512), lds256, st, ip, wp, bp, op, a, lt_rows);
                        } else {
                            const dim3 hgrid(((a.npix + 255) / 256) * a.n_tiles);
                            hipLaunchKernelGGL((k_conv_f16_halo<128, 3>), hgrid, dim3(512), lds128, st, ip, wp, bp, op, a, lt_rows);   // `fits` without `wide` implies `three`
                        }
                        if (i != 30) cur = &c->h_out[i];
                        ord++;
                        break;
                    }
                }
                // (a 256x128 tile with 8 wavefronts and per-tap A staging was measured 8 % SLOWER than 128x128
                //  with two workgroups per CU: without the halo reuse the bigger tile only adds barrier cost)
                const dim3 grid(m_tiles * a.n_tiles);
                if (glds) hipLaunchKernelGGL((k_conv_f16_glds<128>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
                else if (bk64) hipLaunchKernelGGL((k_conv_f16<128, 128, 64>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
                else hipLaunchKernelGGL((k_conv_f16<128, 128, 32>), grid, dim3(256), 0, st, ip, wp, bp, op, of, a);
            }
            if (i != 30) cur = fuse_pool ? &c->h_out[i + 1] : &c->h_out[i];
            ord++;
            break;
        }
        case L_MAX: {
            if (i == skip_pool) break;   // already produced by the conv before it
            const auto &gi = *cur, &go = c->h_out[i];
            const long n = (long)B * go.H * go.W * (go.Cp / 8);
            hipLaunchKernelGGL(k_maxpool2_f16, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const _Float16 *)gi.d, go.d, go.Cp, B,
                               go.H, go.W, gi.Wp, gi.PL, go.Wp, go.PL);
            cur = &c->h_out[i];
            break;
        }
        case L_REORG: {
            const auto &gi = *cur, &go = c->h_cat;
            hipLaunchKernelGGL(k_reorg_f16, dim3(blocks_for((long)B * 256 * 169, 256)), dim3(256), 0, st, (const _Float16 *)gi.d,
                               go.d, B, gi.Cp, gi.Wp, gi.PL, go.Cp, go.Wp, go.PL);
            cur = &c->h_cat;
            break;
        }
        default:
            break;  // route: concat by placement; region: the last conv already wrote the dense fp32 tensor
        }
        if (ev) (void)hipEventRecord(ev[i + 1], st);
    }
    HIP_TRY(hipGetLastError(), YOLO2_ERROR);
    if (ev) c->prof_runs++;
    return YOLO2_SUCCESS;
}

#ifdef Y2_STAMPS
// diagnostic build only: the halo kernel's workgroup timeline of the launch selected by YOLO2_STAMP_LAYER
extern "C" int yolo2_hip_debug_stamps(unsigned long long *dst, int n_wg)
{
    if (n_wg > kStampWGs) n_wg = kStampWGs;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y2_stamps), (size_t)n_wg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int yolo2_hip_run_batch_fp16_host(yolo2_hip_ctx *c, const float *frames, int batch, float *region)
{
    if (!c || !frames || !region) return fail(YOLO2_ERROR, "null argument");
    HIP_TRY(hipSetDevice(c->device), YOLO2_INIT_ERROR);
    float *fd = nullptr, *rd = nullptr;
    HIP_TRY(hipMalloc((void **)&fd, (size_t)batch * YOLO2_FRAME_ELEMS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMalloc((void **)&rd, (size_t)batch * YOLO2_REGION_ELEMS * 4), YOLO2_MMAP_ERROR);
    HIP_TRY(hipMemcpy(fd, frames, (size_t)batch * YOLO2_FRAME_ELEMS * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    int rc = yolo2_hip_run_batch_fp16(c, (uint64_t)(uintptr_t)fd, batch, (uint64_t)(uintptr_t)rd, nullptr);
    if (rc == YOLO2_SUCCESS) {
        hipError_t e = hipMemcpy(region, rd, (size_t)batch * YOLO2_REGION_ELEMS * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(YOLO2_DMA_ERROR, "D2H of region tensor failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(fd);
    (void)hipFree(rd);
    return rc;
}

// ---------------------------------------------------------------------------- streaming host entry

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
