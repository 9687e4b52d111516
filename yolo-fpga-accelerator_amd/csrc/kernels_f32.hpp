// kernels_f32.hpp -- the fp32 form of the layer pipeline in the REFERENCE's arithmetic, tiled.
//
// compute() at Precision::FP32 (hls/core/core_compute.cpp:121-172) is, per output channel m and pixel,
//     acc = bias[m]
//     for each 4-input-channel group, for each tap (i, j):
//         ps  = (((0 + w0*x0) + w1*x1) + w2*x2) + w3*x3        (every * and + rounded to fp32: no FMA on x86-64)
//         acc = acc + ps
//     out = acc < 0 ? acc * 0.1f : acc                           (:201-205)
// Unlike the int16 chain this IS associative-free only in the sense that the ORDER is fixed: reproducing the order
// reproduces the bits.  The one-thread-per-output kernel (k_conv_ref_f32) does that at ~5 frames/s; this file keeps
// the order and borrows the int16 kernel's structure instead - items of 4 channels (here 4 floats = 16 bytes) in the
// flat shared-zero-row/column layout of layout.hpp, the input tile of a channel group staged once per workgroup in LDS
// (double-buffered, conflict-free ds_read_b128 by consecutive lanes), the group's weight slice staged beside it and read
// back as broadcasts, P pixels x 8 output channels of fp32 accumulators per lane.  Out-of-image taps and the 4th lane of the 3-channel input
// read stored zeros and are multiplied and added like the reference's zero-padded buffers are (w * 0 = +-0, ps + 0:
// the same signed-zero behaviour, checked bit for bit against the compiled reference's fixture).
// Issue cost: 4 v_mul_f32 + 5 v_add_f32 per (4 channels x tap x output) = 18 SIMD cycles per step and wavefront at
// 2 cycles each (all operands in VGPRs) -> 2.36 k frames/s is the VALU ceiling of the exact fp32 form (the fp16 MFMA
// path is the fast one).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_common.hpp"   // ConvArgs, flat_of, xcd_partition, k_repack_weights, k_pool_ref
#include "layout.hpp"

namespace y2 {

// one step for 8 output channels of one pixel: the reference's operation order, explicitly un-fused
__device__ __forceinline__ void step_f32(float (&acc)[8], const float4 x, const float4 *w)
{
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float ps = __fmul_rn(w[m].x, x.x);                 // 0 + w0*x0 == w0*x0 exactly (incl. the sign of a zero product: 0 + (-0) = +0 ... see below)
        ps = __fadd_rn(0.0f, ps);                          // the reference starts from +0.0f: +0 + (-0) = +0
        ps = __fadd_rn(ps, __fmul_rn(w[m].y, x.y));
        ps = __fadd_rn(ps, __fmul_rn(w[m].z, x.z));
        ps = __fadd_rn(ps, __fmul_rn(w[m].w, x.w));
        acc[m] = __fadd_rn(acc[m], ps);
    }
}

template <int KS, int P, int NST>
__global__ __launch_bounds__(256) void k_conv_f32(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                   const float4 *__restrict__ wpk, const float *__restrict__ bias, const ConvArgs a)
{
    // Weights go through LDS too (staged with the input tile, read back with BROADCAST ds_read_b128: every lane of a
    // wavefront reads the same 16 bytes), not through scalar loads as in the int16 kernel: on gfx950 any VALU instruction
    // with an SGPR operand issues in 4 cycles instead of 2 (profiles/r02_ubench_valu_sgpr_operand.txt: v_mul_f32 2.09 vs
    // 4.03), which would make the four multiplies of a step cost as much as its five adds together.
    extern __shared__ float4 ldsf[];
    constexpr int T = 64 * P, KT = KS * KS, WIT = KT * 32, NW = (WIT + 255) / 256;   // weight items (float4) per channel group
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x, mb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, mb);
    const int q0 = tile * T, qlast = min(q0 + T, a.npix) - 1;
    const int halo = (KS == 3) ? a.Wp + 1 : 0;
    const int fmin = flat_of(a, q0), fmax = flat_of(a, qlast);
    const int tile_start = fmin - halo;
    const int Lt = min(fmax - fmin + 1 + 2 * halo, a.lt_max);
    const int buf_items = a.lt_max + WIT;          // one LDS buffer: input tile, then the group's 32-channel weight slice

    int fo[P], rowaddr[P][KS];
    bool valid[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = q0 + p * 64 + lane;
        valid[p] = q <= qlast;
        fo[p] = flat_of(a, min(q, qlast));
        const int lo = fo[p] - tile_start;
#pragma unroll
        for (int i = 0; i < KS; ++i) rowaddr[p][i] = (KS == 3) ? (lo + (i - 1) * a.Wp - 1) * 16 : lo * 16;
    }
    float acc[P][8];
    {
        const float *bp = bias + mb * 32 + wave * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float b = bp[m];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p][m] = b;
        }
    }
    const char *lds_b = reinterpret_cast<const char *>(ldsf);
    const float4 *src = in + kLead + tile_start;
    const float4 *wsrc = wpk + (long)mb * a.CGin * WIT;    // this block's 32 channels, group 0: [tap][32] items
    int wrd = (a.lt_max + wave * 8) * 16;                   // byte offset of this wavefront's 8 channels, tap 0, inside a buffer

    float4 stage[NST], wst[NW];
    auto issue = [&]() {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 256;
            if (i < Lt) stage[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = tid + k * 256;
            if (i < WIT) wst[k] = wsrc[i];
        }
    };
    auto commit = [&](int b) {
        float4 *dst = ldsf + b * buf_items;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 256;
            if (i < Lt) dst[i] = stage[k];
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = tid + k * 256;
            if (i < WIT) dst[a.lt_max + i] = wst[k];
        }
    };
    issue();
    commit(0);
    __syncthreads();
    for (int cg = 0; cg < a.CGin; ++cg) {
        const bool more = cg + 1 < a.CGin;
        if (more) {
            src += a.in_cg_stride;
            wsrc += WIT;
            issue();
        }
        const char *tl = lds_b + (cg & 1) * buf_items * 16;
#pragma unroll
        for (int tap = 0; tap < KT; ++tap) {
            float4 w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) w[m] = *reinterpret_cast<const float4 *>(tl + wrd + (tap * 32 + m) * 16);   // broadcast reads
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float4 x = *reinterpret_cast<const float4 *>(tl + rowaddr[p][tap / KS] + (tap % KS) * 16);
                step_f32(acc[p], x, w);
            }
            // Without a tie between taps hipcc hoists the weight reads of all nine taps (288 VGPRs) to the top of the
            // group.  Empty volatile asm statements keep their order: one per accumulator (all of this tap's sums exist
            // here; with fewer, hipcc runs one channel through all taps first) and one on the weight offset the next
            // tap reads from.  No instruction is emitted.
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int m = 0; m < 8; ++m) asm volatile("" : "+v"(acc[p][m]));
            asm volatile("" : "+s"(wrd));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit((cg + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int cgo = mb * 8 + wave * 2 + g;
        if (cgo >= a.CGout) continue;
        float4 *dst = out + a.out_base + (long)cgo * a.out_cg_stride;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float e = acc[p][g * 4 + t];
                v[t] = (a.leaky && e < 0.0f) ? __fmul_rn(e, 0.1f) : e;
            }
            if (valid[p]) dst[fo[p]] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// float [B][3][416][416] -> items (4th lane 0)
__global__ void k_pack_input_f32(const float *__restrict__ frames, float4 *__restrict__ out, int B, int H, int W, int Wp, int PL)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (q >= B * HW) return;
    const int b = q / HW, r = q - b * HW, y = r / W, x = r - y * W;
    const float *f = frames + (long)b * 3 * HW + r;
    out[kLead + (long)b * PL + (long)(y + 1) * Wp + x] = make_float4(f[0], f[HW], f[2 * HW], 0.f);
}

// pool_yolo2 at fp32 (core_compute.cpp:266-305): max over the window starting from the pad value -1024*1024
__global__ void k_maxpool2_f32(const float4 *__restrict__ in, float4 *__restrict__ out, int CG, int B, int OH, int OW, int iWp,
                               int iPL, int oWp, int oPL)
{
    const long n = (long)CG * B * OH * OW;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int x = (int)(t % OW), y = (int)((t / OW) % OH);
    const long pb = t / ((long)OW * OH);
    const float4 *s = in + kLead + pb * iPL + (long)(2 * y + 1) * iWp + 2 * x;
    const float4 v[4] = {s[0], s[1], s[iWp], s[iWp + 1]};
    float o[4] = {-1024.f * 1024.f, -1024.f * 1024.f, -1024.f * 1024.f, -1024.f * 1024.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // `if (v > best) best = v` in the reference's window order
        if (v[k].x > o[0]) o[0] = v[k].x;
        if (v[k].y > o[1]) o[1] = v[k].y;
        if (v[k].z > o[2]) o[2] = v[k].z;
        if (v[k].w > o[3]) o[3] = v[k].w;
    }
    out[kLead + pb * oPL + (long)(y + 1) * oWp + x] = make_float4(o[0], o[1], o[2], o[3]);
}

// legacy reorg of the 64 x 26 x 26 tensor into channel groups [0, 64) of the 1280-channel concat tensor (no Q shift at fp32)
__global__ void k_reorg_f32(const float *__restrict__ in, float *__restrict__ out, int B, int iWp, int iPL, long i_cg_stride, int oWp,
                            int oPL, long o_cg_stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 256 * 169) return;
    const int b = t / (256 * 169), o = t - b * (256 * 169);
    const int k = o / (26 * 416), rem = o - k * (26 * 416), j = rem / 26, i = rem - j * 26;
    const int sidx = (2 * i + (k & 1)) + 52 * (2 * j + (k >> 1));
    const int sc = sidx / 676, sr = sidx - sc * 676, sy = sr / 26, sx = sr - sy * 26;
    const float v = in[(kLead + (long)(sc >> 2) * i_cg_stride + (long)b * iPL + (long)(sy + 1) * iWp + sx) * 4 + (sc & 3)];
    const int oc = o / 169, orr = o - oc * 169, oy = orr / 13, ox = orr - oy * 13;
    out[(kLead + (long)(oc >> 2) * o_cg_stride + (long)b * oPL + (long)(oy + 1) * oWp + ox) * 4 + (oc & 3)] = v;
}

// items -> dense [B][C][H][W] floats (the 13-of-16 region gather, yolo2_model.cpp:406-414)
__global__ void k_unpack_dense_f32(const float *__restrict__ in, float *__restrict__ out, int B, int C, int H, int W, int Wp, int PL,
                                   long cg_stride)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)B * C * H * W;
    if (t >= n) return;
    const int x = (int)(t % W), y = (int)((t / W) % H), c = (int)((t / ((long)W * H)) % C), b = (int)(t / ((long)W * H * C));
    out[t] = in[(kLead + (long)(c >> 2) * cg_stride + (long)b * PL + (long)(y + 1) * Wp + x) * 4 + (c & 3)];
}


// ------------------------------------------------------------------ one-thread-per-output reference-layout kernels
// (the independent second implementation behind yolo2_execute_conv_layer_f32 and yolo2_hip_run_frame_fp32_host)
// fp32 twin, reference operation order with no FMA contraction (core_compute.cpp:121-172, :201-205)
__global__ void k_conv_ref_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ w,
                               const float *__restrict__ bias, int C, int N, int K, int stride, int W, int H,
                               int OW, int OH, int pad, int leaky)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * OH * OW) return;
    const int x = t % OW, y = (t / OW) % OH, m = t / (OW * OH);
    const int W8 = (W + 7) & ~7, OW8 = (OW + 7) & ~7, KK = K * K;
    const int m0 = m / kTm * kTm, tm = m - m0, tm_min = min(kTm, N - m0);
    float acc = bias[m];
    for (int n0 = 0; n0 < C; n0 += kTn) {
        const int tn_min = min(kTn, C - n0);
        const float *wb = w + (long)m0 * C * KK + (long)tm_min * n0 * KK;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                const int sy = y * stride + i - pad, sx = x * stride + j - pad;
                const bool inb = sy >= 0 && sy < H && sx >= 0 && sx < W;
                float ps = 0.f;
                for (int tt = 0; tt < kTn; ++tt) {
                    const float wv = tt < tn_min ? wb[(long)(i * K + j) * tm_min * tn_min + tm * tn_min + tt] : 0.f;
                    const float xv = (inb && tt < tn_min) ? in[((long)(n0 + tt) * H + sy) * W8 + sx] : 0.f;
                    ps = __fadd_rn(ps, __fmul_rn(wv, xv));
                }
                acc = __fadd_rn(acc, ps);
            }
    }
    if (leaky && acc < 0.0f) acc = __fmul_rn(acc, 0.1f);
    out[((long)m * OH + y) * OW8 + x] = acc;
}

// legacy reorg (yolo2_model.cpp:112-129, 358-376) in the reference's [C][H][W8] layout, one frame
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

}  // namespace y2
