// kernels_f16.hpp -- fp16 MFMA path: conv as implicit GEMM on v_mfma_f32_32x32x16_f16.
//
// The floating-point form of the reference's conv (compute() fp32 branch,
// hls/core/core_compute.cpp:121-172) IS a dense contraction -- no per-group rounding -- so this
// path uses the matrix cores: fp16 activations and weights, fp32 accumulate, fp32 bias + leaky
// (x<0 ? 0.1x : x, core_compute.cpp:201-205) fused in the epilogue.  It is validated against the
// fp32 oracle at box-coordinate tolerance, not bit-exactly (different summation order/precision).
//
// Layout: NHWC with the same shared-zero-row/column trick as the int16 path (layout.hpp): one
// *item* per pixel = Cp halves (Cp = channels padded to a multiple of 32), item index
//   f = b*PL + (y+1)*Wp + x,  so the 3x3 taps are the flat offsets {-Wp-1 .. +Wp+1} and every
// out-of-image tap reads a stored zero.  GEMM view per layer:
//   M = real pixels (b,y,x),  N = output channels,  K = taps x Cp  (tap-major, channel-minor)
//   A[m][k] = act[f(m) + tapoff][c]   (64 contiguous bytes per pixel per 32-channel K-step)
//   B[k][n] = wh[n][tap][c]           (K-contiguous per output channel)
// Block = 256 threads = 2x2 wavefronts, tile 128 pixels x 128 channels, K-step 32; each wavefront
// owns 64x64 = 2x2 MFMA tiles (64 fp32 accumulators per lane).  A and B K-step tiles are staged in
// LDS (rows padded from 64 to 80 bytes: conflict-free ds_read_b128 fragments), double-buffered,
// with the global loads of step k+1 in flight while step k is multiplied.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>

#include "layout.hpp"

namespace y2 {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));

struct ConvF16Args {
    int B, H, W, Wp, PL;   // geometry shared by input and output
    int Cp_in;             // input item size in halves (multiple of 32)
    int Cp_out;            // output item size in halves
    int N;                 // real output channels
    int out_ch_off;        // channel offset inside the output item (concat placement)
    int n_store;           // channels [0, n_store) of this layer are stored (N rounded up to the item padding it owns)
    int npix;              // B*H*W
    int leaky;
    int KS;                // 1 or 3
};

constexpr int kBM = 128, kBN = 128, kBK = 32, kLdsRow = 40;  // halves per LDS row (32 + 8 pad)

__device__ __forceinline__ int flat_of_h(int q, int HW, int W, int Wp, int PL)
{
    const int b = q / HW;
    const int r = q - b * HW;
    const int y = r / W;
    const int x = r - y * W;
    return b * PL + (y + 1) * Wp + x;
}

// act: items of Cp_in halves (pointer at item 0 incl. lead); wh: [N_pad][KK][Cp_in] halves;
// bias: [N_pad] fp32; out: items of Cp_out halves; out_f32 (optional): dense [B][N][H][W] fp32.
__global__ __launch_bounds__(256) void k_conv_f16(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                   const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                   float *__restrict__ out_f32, const ConvF16Args a)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[2][kBM][kLdsRow];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][kBN][kLdsRow];
    __shared__ int fo_s[kBM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HW = a.H * a.W;
    const int q0 = blockIdx.x * kBM;
    const int n0 = blockIdx.y * kBN;
    const int KK = a.KS * a.KS;

    if (tid < kBM) fo_s[tid] = flat_of_h(min(q0 + tid, a.npix - 1), HW, a.W, a.Wp, a.PL);
    __syncthreads();

    // staging map: thread -> (row, 16-byte chunk) x 2
    const int srow = tid >> 2, schunk = tid & 3;
    const size_t a_base0 = ((size_t)kLead + fo_s[srow]) * a.Cp_in + schunk * 8;
    const size_t a_base1 = ((size_t)kLead + fo_s[srow + 64]) * a.Cp_in + schunk * 8;
    const size_t b_base0 = (size_t)(n0 + srow) * KK * a.Cp_in + schunk * 8;
    const size_t b_base1 = (size_t)(n0 + srow + 64) * KK * a.Cp_in + schunk * 8;

    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cp_in / kBK;
    const int nsteps = KK * csteps;

    auto koff = [&](int step, long &aoff, long &boff) {
        const int tap = step / csteps, c0 = (step - tap * csteps) * kBK;
        const int toff = (a.KS == 3) ? ((tap / 3 - 1) * a.Wp + (tap % 3 - 1)) : 0;
        aoff = (long)toff * a.Cp_in + c0;
        boff = (long)tap * a.Cp_in + c0;
    };

    half8_t ra0, ra1, rb0, rb1;
    {
        long ao, bo;
        koff(0, ao, bo);
        ra0 = *reinterpret_cast<const half8_t *>(act + a_base0 + ao);
        ra1 = *reinterpret_cast<const half8_t *>(act + a_base1 + ao);
        rb0 = *reinterpret_cast<const half8_t *>(wh + b_base0 + bo);
        rb1 = *reinterpret_cast<const half8_t *>(wh + b_base1 + bo);
        *reinterpret_cast<half8_t *>(&As[0][srow][schunk * 8]) = ra0;
        *reinterpret_cast<half8_t *>(&As[0][srow + 64][schunk * 8]) = ra1;
        *reinterpret_cast<half8_t *>(&Bs[0][srow][schunk * 8]) = rb0;
        *reinterpret_cast<half8_t *>(&Bs[0][srow + 64][schunk * 8]) = rb1;
    }
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool more = step + 1 < nsteps;
        if (more) {  // global loads of the next K-step fly while this one is multiplied
            long ao, bo;
            koff(step + 1, ao, bo);
            ra0 = *reinterpret_cast<const half8_t *>(act + a_base0 + ao);
            ra1 = *reinterpret_cast<const half8_t *>(act + a_base1 + ao);
            rb0 = *reinterpret_cast<const half8_t *>(wh + b_base0 + bo);
            rb1 = *reinterpret_cast<const half8_t *>(wh + b_base1 + bo);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8_t af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const half8_t *>(&As[cur][wm * 64 + t * 32 + frow][kk * 16 + fk]);
                bf[t] = *reinterpret_cast<const half8_t *>(&Bs[cur][wn * 64 + t * 32 + frow][kk * 16 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            const int nxt = cur ^ 1;
            *reinterpret_cast<half8_t *>(&As[nxt][srow][schunk * 8]) = ra0;
            *reinterpret_cast<half8_t *>(&As[nxt][srow + 64][schunk * 8]) = ra1;
            *reinterpret_cast<half8_t *>(&Bs[nxt][srow][schunk * 8]) = rb0;
            *reinterpret_cast<half8_t *>(&Bs[nxt][srow + 64][schunk * 8]) = rb1;
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = n0 + wn * 64 + j * 32 + (lane & 31);
        if (ch >= a.n_store) continue;
        const float bv = bias[ch];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int q = q0 + row;
                if (q >= a.npix) continue;
                float v = acc[i][j][r] + bv;
                if (a.leaky && v < 0.f) v *= 0.1f;
                if (out_f32) {
                    if (ch < a.N) {
                        const int b = q / HW, rem = q - b * HW;
                        out_f32[((size_t)b * a.N + ch) * HW + rem] = v;
                    }
                } else {
                    out[((size_t)kLead + fo_s[row]) * a.Cp_out + a.out_ch_off + ch] = (_Float16)v;
                }
            }
    }
}

// ------------------------------------------------------------------ small fp16 kernels

// float [B][3][416][416] -> layer-0 im2col items: 27 taps x channels (k = c*9 + i*3 + j), padded to 32
__global__ void k_pack_input_f16(const float *__restrict__ frames, _Float16 *__restrict__ out, int B, int H, int W,
                                 int Wp, int PL)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (q >= B * HW) return;
    const int b = q / HW, r = q - b * HW, y = r / W, x = r - y * W;
    _Float16 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = (_Float16)0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sy = y + i - 1, sx = x + j - 1;
                if (sy >= 0 && sy < H && sx >= 0 && sx < W) v[c * 9 + i * 3 + j] = (_Float16)frames[((size_t)b * 3 + c) * HW + (size_t)sy * W + sx];
            }
    half8_t *dst = reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * PL + (size_t)(y + 1) * Wp + x) * 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[k * 8 + e];
        dst[k] = o;
    }
}

// 2x2/2 max pool on items, one thread per (output pixel, 8-channel chunk)
__global__ void k_maxpool2_f16(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int Cp, int B, int OH, int OW,
                               int iWp, int iPL, int oWp, int oPL)
{
    const int chunks = Cp / 8;
    const long n = (long)B * OH * OW * chunks;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int ck = (int)(t % chunks);
    const long p = t / chunks;
    const int x = (int)(p % OW), y = (int)((p / OW) % OH), b = (int)(p / ((long)OW * OH));
    const size_t s = ((size_t)kLead + (size_t)b * iPL + (size_t)(2 * y + 1) * iWp + 2 * x) * Cp + ck * 8;
    const half8_t v0 = *reinterpret_cast<const half8_t *>(in + s), v1 = *reinterpret_cast<const half8_t *>(in + s + Cp);
    const half8_t v2 = *reinterpret_cast<const half8_t *>(in + s + (size_t)iWp * Cp);
    const half8_t v3 = *reinterpret_cast<const half8_t *>(in + s + (size_t)iWp * Cp + Cp);
    half8_t o = __builtin_elementwise_max(__builtin_elementwise_max(v0, v1), __builtin_elementwise_max(v2, v3));
    *reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(y + 1) * oWp + x) * Cp + ck * 8) = o;
}

// Darknet legacy reorg (yolo2_model.cpp:112-129) into channels [0,256) of the 1280-channel concat items
__global__ void k_reorg_f16(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int B, int iCp, int iWp, int iPL,
                            int oCp, int oWp, int oPL)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 256 * 169) return;
    const int b = t / (256 * 169), o = t - b * (256 * 169);
    const int k = o / (26 * 416), rem = o - k * (26 * 416), j = rem / 26, i = rem - j * 26;
    const int sidx = (2 * i + (k & 1)) + 52 * (2 * j + (k >> 1));
    const int sc = sidx / 676, sr = sidx - sc * 676, sy = sr / 26, sx = sr - sy * 26;
    const _Float16 v = in[((size_t)kLead + (size_t)b * iPL + (size_t)(sy + 1) * iWp + sx) * iCp + sc];
    const int oc = o / 169, orr = o - oc * 169, oy = orr / 13, ox = orr - oy * 13;
    out[((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * oCp + oc] = v;
}

// weights_reorg (fp32 stream of one layer) -> wh[N_pad][KK][Cp] halves (zero padded).
// im2col_first: layer 0 is run as a 1x1 conv over k = c*9 + tap (see k_pack_input_f16).
__global__ void k_pack_weights_f16(const float *__restrict__ src, _Float16 *__restrict__ dst, float *__restrict__ bias_dst,
                                   const float *__restrict__ bias_src, int C, int N, int KK, int Cp, int Npad, int im2col_first)
{
    const int KKd = im2col_first ? 1 : KK;
    const long n = (long)Npad * KKd * Cp;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Npad) bias_dst[t] = t < N ? bias_src[t] : 0.f;
    if (t >= n) return;
    const int c = (int)(t % Cp);
    const int tapd = (int)((t / Cp) % KKd);
    const int m = (int)(t / ((long)Cp * KKd));
    int ci, tap;
    if (im2col_first) { ci = c / KK; tap = c - ci * KK; if (c >= C * KK) ci = C; }
    else { ci = c; tap = tapd; }
    float v = 0.f;
    if (m < N && ci < C) {
        const int m0 = m / kTm * kTm, tm = m - m0, tm_min = min(kTm, N - m0);
        const int n0 = ci / kTn * kTn, tn = ci - n0, tn_min = min(kTn, C - n0);
        v = src[(long)m0 * C * KK + (long)tm_min * n0 * KK + (long)tap * tm_min * tn_min + tm * tn_min + tn];
    }
    dst[t] = (_Float16)v;
}

}  // namespace y2
