// conv_common.hpp -- what the int16 and the exact-fp32 conv kernels share: launch arguments, pixel-index arithmetic,
// the XCD-aware workgroup numbering, and the two templated layout kernels.  Header-only device code (inline / templates),
// safe to include from several translation units; every non-template __global__ kernel lives in exactly one kernels_*.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "layout.hpp"

namespace y2 {

typedef short short2_t __attribute__((ext_vector_type(2)));

struct ConvArgs {
    int B, H, W, Wp, PL;       // geometry shared by input and output ('same' conv, stride 1)
    int CGin;                  // input channel groups
    int CGout;                 // output channel groups that exist in the destination tensor
    int npix;                  // B*H*W
    long in_cg_stride;         // items
    long out_cg_stride;        // items
    long out_base;             // item offset of output group 0 (concat placement), incl. lead
    int shift, round;          // fast path: 0 <= shift <= 30, round = shift ? 1<<(shift-1) : 0
    int sh_right, sh_left;     // exact path: direction flags, magnitude = shift
    int bs_right, bs_left, bs_mag;  // bias shift (exact path computes it itself)
    int leaky;
    int lt_max;                // LDS tile capacity in items
    int xcd_remap;             // 0 = launch order; 1 + log2(Xm): XCDs as an (8/Xm) x Xm grid over (tiles, blocks)
    const int *mb_list;        // optional: blockIdx.y -> output-channel block (a layer whose blocks need
                               // different arithmetic forms is launched once per form); nullptr = identity
    // conv + leaky + 2x2/2 max pool fused (k_conv_i16_pool): geometry of the pooled destination tensor
    int nwin;                  // B * (H/2) * (W/2) pool windows
    int oWp, oPL;              // row pitch / plane size of the pooled tensor (items)
    long pool_cg_stride;       // items between channel groups of the pooled tensor
    long pool_base;            // item offset of its channel group 0, incl. lead
    // division by H*W, W (and, fused pool, by (H/2)*(W/2), W/2) as multiply-high + shift: layout.hpp fast_div, set by set_conv_div
    unsigned mHW, sHW, mW, sW, mOHW, sOHW, mOW, sOW;
    // K-split across workgroups (k_conv_i16_ks, small batches): grid.y = blocks x ks_S; split z runs channel groups [z ks_Q, (z + 1) ks_Q)
    // and leaves the clamp-affine triples of its sub-chains in ks_trip[z][channel item][pixel][2 pairs][a, l, h]
    int ks_S, ks_Q, ks_mb;     // splits, groups per split, blocks in this launch (grid.y / ks_S)
    int *ks_trip;
};

inline void set_conv_div(ConvArgs &a)
{
    auto one = [](unsigned d, unsigned &m, unsigned &s) { if (d < 2) { m = 0; s = 32; } else fast_div_magic(d, m, s); };   // s = 32: divisor 1
    one((unsigned)(a.H * a.W), a.mHW, a.sHW);
    one((unsigned)a.W, a.mW, a.sW);
    one((unsigned)((a.H / 2) * (a.W / 2)), a.mOHW, a.sOHW);
    one((unsigned)(a.W / 2), a.mOW, a.sOW);
}
__device__ __forceinline__ int div_c(int n, unsigned m, unsigned s) { return s >= 32 ? n : (int)fast_div((unsigned)n, m, s); }

// core_compute.cpp:191-197: x<0 ? x/10 (C division, toward zero) : x.  For u in [1,32768]
// floor(u/10) == (u*52429)>>19 (checked exhaustively in tests/test_host_logic.py).
__device__ __forceinline__ int leaky_i16(int v)
{
    const unsigned u = (unsigned)(-v);
    const int q = (int)((u * 52429u) >> 19);
    return v < 0 ? -q : v;
}

__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }

__device__ __forceinline__ long shift64(long v, int right, int left, int mag, long round)
{
    if (right) return (v + round) >> mag;
    if (left) return (long)((unsigned long)v << mag);
    return v;
}

// Pixel index q (raster over b, y, x of real pixels) -> flat item offset inside a channel group.
__device__ __forceinline__ int flat_of(const ConvArgs &a, int q)
{
    const int b = div_c(q, a.mHW, a.sHW);
    const int r = q - b * (a.H * a.W);
    const int y = div_c(r, a.mW, a.sW);
    const int x = r - y * a.W;
    return b * a.PL + (y + 1) * a.Wp + x;
}

// Workgroups are dealt to the 8 XCDs round-robin in linear launch order (x fastest) and every XCD
// has its own 4 MiB L2.  Re-number them so that the XCDs form an Xt x Xm grid over (tiles, output-
// channel blocks): XCD (kt, km) owns a contiguous range of tiles and a contiguous range of blocks and
// walks it tile-major.  Neighbouring tiles (which share halo rows) and the blocks of one tile then
// meet in one L2, the input crosses the fabric Xm times and the weights Xt times; the host picks
// the split that minimises that sum (xm_log2).  Launch order in linear id: XCD = id & 7, the slot
// within the XCD = id >> 3; XCD k receives q + (k < r) workgroups, so the logical sequence (parts in
// XCD order) is cut at exactly those counts - with uneven parts a few workgroups spill to the
// neighbouring XCD, which is harmless.
__device__ inline void xcd_partition(int xm_log2, int &tile, int &yb)
{
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const int lin = blockIdx.x + blockIdx.y * gx;
    const int xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    int L = xcd * q + min(xcd, r) + slot;               // bijection onto [0, total)
    const int Xm = 1 << xm_log2, Xt = 8 >> xm_log2;
    const int qt = gx / Xt, rt = gx - qt * Xt, qm = gy / Xm, rm = gy - qm * Xm;
    tile = 0; yb = 0;
    for (int k = 0; k < 8; ++k) {
        const int kt = k >> xm_log2, km = k & (Xm - 1);
        const int nt = qt + (kt < rt), nm = qm + (km < rm), cnt = nt * nm;
        if (L < cnt) {
            const int dt = L / nm;
            tile = kt * qt + min(kt, rt) + dt;
            yb = km * qm + min(km, rm) + (L - dt * nm);
            break;
        }
        L -= cnt;
    }
}

// weights_reorg stream of one layer -> wpk[MB][CG][KK][32][4] with partial tiles zero-padded.
// Source block (m0,n0) starts at m0*C*KK + TM_MIN*n0*KK and is [kk][TM_MIN][TN_MIN]
// (yolov2_weight_gen.cpp:43-67; consumed in this order by core_io.cpp:154-198).
template <typename T>
__global__ void k_repack_weights(const T *__restrict__ src, T *__restrict__ dst, int C, int N, int KK)
{
    const int CG = (C + kTn - 1) / kTn, MB = (N + kTm - 1) / kTm;
    const long n = (long)MB * CG * KK * 128;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int tn = (int)(t & 3), tm = (int)((t >> 2) & 31);
    const long r = t >> 7;
    const int tap = (int)(r % KK);
    const int cg = (int)((r / KK) % CG);
    const int mb = (int)(r / ((long)KK * CG));
    const int m0 = mb * kTm, n0 = cg * kTn;
    const int tm_min = min(kTm, N - m0), tn_min = min(kTn, C - n0);
    T v = 0;
    if (tm < tm_min && tn < tn_min)
        v = src[(long)m0 * C * KK + (long)tm_min * n0 * KK + (long)tap * tm_min * tn_min + tm * tn_min + tn];
    dst[t] = v;
}

// any KxK / stride pool with the reference's pad value (core_io.cpp:96-103), reference layout
template <typename T>
__global__ void k_pool_ref(const T *__restrict__ in, T *__restrict__ out, int C, int K, int stride, int W, int H,
                           int OW, int OH, T padv)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * OH * OW) return;
    const int x = t % OW, y = (t / OW) % OH, c = t / (OW * OH);
    const int W8 = (W + 7) & ~7, OW8 = (OW + 7) & ~7;
    T best = padv;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            const int sy = y * stride + i, sx = x * stride + j;
            const T v = (sy < H && sx < W) ? in[((long)c * H + sy) * W8 + sx] : padv;
            if (v > best) best = v;
        }
    out[((long)c * OH + y) * OW8 + x] = best;
}

}  // namespace y2
