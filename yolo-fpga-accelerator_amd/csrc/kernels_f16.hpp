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
// Kernel family (yolo2_hip_run_batch_fp16 picks per layer):
//   k_conv0_pool_mfma   layers 0+1: 3->32 conv + leaky + 2x2 pool straight from the float frames, im2col
//                       GEMM with K padded 27->32 (k_conv0_pool_f16 is the fp32-VALU form of the same)
//   k_conv_f16<..,32>   register-staged, K-step 32: the one layer whose items are 32 channels (layer 2)
//   k_conv_f16_glds     LDS-DMA staging per (tap, 64-channel chunk): 1x1 layers and the 104x104 3x3 layers
//   k_conv_f16_halo     3x3 layers at <= 52x52: input tile staged once per chunk with its halo, nine taps
//                       read it shifted; 256 x 256 (or 128) tiles, 8 wavefronts
// All use 32x32x16 MFMA tiles, one wavefront per 64x64 (or 128x64) of the output, double-buffered LDS
// stages and an epilogue that transposes through LDS into 16-byte stores.
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
    int n_tiles;           // output-channel tiles per pixel tile (grid = pixel tiles x n_tiles, 1-D)
    // fused 2x2/2 max pool (conv layers whose only consumer is a pool layer): the tile's MFMA rows are
    // ordered row = 4*pooled_pixel + 2*dy + dx, the epilogue stores max over each group of four rows
    // to the POOLED tensor (geometry below) and the full-resolution tensor is never written
    int pool;              // 0 / 1
    int oWp, oPL, npool;   // pooled tensor: row pitch, plane size (items), B * (H/2) * (W/2)
    // division by H*W and by W as multiply-high + shift (set_fast_div): the generic 32-bit division the compiler emits
    // is ~30 instructions, and a halo-kernel lane needs 14 of them before its first MFMA
    unsigned mHW, sHW, mW, sW;
    int stamp;             // diagnostic builds (-DY2_STAMPS): this launch records its workgroups' timeline in y2_stamps
    // split-fp16 ("fp32tol") mode, SPLIT instantiations only: every fp32 value v travels as hi = fp16(v), lo = fp16(v - hi) and an
    // item holds three parts [hi | lo | hi] of split_n channels each (the weights are packed [w_hi | w_hi | w_lo] to match, so that the
    // plain fp16 contraction over the tripled channels is a_hi w_hi + a_lo w_hi + a_hi w_lo: fp32 accuracy but for the 2^-22 lo x lo term).
    // The epilogue stores channel ch of part p at out_ch_off + p * split_n + ch.
    int split_n;
};

__device__ __forceinline__ void split_f32(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);      // exact difference (hi is v rounded to 11 bits), then rounded to fp16
}

inline void set_fast_div(ConvF16Args &a)
{
    fast_div_magic((unsigned)(a.H * a.W), a.mHW, a.sHW);
    fast_div_magic((unsigned)a.W, a.mW, a.sW);
}
// (b, y, x) of pixel index q
__device__ __forceinline__ void pixel_of(const ConvF16Args &a, int q, int &b, int &y, int &x)
{
    b = (int)fast_div((unsigned)q, a.mHW, a.sHW);
    const int r = q - b * (a.H * a.W);
    y = (int)fast_div((unsigned)r, a.mW, a.sW);
    x = r - y * a.W;
}
__device__ __forceinline__ int flat_of_fast(const ConvF16Args &a, int q)
{
    int b, y, x;
    pixel_of(a, q, b, y, x);
    return b * a.PL + (y + 1) * a.Wp + x;
}

#ifndef Y2_C0_ABL
#define Y2_C0_ABL 0         // diagnostic builds of k_conv0_pool_mfma: 1 = no global stores, 2 = cache-hot loads, 4 = no gathers / MFMAs / pooling
#endif
#ifndef Y2_ABL
#define Y2_ABL 0            // diagnostic builds only (tools/ab.sh): 1 = one fragment read per tap, 2 = no weight-tile fills, 4 = a quarter of the MFMAs, 8 = weight-tile fills from one cache-hot tile, 16 = the centre tap's rows for every tap (no per-tap address arithmetic), 32 = one barrier per nine taps
#endif
constexpr int kBN = 128;   // LDS rows are BK + 8 halves: conflict-free ds_read_b128 for BK = 32 and 64
constexpr int kCtRow = 136;  // halves per row of the epilogue staging tile (128 + 8 pad: 16-byte aligned, conflict-light)

#ifdef Y2_STAMPS
// In-kernel timeline of the halo kernel's workgroups (diagnostic build only; read back by yolo2_hip_debug_stamps):
// per workgroup 8 x u64 = {s_memrealtime at entry, s_memtime at entry, after the set-up, after the prologue fills landed,
// after the main loop, after the epilogue, HW_ID, s_memrealtime at exit}.  No output value depends on a stamp.
constexpr int kStampWGs = 16384;
__device__ unsigned long long y2_stamps[kStampWGs * 8];
#define Y2_STAMP(slot) do { if (a.stamp && tid == 0 && blockIdx.x < kStampWGs) y2_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define Y2_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int flat_of_h(int q, int HW, int W, int Wp, int PL)
{
    const int b = q / HW;
    const int r = q - b * HW;
    const int y = r / W;
    const int x = r - y * W;
    return b * PL + (y + 1) * Wp + x;
}

// Workgroups are dealt to the 8 XCDs round-robin in launch order (id & 7), and every XCD has its own 4 MiB L2.  With the
// plain 1-D grid the n-tiles of one pixel tile (which stage the SAME input tile) land on different XCDs, so the input
// crosses the fabric once per n-tile.  Re-number: XCD k owns a CONTIGUOUS range of the logical sequence (pixel tile
// major, n-tile minor), so the n-tiles of a pixel tile meet in one L2 and the pixel tiles an XCD runs side by side share
// each weight tile.  (Ablation on the 13x13 512->1024 layer at batch 256: removing the weight-tile fills saved 22 %;
// Infinity-Cache / HBM traffic, not LDS, is what they cost.)
__device__ __forceinline__ int xcd_logical_id()
{
#if defined(Y2_F16_NO_XCD)
    return (int)blockIdx.x;
#else
    const int total = (int)gridDim.x, id = (int)blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    return xcd * q + min(xcd, r) + slot;
#endif
}

// flat item offset of MFMA row m of pixel tile `tile` (BM rows per tile)
__device__ __forceinline__ int tile_row_flat(const ConvF16Args &a, int tile, int BM, int m)
{
    const int HW = a.H * a.W;
    if (!a.pool) return flat_of_h(min(tile * BM + m, a.npix - 1), HW, a.W, a.Wp, a.PL);
    const int OW = a.W / 2, OHW = (a.H / 2) * OW;
    const int pq = min(tile * (BM / 4) + (m >> 2), a.npool - 1);
    const int b = pq / OHW, r = pq - b * OHW, oy = r / OW, ox = r - oy * OW;
    return b * a.PL + (2 * oy + ((m >> 1) & 1) + 1) * a.Wp + 2 * ox + (m & 1);
}

// pooled epilogue: Ct holds the BM x BN tile (bias + leaky applied); store max over rows 4p..4p+3
template <int BM, int BN, int NT, int CTROW>
__device__ __forceinline__ void store_pooled(const _Float16 (*Ct)[CTROW], _Float16 *__restrict__ out, const ConvF16Args &a, int tile,
                                             int n0, int tid)
{
    constexpr int CH = BN / 8, RPP = NT / CH, PR = BM / 4;
    const int chunk = tid % CH, r0 = tid / CH, ch0 = n0 + chunk * 8;
    if (ch0 >= a.n_store) return;
    const int OW = a.W / 2, OHW = (a.H / 2) * OW;
#pragma unroll
    for (int rr = 0; rr < (PR + RPP - 1) / RPP; ++rr) {
        const int p = r0 + rr * RPP, pq = tile * PR + p;
        if (p >= PR || pq >= a.npool) continue;
        half8_t v = *reinterpret_cast<const half8_t *>(&Ct[4 * p][chunk * 8]);
#pragma unroll
        for (int k = 1; k < 4; ++k) v = __builtin_elementwise_max(v, *reinterpret_cast<const half8_t *>(&Ct[4 * p + k][chunk * 8]));
        const int b = pq / OHW, r = pq - b * OHW, oy = r / OW, ox = r - oy * OW;
        *reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp + ox) * a.Cp_out + a.out_ch_off + ch0) = v;
    }
}

// split mode: Ct holds the tile as FLOATS (bias + leaky applied, not yet rounded): the pool is a max of fp32 values, the winner is
// split into (hi, lo) and stored to the three parts of the pooled item.  8 channels per thread and pooled pixel.
template <int BM, int BN, int NT, int CTROW>
__device__ __forceinline__ void store_pooled_split(const float (*Ct)[CTROW], _Float16 *__restrict__ out, const ConvF16Args &a, int tile,
                                                   int n0, int tid)
{
    constexpr int CH = BN / 8, RPP = NT / CH, PR = BM / 4;
    const int chunk = tid % CH, r0 = tid / CH, ch0 = n0 + chunk * 8;
    if (ch0 >= a.n_store) return;
    const int OW = a.W / 2, OHW = (a.H / 2) * OW;
#pragma unroll
    for (int rr = 0; rr < (PR + RPP - 1) / RPP; ++rr) {
        const int p = r0 + rr * RPP, pq = tile * PR + p;
        if (p >= PR || pq >= a.npool) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Ct[4 * p][chunk * 8 + e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], Ct[4 * p + k][chunk * 8 + e]);
        half8_t hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f32(v[e], h, l); hi[e] = h; lo[e] = l; }
        const int b = pq / OHW, r = pq - b * OHW, oy = r / OW, ox = r - oy * OW;
        _Float16 *o = out + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp + ox) * a.Cp_out + a.out_ch_off + ch0;
        *reinterpret_cast<half8_t *>(o) = hi;
        *reinterpret_cast<half8_t *>(o + a.split_n) = lo;
        *reinterpret_cast<half8_t *>(o + 2 * a.split_n) = hi;
    }
}

// act: items of Cp_in halves (pointer at item 0 incl. lead); wh: [N_pad][KK][Cp_in] halves;
// bias: [N_pad] fp32; out: items of Cp_out halves; out_f32 (optional): dense [B][N][H][W] fp32.
// BN = 128: 2x2 wavefronts of 64x64;  BN = 64 (layers with <= 64 output channels): 4x1 of 32x64.
template <int BM, int BN, int BK>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64 < 256 ? 256 : (BM / 64) * (BN / 64) * 64) void k_conv_f16(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                   const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                   float *__restrict__ out_f32, const ConvF16Args a)
{
    constexpr int kBM = BM;
    constexpr int WN = BN / 64;                                     // wavefront grid: WM x WN
    constexpr int NT = ((BM / 64) * WN * 64 < 256) ? 256 : (BM / 64) * WN * 64;   // threads: 256 or 512
    constexpr int WM = NT / 64 / WN;
    constexpr int MT = kBM / WM / 32;          // 32-row MFMA tiles per wavefront along M (2 or 1)
    constexpr int kBK = BK, kLdsRow = BK + 8;
    constexpr int CPR = BK / 8;                // 16-byte chunks per staged row (4 or 8)
    constexpr int RPP = NT / CPR;              // rows staged per pass of the workgroup
    constexpr int APASS = kBM / RPP, BPASS = BN / RPP;
    // one LDS arena: K-loop staging (A: 2 x 128 x 40, B: 2 x BN x 40 halves), reused by the epilogue
    // as a [128][136] fp16 tile so that the output leaves in 16-byte stores
    constexpr int kStageHalves = 2 * kBM * kLdsRow + 2 * BN * kLdsRow;
    constexpr int kEpiHalves = kBM * kCtRow;
    __shared__ __attribute__((aligned(16))) _Float16 smem[kStageHalves > kEpiHalves ? kStageHalves : kEpiHalves];
    __shared__ int fo_s[kBM];
    _Float16 (*As)[kBM][kLdsRow] = reinterpret_cast<_Float16 (*)[kBM][kLdsRow]>(smem);
    _Float16 (*Bs)[BN][kLdsRow] = reinterpret_cast<_Float16 (*)[BN][kLdsRow]>(smem + 2 * kBM * kLdsRow);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HW = a.H * a.W;
    // 1-D grid, channel tile fastest: the gridDim.y... workgroups that share one pixel tile (the A
    // operand) are launched back to back, so A is fetched from HBM once and hit in L2 by the rest
    const int bid = xcd_logical_id();
    const int q0 = (bid / a.n_tiles) * kBM;
    const int n0 = (bid % a.n_tiles) * BN;
    const int KK = a.KS * a.KS;

    const int tile = bid / a.n_tiles;
    for (int i = tid; i < kBM; i += NT) fo_s[i] = tile_row_flat(a, tile, kBM, i);
    __syncthreads();

    // staging map: thread -> (row, 16-byte chunk), APASS / BPASS rows per thread
    const int srow = tid / CPR, schunk = tid % CPR;
    size_t a_base[APASS], b_base[BPASS];
#pragma unroll
    for (int k = 0; k < APASS; ++k) a_base[k] = ((size_t)kLead + fo_s[srow + k * RPP]) * a.Cp_in + schunk * 8;
#pragma unroll
    for (int k = 0; k < BPASS; ++k) b_base[k] = (size_t)(n0 + srow + k * RPP) * KK * a.Cp_in + schunk * 8;

    float16_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cp_in / kBK;
    const int nsteps = KK * csteps;

    // K-step -> (tap, channel chunk); steps are visited in order, so the pair is advanced
    // incrementally (no integer division in the loop)
    int k_tap = 0, k_c0 = 0;
    auto koff = [&](int /*step*/, long &aoff, long &boff) {
        const int toff = (a.KS == 3) ? ((k_tap / 3 - 1) * a.Wp + (k_tap % 3 - 1)) : 0;
        aoff = (long)toff * a.Cp_in + k_c0;
        boff = (long)k_tap * a.Cp_in + k_c0;
        k_c0 += kBK;
        if (k_c0 >= a.Cp_in) { k_c0 = 0; ++k_tap; }
    };
    half8_t ra[APASS], rb[BPASS];
    auto stage_load = [&](int step) {
        long ao, bo;
        koff(step, ao, bo);
#pragma unroll
        for (int k = 0; k < APASS; ++k) ra[k] = *reinterpret_cast<const half8_t *>(act + a_base[k] + ao);
#pragma unroll
        for (int k = 0; k < BPASS; ++k) rb[k] = *reinterpret_cast<const half8_t *>(wh + b_base[k] + bo);
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int k = 0; k < APASS; ++k) *reinterpret_cast<half8_t *>(&As[buf][srow + k * RPP][schunk * 8]) = ra[k];
#pragma unroll
        for (int k = 0; k < BPASS; ++k) *reinterpret_cast<half8_t *>(&Bs[buf][srow + k * RPP][schunk * 8]) = rb[k];
    };
    stage_load(0);
    stage_write(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool more = step + 1 < nsteps;
        if (more) stage_load(step + 1);  // global loads of the next K-step fly while this one is multiplied
        // fragment reads of sub-step kk+1 are issued before the MFMAs of sub-step kk (register
        // double buffer), so the matrix pipe does not wait on LDS latency four times per K-step
        half8_t af[2][MT], bf[2][2];
        auto read_frags = [&](int kk, int set) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                af[set][t] = *reinterpret_cast<const half8_t *>(&As[cur][wm * (32 * MT) + t * 32 + frow][kk * 16 + fk]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                bf[set][t] = *reinterpret_cast<const half8_t *>(&Bs[cur][wn * 64 + t * 32 + frow][kk * 16 + fk]);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) read_frags(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        // pin the interleave hipcc would otherwise undo: reads(0), then {reads(kk+1), MFMAs(kk)}...
        // (sched_group_barrier masks: 0x100 = DS read, 0x008 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
#pragma unroll
        for (int kk = 0; kk + 1 < BK / 16; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * 2, 0);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
    if (out_f32) {  // last layer: dense fp32 [B][N][H][W], straight from the accumulators
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = n0 + wn * 64 + j * 32 + (lane & 31);
            if (ch >= a.N) continue;
            const float bv = bias[ch];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int q = q0 + row;
                    if (q >= a.npix) continue;
                    float v = acc[i][j][r] + bv;
                    if (a.leaky && v < 0.f) v *= 0.1f;
                    const int b = q / HW, rem = q - b * HW;
                    out_f32[((size_t)b * a.N + ch) * HW + rem] = v;
                }
        }
        return;
    }
    // fp16 items: bias + leaky in fp32, transpose through LDS, leave in 16-byte (8-channel) stores
    _Float16 (*Ct)[kCtRow] = reinterpret_cast<_Float16 (*)[kCtRow]>(smem);   // the K loop ended with a barrier
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane & 31);
        const float bv = bias[n0 + col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bv;
                if (a.leaky && v < 0.f) v *= 0.1f;
                Ct[row][col] = (_Float16)v;
            }
    }
    __syncthreads();
    if (a.pool) {
        store_pooled<kBM, BN, NT, kCtRow>(Ct, out, a, tile, n0, tid);
        return;
    }
    constexpr int CH = BN / 8;               // 16-byte chunks per pixel row
    constexpr int ROWS_PER_PASS = NT / CH;
    const int chunk = tid % CH, r0 = tid / CH;
    const int ch0 = n0 + chunk * 8;
    if (ch0 < a.n_store) {
#pragma unroll
        for (int rr = 0; rr < kBM / ROWS_PER_PASS; ++rr) {
            const int row = r0 + rr * ROWS_PER_PASS;
            if (q0 + row >= a.npix) continue;
            const half8_t v = *reinterpret_cast<const half8_t *>(&Ct[row][chunk * 8]);
            *reinterpret_cast<half8_t *>(out + ((size_t)kLead + fo_s[row]) * a.Cp_out + a.out_ch_off + ch0) = v;
        }
    }
}

// ---- LDS-DMA variant (K-step 64) -------------------------------------------------------------
// Ablation of the register-staged kernel above (YOLO2_F16_DBG): of 0.505 ms for a
// 13x13x512->1024 layer at batch 256, 0.12 ms are the global->VGPR loads and 0.13 ms the
// ds_write_b128 pass that copies them into LDS; the MFMAs need 0.18 ms.  This variant lets the
// loads write LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write): each wave-instruction
// fills 1 KiB = 8 rows x 128 B of an UNPADDED tile, so bank conflicts are avoided by permuting the
// 16-byte chunk each lane fetches instead of padding rows:
//     LDS slot = chunk ^ ((row >> 1) & 7)        (conflict-free for the ds_read_b128 lane groups)
// applied on the global source address when filling and on the LDS address when reading fragments.
// All LDS lives in ONE array (a second __shared__ object makes hipcc drain the DMA before every
// fragment read, cdna_hip_programming.md section 5).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// One LDS-DMA piece (64 lanes x 16 B) from `base + off` (wave-uniform base, 32-bit byte offset per lane).  The empty asm
// keeps the zero-extension of `off` in the block of the load: instruction selection then picks the `saddr + voffset` form;
// with the extension hoisted out of a loop it falls back to a 64-bit address per lane (an add and two registers per piece).
__device__ __forceinline__ void lds_dma16(const char *base, unsigned off, _Float16 *lds)
{
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((glb_void_t *)(base + off), (lds_void_t *)lds, 16, 0, 0);
}

template <int BN, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_conv_f16_glds(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                        const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                        float *__restrict__ out_f32, const ConvF16Args a)
{
    constexpr int BM = 128, BK = 64, ROWH = BK;            // halves per (unpadded) LDS row = 128 B
    constexpr int WN = BN / 64, WM = 4 / WN, MT = BM / WM / 32;
    constexpr int AG = BM / 8 / 4, BG = BN / 8 / 4;        // 8-row groups per wavefront for A and B
    constexpr int kStageHalves = 2 * BM * ROWH + 2 * BN * ROWH;
    constexpr int kCtF = 64 + 4;                           // floats per row of the split mode's fp32 epilogue tile (64 columns per pass)
    constexpr int kEpiHalves = SPLIT ? BM * kCtF * 2 : BM * kCtRow;
    constexpr int kArena = kStageHalves > kEpiHalves ? kStageHalves : kEpiHalves;
    __shared__ __attribute__((aligned(1024))) _Float16 smem[kArena + 2 * BM];   // + fo table (BM ints) at the end
    int *fo_s = reinterpret_cast<int *>(smem + kArena);
    _Float16 *As = smem;                       // [2][BM][64]
    _Float16 *Bs = smem + 2 * BM * ROWH;       // [2][BN][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HW = a.H * a.W;
    const int bid = xcd_logical_id();
    const int q0 = (bid / a.n_tiles) * BM;
    const int n0 = (bid % a.n_tiles) * BN;
    const int KK = a.KS * a.KS;

    const int tile = bid / a.n_tiles;
    if (tid < BM) fo_s[tid] = tile_row_flat(a, tile, BM, tid);
    __syncthreads();

    // fill map: wave-instruction i covers rows R0 = (wave*G + i)*8 .. +7; lane -> (row, LDS slot); source chunk = slot ^ swz(row)
    const int lrow = lane >> 3, lslot = lane & 7;
    size_t a_src[AG], b_src[BG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int row = (wave * AG + i) * 8 + lrow;
        a_src[i] = ((size_t)kLead + fo_s[row]) * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8);
    }
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave * BG + i) * 8 + lrow;
        b_src[i] = (size_t)(n0 + row) * KK * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8);
    }

    float16_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsteps = KK * (a.Cp_in / BK);
    int k_tap = 0, k_c0 = 0;
    auto fill = [&](int buf) {   // issue the LDS-DMA loads of the next K-step into buffer `buf`
        const int toff = (a.KS == 3) ? ((k_tap / 3 - 1) * a.Wp + (k_tap % 3 - 1)) : 0;
        const long ao = (long)toff * a.Cp_in + k_c0, bo = (long)k_tap * a.Cp_in + k_c0;
        k_c0 += BK;
        if (k_c0 >= a.Cp_in) { k_c0 = 0; ++k_tap; }
#pragma unroll
        for (int i = 0; i < AG; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(act + a_src[i] + ao),
                                             (lds_void_t *)(As + ((size_t)buf * BM + (wave * AG + i) * 8) * ROWH), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BG; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(wh + b_src[i] + bo),
                                             (lds_void_t *)(Bs + ((size_t)buf * BN + (wave * BG + i) * 8) * ROWH), 16, 0, 0);
    };
    fill(0);
    __syncthreads();   // hipcc drains the DMA (vmcnt(0)) ahead of the barrier

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) fill(cur ^ 1);   // lands while this K-step is multiplied
        half8_t af[2][MT], bf[2][2];
        auto read_frags = [&](int kk, int set) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int row = wm * (32 * MT) + t * 32 + frow;
                af[set][t] = *reinterpret_cast<const half8_t *>(As + ((size_t)cur * BM + row) * ROWH + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) * 8));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = wn * 64 + t * 32 + frow;
                bf[set][t] = *reinterpret_cast<const half8_t *>(Bs + ((size_t)cur * BN + row) * ROWH + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) * 8));
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) read_frags(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
#pragma unroll
        for (int kk = 0; kk + 1 < BK / 16; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * 2, 0);
        __syncthreads();
    }

    if (out_f32) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = n0 + wn * 64 + j * 32 + (lane & 31);
            if (ch >= a.N) continue;
            const float bv = bias[ch];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int q = q0 + row;
                    if (q >= a.npix) continue;
                    float v = acc[i][j][r] + bv;
                    if (a.leaky && v < 0.f) v *= 0.1f;
                    const int b = q / HW, rem = q - b * HW;
                    out_f32[((size_t)b * a.N + ch) * HW + rem] = v;
                }
        }
        return;
    }
    if constexpr (SPLIT) {
        // fp32 tile (bias + leaky in fp32, nothing rounded yet), 64 columns per pass (the whole 128 x 128 fp32 tile would not fit the
        // static LDS limit), then per 8 channels: pool (if fused), split into (hi, lo), three stores
        float (*Cf)[kCtF] = reinterpret_cast<float (*)[kCtF]>(smem);
#pragma unroll
        for (int hcol = 0; hcol < WN; ++hcol) {
            if (hcol) __syncthreads();       // the previous pass's tile has been read
            if (wn == hcol) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = j * 32 + (lane & 31);
                    const float bv = bias[n0 + hcol * 64 + col];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            float v = acc[i][j][r] + bv;
                            if (a.leaky && v < 0.f) v *= 0.1f;
                            Cf[row][col] = v;
                        }
                }
            }
            __syncthreads();
            if (a.pool) {
                store_pooled_split<BM, 64, 256, kCtF>(Cf, out, a, tile, n0 + hcol * 64, tid);
                continue;
            }
            constexpr int CH = 8, ROWS_PER_PASS = 256 / CH;
            const int chunk = tid % CH, r0 = tid / CH, ch0 = n0 + hcol * 64 + chunk * 8;
            if (ch0 < a.n_store) {
#pragma unroll
                for (int rr = 0; rr < BM / ROWS_PER_PASS; ++rr) {
                    const int row = r0 + rr * ROWS_PER_PASS;
                    if (q0 + row >= a.npix) continue;
                    half8_t hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f32(Cf[row][chunk * 8 + e], h, l); hi[e] = h; lo[e] = l; }
                    _Float16 *o = out + ((size_t)kLead + fo_s[row]) * a.Cp_out + a.out_ch_off + ch0;
                    *reinterpret_cast<half8_t *>(o) = hi;
                    *reinterpret_cast<half8_t *>(o + a.split_n) = lo;
                    *reinterpret_cast<half8_t *>(o + 2 * a.split_n) = hi;
                }
            }
        }
        return;
    }
    _Float16 (*Ct)[kCtRow] = reinterpret_cast<_Float16 (*)[kCtRow]>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane & 31);
        const float bv = bias[n0 + col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bv;
                if (a.leaky && v < 0.f) v *= 0.1f;
                Ct[row][col] = (_Float16)v;
            }
    }
    __syncthreads();
    if (a.pool) {
        store_pooled<BM, BN, 256, kCtRow>(Ct, out, a, tile, n0, tid);
        return;
    }
    constexpr int CH = BN / 8, ROWS_PER_PASS = 256 / CH;
    const int chunk = tid % CH, r0 = tid / CH, ch0 = n0 + chunk * 8;
    if (ch0 < a.n_store) {
#pragma unroll
        for (int rr = 0; rr < BM / ROWS_PER_PASS; ++rr) {
            const int row = r0 + rr * ROWS_PER_PASS;
            if (q0 + row >= a.npix) continue;
            const half8_t v = *reinterpret_cast<const half8_t *>(&Ct[row][chunk * 8]);
            *reinterpret_cast<half8_t *>(out + ((size_t)kLead + fo_s[row]) * a.Cp_out + a.out_ch_off + ch0) = v;
        }
    }
}

// ---- halo-tile variant for 3x3 layers (BM = 256, 8 wavefronts) -------------------------------
// The LDS-DMA kernel above is bound by operand movement: it stages a fresh A tile (128 pixels x 64
// channels) for every one of the 9 taps although the taps read the SAME pixels shifted by one row
// or column - at 100 % MFMA rate that asks the vector memory path for its full 64 B/clk/CU.  Here
// the input tile of one 64-channel chunk is staged ONCE with its halo and the nine taps read their
// A fragments from it at row offsets {-W-1 .. +W+1}; only the weight tile is staged per tap.
// Staged bytes per FLOP drop ~3x.  K order is channel-chunk-major, tap-minor (a different fp32
// summation order than the kernels above; same tolerance).
// The tile is DENSE in pixel index (LDS row R = pixel q0 - W - 1 + R, no pad rows): the lanes of a
// ds_read_b128 group then always read 16 rows with distinct residues mod 16, which the XOR swizzle
// (slot = chunk ^ ((row >> 1) & 7), keyed on the LDS row) maps to 16 distinct bank groups for every
// tap.  (A first version staged the flat run incl. the layout's pad pixels: wherever a lane group
// straddled an image-row end its rows skipped one and 39 % of the LDS cycles were bank conflicts.)
// A dense tile has no zeros for out-of-image taps, so a lane whose tap leaves the image reads a
// dedicated all-zero row instead (one v_cndmask on the address; identical addresses broadcast).
// FUSE1 (BN = 256, 16 wavefronts, 32x32 MFMAs; layer 8 + layer 9): the tile holds ALL output channels of the 3x3 and its only consumer
// is the 1x1 after it (256 -> 128): the leaky'd fp16 tile the epilogue lays out in LDS as [pixel][channel] is exactly the 1x1's
// A operand, so the 1x1 runs on it right there - 32 more MFMAs per wavefront (64 pixels x 32 of the 128 output channels; its weight
// fragments straight from L2 into registers) - and only the 1x1's tensor is stored (wh2: [128][1][256] halves; a.Cp_out / a.n_store
// describe THAT tensor).  The 256-channel tensor between the two layers is never written or read.
template <int BN, int NB, int NW = 8, int TS = 32, bool SPLIT = false, bool FUSE1 = false>
__global__ __launch_bounds__(NW * 64) void k_conv_f16_halo(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                        const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                        const ConvF16Args a, const int lt_rows, const _Float16 *__restrict__ wh2 = nullptr,
                                                        const float *__restrict__ bias2 = nullptr)
{
    static_assert(!FUSE1 || (BN == 256 && NW == 16 && TS == 32 && !SPLIT), "the fused 1x1 is built for the 256-channel, 16-wavefront tile");
    constexpr int BM = 256, BK = 64, ROWH = BK, NT = NW * 64;
    // BN = 128: 4 x 2 wavefronts of 64 x 64, NB = 3 weight-tile buffers;
    // BN = 256: 2 x 4 wavefronts of 128 x 64 (NW = 8) or 4 x 4 of 64 x 64 (NW = 16: four wavefronts per SIMD fill the
    //           bubbles around the per-tap barrier better, +4 %), NB = 2
    // TS = 32: v_mfma_f32_32x32x16_f16; TS = 16: v_mfma_f32_16x16x32_f16 - the same FLOPs per cycle and the same LDS bytes
    // per wavefront tile, but the chip, which holds its clock down under this load (~1.75 GHz), holds a higher clock on
    // the small shape (MI355X_MICROARCH.md, DVFS give-back item 7)
    constexpr int WN = BN / 64, WM = NW / WN, MT = BM / WM / TS, NJ = 64 / TS;   // MT x NJ MFMA tiles per wavefront
    constexpr int KQ = 64 / TS, KSL = 8 * KQ, NKS = BK / KSL;                     // lanes' k-groups, k per MFMA, MFMA k-slices per step
    typedef float acc_t __attribute__((ext_vector_type(TS == 32 ? 16 : 4)));
    constexpr int BG = BN / 8 / NW;                                // B fill instructions per wavefront and tap
    constexpr int kCt = BN + 8;                                    // halves per row of the epilogue staging tile
    constexpr int kMaxAIters = 64 / NW;                            // A fill instructions per wavefront (host: lt_rows <= 64*8 + 8)
    static_assert((MT * TS == 64 || MT * TS == 128) && (TS == 32 || TS == 16) && BG >= 1 && (NB == 2 || NB == 3) && (NW == 8 || NW == 16), "tile shape");
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    _Float16 *As = smem_h;                                   // [2][lt_rows][64]; rows lt_rows-8.. of each buffer stay zero
    _Float16 *Bs = smem_h + (size_t)2 * lt_rows * ROWH;      // [NB][BN][64]: with NB = 3 the weight tile of tap t+2 is in flight while t is multiplied
    int *fo_s = reinterpret_cast<int *>(Bs + NB * BN * ROWH); // [BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef Y2_STAMPS
    if (a.stamp && tid == 0 && blockIdx.x < kStampWGs) {
        y2_stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        y2_stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |          // HW_ID: cu_id [11:8], sh [12], se [15:13]
                                                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // XCC_ID
    }
#endif
    Y2_STAMP(1);
    const int bid = xcd_logical_id();
    const int q0 = (bid / a.n_tiles) * BM;
    const int n0 = (bid % a.n_tiles) * BN;
    const int d0 = q0 - a.W - 1;                 // pixel index of LDS row 0
    const int NA = lt_rows / 8 - 1;              // 8-row groups filled by DMA; the last group is the zero rows
    const int zrow = NA * 8;

    if (tid < BM) fo_s[tid] = flat_of_fast(a, min(q0 + tid, a.npix - 1));
    if (tid < 2 * 8 * 8) {                       // zero rows of both buffers: 2 x 8 rows x 128 B = 128 x 16 B
        const int buf = tid >> 6, r = (tid >> 3) & 7, c = tid & 7;
        *reinterpret_cast<int4 *>(As + ((size_t)buf * lt_rows + zrow + r) * ROWH + c * 8) = make_int4(0, 0, 0, 0);
    }
    __syncthreads();

    const int lrow = lane >> 3, lslot = lane & 7;
    const int a_iters = (NA + NW - 1) / NW;
    // Per-lane sources as 32-bit BYTE offsets from a wave-uniform base (the tensors stay below 4 GiB: checked by the
    // host): the LDS-DMA instruction then takes `saddr + 32-bit voffset` instead of a 64-bit address per lane, which
    // halves what its issue has to move.
    unsigned a_src[kMaxAIters];                  // source of this lane's 16 bytes in each of its A fill instructions
#pragma unroll
    for (int it = 0; it < kMaxAIters; ++it) {
        const int row = (wave + it * NW) * 8 + lrow;
        const int d = min(max(d0 + row, 0), a.npix - 1);     // rows outside the tensor are only ever read masked
        a_src[it] = (unsigned)((((size_t)kLead + flat_of_fast(a, d)) * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8)) * 2);
    }
    unsigned b_src[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave * BG + i) * 8 + lrow;
        b_src[i] = (unsigned)(((size_t)row * 9 * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8)) * 2);
    }
    const char *wbase = reinterpret_cast<const char *>(wh + (size_t)n0 * 9 * a.Cp_in);   // this n-tile's weights (uniform)
    auto fill_a = [&](int buf, int c0) {        // the whole halo tile of one 64-channel chunk
        const char *abase = reinterpret_cast<const char *>(act + c0);
#pragma unroll
        for (int it = 0; it < kMaxAIters; ++it) {
            const int g = wave + it * NW;
            if (it < a_iters && g < NA)
                lds_dma16(abase, a_src[it], As + ((size_t)buf * lt_rows + g * 8) * ROWH);
        }
    };
    auto fill_b = [&](int buf, int tap, int c0) {
        const char *bb = wbase + ((long)tap * a.Cp_in + c0) * 2;
#pragma unroll
        for (int i = 0; i < BG; ++i)
            lds_dma16(bb, b_src[i], Bs + ((size_t)buf * BN + (wave * BG + i) * 8) * ROWH);
    };

    acc_t acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < (TS == 32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;

    const int frow = lane & (TS - 1), fhalf = lane / TS;   // fragment row, k-group (8 halves) of this lane
    int lo[MT];                                  // LDS row of this lane's A rows at the centre tap
    unsigned tapmask[MT];                        // bit t: tap t of that pixel lies inside the image
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = wm * (TS * MT) + t * TS + frow;
        lo[t] = m + a.W + 1;
        int bq, y, x;
        pixel_of(a, min(q0 + m, a.npix - 1), bq, y, x);
        unsigned mk = 0;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) mk |= 1u << tp;
        }
        tapmask[t] = mk;
    }
    int brow[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) brow[t] = wn * 64 + t * TS + frow;

    const int csteps = a.Cp_in / BK;
    const int nsteps = csteps * 9;
    Y2_STAMP(2);
    fill_a(0, 0);
    fill_b(0, 0, 0);
    if (NB == 3) fill_b(1, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Y2_STAMP(3);

#ifdef Y2_HALO_PRIO
    // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4: the later-dispatched half of a big workgroup loses every
    //  arbitration at equal priority; a static s_setprio 1 for it before the main loop)
    if (NW == 16 && wave >= NW / 2) __builtin_amdgcn_s_setprio(Y2_HALO_PRIO);
#endif
    int step = 0, cur = 0;                // cur = step % NB
    int n_tap = NB - 1, n_c0 = 0;         // (tap, channel offset) of step + NB - 1
    // (fetching the next tap's first A fragments ahead of the barrier that ends a tap was tried: -1 %)
    for (int ci = 0; ci < csteps; ++ci) {
        const int abuf = ci & 1;
        if (ci + 1 < csteps) fill_a(abuf ^ 1, (ci + 1) * BK);   // lands during the nine taps of this chunk
        const _Float16 *At = As + (size_t)abuf * lt_rows * ROWH;
        // (Not unrolled: with the nine taps written out hipcc hoists their address arithmetic above the chunk loop - 77 spilled registers.
        //  Per tap-step a wavefront issues ~34 VALU address instructions beside its 16 fragment reads, 2 LDS-DMA pieces and 16 MFMAs; they
        //  are not what the loop waits for: with the arithmetic hoisted out of the loop - Y2_ABL = 16, wrong results - the 13 x 13,
        //  26 x 26 and 52 x 52 layers run 1.5-2.5 % faster, no more.)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const bool more = step + NB - 1 < nsteps;
            // (Issuing this fill after the MFMAs of the first or second k-slice instead - so that the sixteen wavefronts do
            //  not queue their pieces at once straight after the barrier - measured -1.7 % / -3 %: the fill then lands late.)
            if (more) {
#if (Y2_ABL & 8)
                fill_b(cur == 0 ? NB - 1 : cur - 1, 0, 0);          // diagnostic: always the same (cache-hot) weight tile
#elif !(Y2_ABL & 2)
                fill_b(cur == 0 ? NB - 1 : cur - 1, n_tap, n_c0);   // buffer (step + NB - 1) % NB: last read in step - 1
#endif
                if (++n_tap == 9) { n_tap = 0; n_c0 += BK; }
            }
            const int toff = (tap / 3 - 1) * a.W + (tap % 3 - 1);
            int arow[MT], asw[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#if (Y2_ABL & 16)
                arow[t] = lo[t];                 // diagnostic: the centre tap's rows for every tap - the address arithmetic leaves the loop
#else
                arow[t] = ((tapmask[t] >> tap) & 1) ? lo[t] + toff : zrow;
#endif
                asw[t] = (arow[t] >> 1) & 7;
            }
#if (Y2_ABL & 16)
            const _Float16 *Bt = Bs;             // (and one weight buffer)
#else
            const _Float16 *Bt = Bs + (size_t)cur * BN * ROWH;
#endif
            half8_t af[TS == 32 ? 2 : 1][MT], bf[TS == 32 ? 2 : 1][NJ];
            auto read_frags = [&](int kk, int set) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    af[set][t] = *reinterpret_cast<const half8_t *>(At + (size_t)arow[t] * ROWH + (((kk * KQ + fhalf) ^ asw[t]) * 8));
#pragma unroll
                for (int t = 0; t < NJ; ++t)
                    bf[set][t] = *reinterpret_cast<const half8_t *>(Bt + (size_t)brow[t] * ROWH + (((kk * KQ + fhalf) ^ ((brow[t] >> 1) & 7)) * 8));
            };
            // (TS = 16: eight fragments per k-slice - a second set for the next slice does not fit 128 registers beside the 64
            //  accumulators, so each slice's reads are issued when the previous slice's MFMAs have their operands; the other
            //  three wavefronts of the SIMD cover the wait)
            constexpr bool kDouble = TS == 32;
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) {
                if constexpr (!kDouble) {
                    if (kk > 0) read_frags(kk, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                    continue;
                }
#if (Y2_ABL & 1)
                if (kk + 1 < NKS) { af[(kk + 1) & 1][0] = af[kk & 1][0]; af[(kk + 1) & 1][MT - 1] = af[kk & 1][MT - 1]; bf[(kk + 1) & 1][0] = bf[kk & 1][0]; bf[(kk + 1) & 1][1] = bf[kk & 1][1]; }
#else
                if (kk + 1 < NKS) read_frags(kk + 1, (kk + 1) & 1);
#endif
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#if (Y2_ABL & 4)
                        if (kk == 0)
#endif
                    {
                        if constexpr (TS == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
                    }
            }
            // interleave: the fragment reads of k-slice kk+1 go out between the MFMAs of slice kk
            if constexpr (kDouble) {
                __builtin_amdgcn_sched_group_barrier(0x100, MT + NJ, 0);
#pragma unroll
                for (int kk = 0; kk + 1 < NKS; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NJ, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NJ, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NJ, 0);
            }
            // Loads return in order: all but the weight tile issued in THIS step (BG instructions per wave) have
            // landed, i.e. step + 1's weight tile and, in a chunk's first tap, the next chunk's input tile.
#if (Y2_ABL & 32)
            if (tap == 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }      // diagnostic: one barrier per chunk instead of one per tap
#else
            if (more && NB == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#endif
            cur = cur == NB - 1 ? 0 : cur + 1;
        }
    }

    Y2_STAMP(4);
    // epilogue: bias + leaky, transposed through LDS into 16-byte stores (the staging arena is free now)
    _Float16 (*Ct)[kCt] = reinterpret_cast<_Float16 (*)[kCt]>(smem_h);
    int fo_r[BM / (NT / (BN / 8))];   // fo_s lives behind the arena the Ct tile may overlap: keep what this thread needs
    constexpr int CH = BN / 8, ROWS_PER_PASS = NT / CH;
    const int chunk = tid % CH, r0 = tid / CH, ch0 = n0 + chunk * 8;
#pragma unroll
    for (int rr = 0; rr < BM / ROWS_PER_PASS; ++rr) fo_r[rr] = fo_s[r0 + rr * ROWS_PER_PASS];
    __syncthreads();
    // split mode: two passes through the same fp16 tile - the hi halves (stored to parts 0 and 2 of the item), then the lo halves (part 1)
#pragma unroll
    for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {
        if (pass) __syncthreads();           // the hi tile has been read
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = wn * 64 + j * TS + frow;
            const float bv = bias[n0 + col];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < (TS == 32 ? 16 : 4); ++r) {
                    // D layout: 32x32: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); 16x16: row = r + 4 (lane >> 4); column = lane % TS
                    const int row = wm * (TS * MT) + i * TS + (TS == 32 ? (r & 3) + 8 * (r >> 2) + 4 * fhalf : r + 4 * fhalf);
                    float v = acc[i][j][r] + bv;
                    if (a.leaky && v < 0.f) v *= 0.1f;
                    if constexpr (SPLIT) {
                        _Float16 h, l;
                        split_f32(v, h, l);
                        Ct[row][col] = pass ? l : h;
                    } else
                        Ct[row][col] = (_Float16)v;
                }
        }
        __syncthreads();
        if constexpr (FUSE1) {
            // out2[pixel][32 wn + ...] = sum over the tile's 256 channels; operands swapped (D rows = output channels), so a lane ends up
            // with 4 consecutive channels of a pixel per register group and v_permlane32_swap pairs the lane halves into 16-byte stores
            half8_t b2[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) b2[s2] = *reinterpret_cast<const half8_t *>(wh2 + (size_t)(32 * wn + frow) * 256 + 16 * s2 + 8 * fhalf);
            float16_t acc2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const half8_t cf = *reinterpret_cast<const half8_t *>(&Ct[wm * 64 + i * 32 + frow][16 * s2 + 8 * fhalf]);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2[s2], cf, acc2[i], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = q0 + wm * 64 + i * 32 + frow;
                const bool qok = q < a.npix;
                _Float16 *orow = out + ((size_t)kLead + flat_of_fast(a, min(q, a.npix - 1))) * a.Cp_out + a.out_ch_off;
                unsigned pk[4][2];       // [group g = r >> 2][dword]: this lane's channels 8 g + 4 (lane >> 5) .. + 3 of the 32-channel block
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias2 + 32 * wn + 8 * g + 4 * fhalf);
                    float v0 = acc2[i][4 * g + 0] + bv.x, v1 = acc2[i][4 * g + 1] + bv.y, v2 = acc2[i][4 * g + 2] + bv.z, v3 = acc2[i][4 * g + 3] + bv.w;
                    if (a.leaky) { v0 = fmaxf(v0, v0 * 0.1f); v1 = fmaxf(v1, v1 * 0.1f); v2 = fmaxf(v2, v2 * 0.1f); v3 = fmaxf(v3, v3 * 0.1f); }
                    const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                    pk[g][0] = __builtin_bit_cast(unsigned, h01);
                    pk[g][1] = __builtin_bit_cast(unsigned, h23);
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                    const uint2v s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                    const uint2v s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                    const int c0 = 32 * wn + 8 * (2 * pr + fhalf);
                    if (qok && c0 < a.n_store) *reinterpret_cast<uint4 *>(orow + c0) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        } else
        if (ch0 < a.n_store) {
#pragma unroll
            for (int rr = 0; rr < BM / ROWS_PER_PASS; ++rr) {
                const int row = r0 + rr * ROWS_PER_PASS;
                if (q0 + row >= a.npix) continue;
                const half8_t v = *reinterpret_cast<const half8_t *>(&Ct[row][chunk * 8]);
                _Float16 *o = out + ((size_t)kLead + fo_r[rr]) * a.Cp_out + a.out_ch_off + ch0;
                if constexpr (SPLIT) {
                    if (pass == 0) { *reinterpret_cast<half8_t *>(o) = v; *reinterpret_cast<half8_t *>(o + 2 * a.split_n) = v; }
                    else *reinterpret_cast<half8_t *>(o + a.split_n) = v;
                } else
                    *reinterpret_cast<half8_t *>(o) = v;
            }
        }
    }
#ifdef Y2_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Y2_STAMP(5);
    if (a.stamp && tid == 0 && blockIdx.x < kStampWGs) y2_stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- persistent halo-tile kernel ----------------------------------------------------------------
// tools/stamps.py on k_conv_f16_halo (profiles/r02_f16_halo_wg_timeline.txt): a workgroup spends ~3.0k cycles setting up,
// ~4.3k waiting for its first tiles, ~9.6k in the LDS-transposed epilogue, and the CU then sits ~1.3 us until the next
// 1024-thread workgroup starts - ~19k cycles per tile that no MFMA covers, 28 % of a 52x52 tile's life, 16 % at 26x26.
// Here the workgroup is PERSISTENT and the (tile, chunk, tap) steps of all its tiles form one sequence through the same
// two input-tile buffers and two weight-tile buffers: the next tile's first input tile is staged during this tile's
// last chunk, its first weight tile during this tile's last tap, and the epilogue needs no LDS (operands swapped: a lane
// ends up with 4 consecutive channels of a pixel, see k_gemm1_f16_p), so the next tile's MFMAs start while the stores
// drain.  Same dense input tile, swizzle and zero-row masking as k_conv_f16_halo.  NB = 2.
template <int BN, int NW, int TS, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void k_conv_f16_halo_p(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                          const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                          const ConvF16Args a, const int lt_rows, const int n_tile_total)
{
    constexpr int BM = 256, BK = 64, ROWH = BK;
    constexpr int WN = BN / 64, WM = NW / WN, MT = BM / WM / TS, NJ = 64 / TS;   // MT x NJ MFMA tiles per wavefront (64 x 64)
    constexpr int KQ = 64 / TS, KSL = 8 * KQ, NKS = BK / KSL;                     // lanes' k-groups, k per MFMA, k-slices per step
    constexpr int NR = TS == 32 ? 16 : 4;                                         // accumulator registers per MFMA tile
    typedef float acc_t __attribute__((ext_vector_type(NR)));
    constexpr int BG = BN / 8 / NW, kMaxAIters = 64 / NW;
    constexpr bool kDouble = TS == 32;                                            // second fragment set (see k_conv_f16_halo)
    static_assert(MT * TS == 64 && BG >= 1 && (NW == 8 || NW == 16), "tile shape");
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    _Float16 *As = smem_h;                                   // [2][lt_rows][64]; rows lt_rows-8.. of each buffer stay zero
    _Float16 *Bs = smem_h + (size_t)2 * lt_rows * ROWH;      // [2][BN][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int NA = lt_rows / 8 - 1, zrow = NA * 8;
    const int xcd = (int)blockIdx.x & 7, S = (int)gridDim.x >> 3;
    const int t_first = (int)((long)n_tile_total * xcd / 8) + ((int)blockIdx.x >> 3);
    const int t_end = (int)((long)n_tile_total * (xcd + 1) / 8);
    const int my_n = t_first < t_end ? (t_end - t_first + S - 1) / S : 0;
    if (my_n == 0) return;
#ifdef Y2_STAMPS
    if (a.stamp && tid == 0 && blockIdx.x < kStampWGs) {   // slots: 0 real start, 1 start, 2 first tiles staged, 3 / 4 first tile's loop / epilogue done, 5 end, 6 tiles, 7 real end
        y2_stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        y2_stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)my_n;
    }
#endif
    Y2_STAMP(1);
    const int csteps = a.Cp_in / BK, tile_steps = csteps * 9, total_steps = my_n * tile_steps, total_chunks = my_n * csteps;

    if (tid < 2 * 8 * 8) {                       // zero rows of both input buffers
        const int buf = tid >> 6, r = (tid >> 3) & 7, c = tid & 7;
        *reinterpret_cast<int4 *>(As + ((size_t)buf * lt_rows + zrow + r) * ROWH + c * 8) = make_int4(0, 0, 0, 0);
    }

    const int lrow = lane >> 3, lslot = lane & 7;
    const int a_iters = (NA + NW - 1) / NW;
    unsigned b_src[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave * BG + i) * 8 + lrow;
        b_src[i] = ((unsigned)row * 9u * (unsigned)a.Cp_in + (unsigned)((lslot ^ ((row >> 1) & 7)) * 8)) * 2u;
    }
    // ---- producers: the input tile of chunk (fa_i, fa_c) and the weight tile of step (fb_i, fb_c, fb_tap)
    // (The per-lane source offsets of the input tile are recomputed at every fill - ~60 VALU instructions per nine taps -
    //  rather than kept: with them live across the tap loop the 16-wavefront shape no longer fits its 128 registers.)
    int fa_i = 0, fa_c = 0, fa_buf = 0, chunks_issued = 0;
    auto fill_a = [&]() {
        const char *abase = reinterpret_cast<const char *>(act + fa_c * BK);
        const int t = t_first + fa_i * S, d0 = (t / a.n_tiles) * BM - a.W - 1;
        int lr = lrow, ls = lslot;
        asm volatile("" : "+v"(lr), "+v"(ls));   // (keeps the per-piece row / swizzle terms from being hoisted out of the tile loop and spilled)
#pragma unroll
        for (int it = 0; it < kMaxAIters; ++it) {
            const int g = wave + it * NW;
            if (it < a_iters && g < NA) {
                const int row = g * 8 + lr;
                const int d = min(max(d0 + row, 0), a.npix - 1);     // rows outside the tensor are only ever read masked
                const unsigned src = ((unsigned)(kLead + flat_of_fast(a, d)) * (unsigned)a.Cp_in + (unsigned)((ls ^ ((row >> 1) & 7)) * 8)) * 2u;
                lds_dma16(abase, src, As + ((size_t)fa_buf * lt_rows + g * 8) * ROWH);
            }
        }
        fa_buf ^= 1;
        ++chunks_issued;
        if (++fa_c == csteps) { fa_c = 0; ++fa_i; }
    };
    int fb_i = 0, fb_c = 0, fb_tap = 0, fb_buf = 0, steps_issued = 0;
    const char *fb_wbase = reinterpret_cast<const char *>(wh + (size_t)((t_first % a.n_tiles) * BN) * 9 * a.Cp_in);
    auto fill_b = [&]() {
        const char *bb = fb_wbase + ((long)fb_tap * a.Cp_in + fb_c * BK) * 2;
#pragma unroll
        for (int i = 0; i < BG; ++i)
            lds_dma16(bb, b_src[i], Bs + ((size_t)fb_buf * BN + (wave * BG + i) * 8) * ROWH);
        fb_buf ^= 1;
        ++steps_issued;
        if (++fb_tap == 9) {
            fb_tap = 0;
            if (++fb_c == csteps) {
                fb_c = 0;
                ++fb_i;
                const int t = t_first + min(fb_i, my_n - 1) * S;
                fb_wbase = reinterpret_cast<const char *>(wh + (size_t)((t % a.n_tiles) * BN) * 9 * a.Cp_in);
            }
        }
    };
    __syncthreads();                             // zero rows written
    fill_a();
    fill_b();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Y2_STAMP(2);

    const int frow = lane & (TS - 1), fhalf = lane / TS;   // fragment row, k-group (8 halves) of this lane
    int lo[MT];                                  // LDS row of this lane's pixel rows at the centre tap (the same for every tile)
#pragma unroll
    for (int u = 0; u < MT; ++u) lo[u] = wm * 64 + u * TS + frow + a.W + 1;
    int brow[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) brow[u] = wn * 64 + u * TS + frow;

    int cur_b = 0, cur_a = 0;
    for (int ti = 0; ti < my_n; ++ti) {
        const int t = t_first + ti * S, pt = t / a.n_tiles;
        const int q0 = pt * BM, n0 = (t - pt * a.n_tiles) * BN;
        unsigned tapmask[MT];                    // bit t: tap t of that pixel lies inside the image
        int fr = lane;
        asm volatile("" : "+v"(fr));             // (tile-loop invariants derived from the lane id are recomputed, not kept)
        fr &= TS - 1;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            int bq, y, x;
            pixel_of(a, min(q0 + wm * 64 + u * TS + fr, a.npix - 1), bq, y, x);
            unsigned mk = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) mk |= 1u << tp;
            }
            tapmask[u] = mk;
        }
        acc_t acc[MT][NJ];                       // [pixel tile][channel tile]: D rows = channels, D columns = pixels
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

        for (int ci = 0; ci < csteps; ++ci) {
            if (chunks_issued < total_chunks) fill_a();     // the next chunk's input tile (possibly the next tile's first)
            const _Float16 *At = As + (size_t)cur_a * lt_rows * ROWH;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (steps_issued < total_steps) fill_b();   // the next step's weight tile
                const int toff = (tap / 3 - 1) * a.W + (tap % 3 - 1);
                int arow[MT], asw[MT];
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    arow[u] = ((tapmask[u] >> tap) & 1) ? lo[u] + toff : zrow;
                    asw[u] = (arow[u] >> 1) & 7;
                }
                const _Float16 *Bt = Bs + (size_t)cur_b * BN * ROWH;
                half8_t af[kDouble ? 2 : 1][MT], bf[kDouble ? 2 : 1][NJ];
                auto read_frags = [&](int kk, int set) {
#pragma unroll
                    for (int u = 0; u < MT; ++u)
                        af[set][u] = *reinterpret_cast<const half8_t *>(At + (size_t)arow[u] * ROWH + (((kk * KQ + fhalf) ^ asw[u]) * 8));
#pragma unroll
                    for (int u = 0; u < NJ; ++u)
                        bf[set][u] = *reinterpret_cast<const half8_t *>(Bt + (size_t)brow[u] * ROWH + (((kk * KQ + fhalf) ^ ((brow[u] >> 1) & 7)) * 8));
                };
                read_frags(0, 0);
#pragma unroll
                for (int kk = 0; kk < NKS; ++kk) {
                    const int set = kDouble ? (kk & 1) : 0;
                    if constexpr (kDouble) {
                        if (kk + 1 < NKS) read_frags(kk + 1, (kk + 1) & 1);
                    } else {
                        if (kk > 0) read_frags(kk, 0);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if constexpr (TS == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[set][j], af[set][i], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[set][j], af[set][i], acc[i][j], 0, 0, 0);
                        }
                }
                if constexpr (kDouble) {         // the fragment reads of k-slice kk+1 go out between the MFMAs of slice kk
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NJ, 0);
#pragma unroll
                    for (int kk = 0; kk + 1 < NKS; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x100, MT + NJ, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, MT * NJ, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NJ, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur_b ^= 1;
            }
            cur_a ^= 1;
        }

        if (ti == 0) Y2_STAMP(3);
        // ---- epilogue straight from the accumulators (no LDS; the next tile's first tiles are already staged).
        // Bias is added AFTER the sum like in k_conv_f16_halo, so both kernels give the same bits.  The bias values of one
        // channel tile are loaded just before use (the empty asm stops the compiler from hoisting all of them above the
        // epilogue, which no longer fits 128 registers beside the 64 accumulators).
        int fe = lane;
        asm volatile("" : "+v"(fe));
        const int fhe = fe / TS;
        fe &= TS - 1;
        unsigned orow[MT];                       // 32-bit byte offset of each pixel's item (output tensor below 4 GiB: host check)
        bool qok[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = q0 + wm * 64 + i * TS + fe;
            qok[i] = q < a.npix;
            orow[i] = (unsigned)(kLead + flat_of_fast(a, min(q, a.npix - 1))) * (unsigned)(a.Cp_out * 2);
        }
        char *obase = reinterpret_cast<char *>(out + a.out_ch_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cb = n0 + wn * 64 + j * TS;
            int fhj = fhe;
            asm volatile("" : "+v"(fhj));
            float4 bv[NR / 4];                   // register r of a 32x32 tile = channel 8 (r >> 2) + 4 (lane >> 5) + (r & 3); 16x16: 4 (lane >> 4) + r
#pragma unroll
            for (int g = 0; g < NR / 4; ++g) bv[g] = *reinterpret_cast<const float4 *>(bias + cb + 8 * g + 4 * fhj);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned pk[NR / 4][2];
                unsigned pkl[SPLIT ? NR / 4 : 1][2];
#pragma unroll
                for (int g = 0; g < NR / 4; ++g) {
                    float v0 = acc[i][j][4 * g + 0] + bv[g].x, v1 = acc[i][j][4 * g + 1] + bv[g].y;
                    float v2 = acc[i][j][4 * g + 2] + bv[g].z, v3 = acc[i][j][4 * g + 3] + bv[g].w;
                    if (a.leaky) {
                        v0 = v0 < 0.f ? v0 * 0.1f : v0; v1 = v1 < 0.f ? v1 * 0.1f : v1;
                        v2 = v2 < 0.f ? v2 * 0.1f : v2; v3 = v3 < 0.f ? v3 * 0.1f : v3;
                    }
                    const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                    pk[g][0] = __builtin_bit_cast(unsigned, h01);
                    pk[g][1] = __builtin_bit_cast(unsigned, h23);
                    if constexpr (SPLIT) {       // the lo halves of the same four values
                        const half2_t l01 = {(_Float16)(v0 - (float)h01[0]), (_Float16)(v1 - (float)h01[1])};
                        const half2_t l23 = {(_Float16)(v2 - (float)h23[0]), (_Float16)(v3 - (float)h23[1])};
                        pkl[g][0] = __builtin_bit_cast(unsigned, l01);
                        pkl[g][1] = __builtin_bit_cast(unsigned, l23);
                    }
                }
                if constexpr (TS == 32) {        // pair the lane halves' groups: 8 consecutive channels = 16 bytes per lane
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                        const uint2v s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                        const uint2v s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                        const int c0 = cb + 8 * (2 * pr + fhe);
                        if (qok[i] && c0 < a.n_store) {
                            *reinterpret_cast<uint4 *>(obase + (orow[i] + (unsigned)(c0 * 2))) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            if constexpr (SPLIT) *reinterpret_cast<uint4 *>(obase + (orow[i] + (unsigned)((c0 + 2 * a.split_n) * 2))) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                        if constexpr (SPLIT) {
                            const uint2v t0 = __builtin_amdgcn_permlane32_swap(pkl[2 * pr][0], pkl[2 * pr + 1][0], false, false);
                            const uint2v t1 = __builtin_amdgcn_permlane32_swap(pkl[2 * pr][1], pkl[2 * pr + 1][1], false, false);
                            if (qok[i] && c0 < a.n_store)
                                *reinterpret_cast<uint4 *>(obase + (orow[i] + (unsigned)((c0 + a.split_n) * 2))) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                        }
                    }
                } else {
                    const int c0 = cb + 4 * fhe;
                    if (qok[i] && c0 < a.n_store) {
                        *reinterpret_cast<uint2 *>(obase + (orow[i] + (unsigned)(c0 * 2))) = make_uint2(pk[0][0], pk[0][1]);
                        if constexpr (SPLIT) {
                            *reinterpret_cast<uint2 *>(obase + (orow[i] + (unsigned)((c0 + 2 * a.split_n) * 2))) = make_uint2(pk[0][0], pk[0][1]);
                            *reinterpret_cast<uint2 *>(obase + (orow[i] + (unsigned)((c0 + a.split_n) * 2))) = make_uint2(pkl[0][0], pkl[0][1]);
                        }
                    }
                }
            }
        }
        if (ti == 0) Y2_STAMP(4);
    }
#ifdef Y2_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Y2_STAMP(5);
    if (a.stamp && tid == 0 && blockIdx.x < kStampWGs) y2_stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}


// ---- register-resident weights: the 64 -> 128 channel 3x3 layers at 104 x 104 (layers 4 and 6) ---------------------------------
// These two layers (K = 576, N = 128) ran at 0.69-0.78 PF against 1.0 for the 26 x 26 layers: with only 128 output channels a
// tap-step of the halo kernels is 1024 MFMA cycles per SIMD between two barriers (half of the 256-channel tiles'), and the weight
// tile - 147 KB per 256 pixels, re-staged for every pixel tile - is 70 % of what the LDS-DMA moves.  Here NOTHING is staged per tap:
//   * a workgroup is four wavefronts, one per SIMD; wavefront w owns output channels 32 w .. 32 w + 31 and keeps ITS slice of the
//     weights - all nine taps x 64 input channels = 36 fragments of v_mfma_f32_16x16x32_f16 - in 144 registers for the whole kernel;
//   * a tile is TWO IMAGE ROWS (2 x 104 = 208 pixels = 13 blocks of 16 MFMA rows), staged once with its halo rows (dense in pixel
//     index, XOR swizzle and zero-row masking as in k_conv_f16_halo) into one of two LDS buffers by LDS-DMA while the previous tile
//     is multiplied: ONE barrier per tile, 468 MFMAs per wavefront between barriers;
//   * MFMA rows are ordered m = 4 w + 2 dy + dx over 2x2 pool windows w (a block = 4 windows), so that
//       MODE 1 (layer 6, whose only consumer is the pool): D rows = pixels, the four members of a window are the four registers of
//              one lane - in-lane max, bias + leaky (monotonic), the pooled row leaves through a small LDS tile in 16-byte stores;
//       MODE 0 (plain store) / MODE 2: operands swapped (D rows = channels): a lane holds 4 consecutive channels of a pixel;
//       MODE 2 (layer 4 + layer 5): the 128-channel result of the 3x3 is consumed ONLY by the 1x1 after it (128 -> 64), so it is
//              never written to memory: leaky'd fp16 [208 pixels][128 channels] into the LDS buffer the tile was read from, then
//              52 more MFMAs per wavefront (its 16 of the 64 output channels, weights again in registers) and 8-byte stores of layer
//              5's tensor.  Same bits as the two-kernel route's fp16 intermediate.
// K order tap-major like k_conv_f16; operands fp16, accumulate fp32.  wh: [N][9][64] halves; wh2 (MODE 2): [64][1][128].
#ifndef Y2_RW_ABL
#define Y2_RW_ABL 0         // diagnostic builds of k_conv_f16_rw: 2 = no fragment reads, 4 = no epilogue, 8 = no input staging
#endif
// LDS slot swizzle of the input tile (the halo kernels' key).  (A key that only moves slot bits 1-2 - conflict-free by brute force for
// the 16x16x32 fragment reads, where lanes of two k-groups share a ds_read_b128 lane group - measured 3 % SLOWER in the same box: the
// reads are not what this kernel waits for.)
__device__ __forceinline__ int rw_swz(int row) { return (row >> 1) & 7; }

template <int NBLK, int MODE>
__global__ __launch_bounds__(256) void k_conv_f16_rw(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh, const float *__restrict__ bias,
                                                      _Float16 *__restrict__ out, const _Float16 *__restrict__ wh2, const float *__restrict__ bias2,
                                                      const ConvF16Args a, const int lt_rows, const int n_tile_total)
{
    constexpr int ROWH = 64, BM = NBLK * 16, PD = 6;           // halves per LDS row; pixels per tile; A fragments in flight
    constexpr int kCtP = 136, kTP = 136;                       // halves per row of the pooled tile / of the intermediate tile (MODE 2)
    typedef float acc_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    _Float16 *As = smem_h;                                     // [2][lt_rows][64]
    _Float16 *Zs = smem_h + (size_t)2 * lt_rows * ROWH;        // [NBLK * 8][64] zeros: what a lane whose tap leaves the image reads (block k at + 8 k rows)
    _Float16 *Cts = Zs + (size_t)NBLK * 8 * ROWH;              // MODE 1: [2][BM / 4][kCtP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HH = a.H >> 1;                                   // row pairs per image
    // this workgroup's tiles (XCD-aware: XCD k owns a contiguous range, neighbouring row pairs share halo rows in its L2)
    const int xcd = (int)blockIdx.x & 7, S = (int)gridDim.x >> 3;
    const int t_first = (int)((long)n_tile_total * xcd / 8) + ((int)blockIdx.x >> 3);
    const int t_end = (int)((long)n_tile_total * (xcd + 1) / 8);
    const int my_n = t_first < t_end ? (t_end - t_first + S - 1) / S : 0;
    if (my_n == 0) return;

    for (int i = tid; i < NBLK * 8 * 8; i += 256) *reinterpret_cast<int4 *>(Zs + (size_t)i * 8) = make_int4(0, 0, 0, 0);
    const int lrow = lane >> 3, lslot = lane & 7;
    const int n_stage = (2 * a.W + 2 * (a.W + 1) + 7) / 8;     // 8-row groups that hold real pixels
    // Staging sources, once per kernel: LDS row R of a tile is pixel p = R - (W + 1) relative to the tile's first pixel (b, y0, 0), i.e.
    // image row y0 + floor(p / W), column p mod W - the same for every tile, so a lane keeps the BYTE offset of each of its pieces
    // relative to item (b, y0, 0) (swizzled 16-byte chunk included) and a fill is one add per piece.  (Recomputing the pixel -> item
    // map per piece - two multiply-high divisions - was ~25 instructions x 14 pieces per tile in front of every tile's first MFMA.)
    // No clamping: the rows above image 0 / below the last image fall into the tensor's lead / tail items (kLead, kTail) and are only
    // ever read masked.
    constexpr int kMaxFill = (4 * 8 * NBLK + 2 + 7) / 8 / 4 + 1;
    int relb[kMaxFill];
#pragma unroll
    for (int it = 0; it < kMaxFill; ++it) {
        const int row = (wave + 4 * it) * 8 + lrow, pp = row - (a.W + 1) + 2 * a.W;      // >= 0
        const int ry = pp / a.W - 2, x = pp - (ry + 2) * a.W;
        relb[it] = (ry * a.Wp + x) * 128 + ((lslot ^ rw_swz(row)) * 16);
    }
    auto fill_a = [&](int ti) {                 // the halo tile of this workgroup's ti-th tile -> buffer ti & 1
        const int t = t_first + ti * S, b = t / HH, y0 = 2 * (t - b * HH);
        const int base = (kLead + b * a.PL + (y0 + 1) * a.Wp) * 128;      // byte offset of item (b, y0, 0); the tensor stays below 2 GiB (host check)
        const char *abase = reinterpret_cast<const char *>(act);
#pragma unroll
        for (int it = 0; it < kMaxFill; ++it) {
            const int g = wave + 4 * it;
            if (g < n_stage) lds_dma16(abase, (unsigned)(base + relb[it]), As + ((size_t)(ti & 1) * lt_rows + g * 8) * ROWH);
        }
    };

    // ---- resident weights: lane (i = lane & 15, kq = lane >> 4) holds k = 8 kq .. 8 kq + 7 of row / column i of every fragment
    const int mi = lane & 15, kq = lane >> 4;
    half8_t bfr[18][2];
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            bfr[s][cb] = *reinterpret_cast<const half8_t *>(wh + ((size_t)(32 * wave + 16 * cb + mi) * 9 + (s >> 1)) * 64 + 32 * (s & 1) + 8 * kq);
    half8_t b2fr[MODE == 2 ? 4 : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) b2fr[s2] = *reinterpret_cast<const half8_t *>(wh2 + (size_t)(16 * wave + mi) * 128 + 32 * s2 + 8 * kq);
    }
    // this lane's pixel inside a block: MFMA row / column mi = 4 (window in block) + 2 dy + dx
    const int dy = (mi >> 1) & 1, dx = mi & 1, pwl = mi >> 2;
    const int lo0 = (a.W + 1) + dy * a.W + 2 * pwl + dx;      // LDS row of the centre tap, block 0 (block k: + 8 k)

    __syncthreads();                             // zero rows written
    fill_a(0);

    auto store_pooled_tile = [&](int ti) {       // MODE 1: the pooled row of tile ti from its LDS tile, 16 bytes per piece
        const int t = t_first + ti * S, b = t / HH, oy = t - b * HH;
        const _Float16 *Ct = Cts + (size_t)(ti & 1) * (BM / 4) * kCtP;
#pragma unroll
        for (int it = 0; it < (BM / 4 * 16 + 255) / 256; ++it) {
            const int piece = tid + it * 256, px = piece >> 4, ck = piece & 15;
            if (piece < BM / 4 * 16 && ck * 8 < a.n_store)
                *reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp + px) * a.Cp_out + a.out_ch_off + ck * 8) =
                    *reinterpret_cast<const half8_t *>(Ct + (size_t)px * kCtP + ck * 8);
        }
    };

    for (int ti = 0; ti < my_n; ++ti) {
        const int t = t_first + ti * S, b = t / HH, y0 = 2 * (t - b * HH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tile ti is staged; everybody is done with buffer (ti + 1) & 1 and, MODE 1, has written Ct(ti - 1)
        if constexpr (MODE == 1) { if (ti > 0) store_pooled_tile(ti - 1); }
#if !(Y2_RW_ABL & 8)
        if (ti + 1 < my_n) fill_a(ti + 1);
#endif
        const _Float16 *At = As + (size_t)(ti & 1) * lt_rows * ROWH;

        acc_t acc[NBLK][2];
#pragma unroll
        for (int k = 0; k < NBLK; ++k)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[k][cb] = acc_t{0.f, 0.f, 0.f, 0.f};

        // A fragment `idx` = (tap, block, channel half): issued PD fragments ahead of the MFMAs that consume it
        constexpr int NF = 9 * NBLK * 2;
        const bool top = y0 == 0 && dy == 0, bot = y0 == a.H - 2 && dy == 1;
        // LDS address of A fragment (tap, block, channel half).  Per tap a lane needs TWO registers: its row for block 0 is
        // r0 = lo0 + tap offset, block k is 8 k rows (1024 bytes: the instruction's immediate offset) further, and the swizzle key
        // ((row >> 1) & 7) of row r0 + 8 k is key(r0) ^ 4 (k & 1) - flipping slot bit 2, which is the same as taking the OTHER channel
        // half's slot: E[h] = address of half h in block 0; block k, half h reads E[h ^ (k & 1)] + 1024 k.  A lane whose tap leaves
        // the image (top / bottom image row: every block; left / right image column: block 0 / NBLK - 1 only) reads the zero region.
        const unsigned at_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)At;
        const unsigned z_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)Zs;
        auto tap_base = [&](int tap, int h) -> unsigned {
            const int ty = tap / 3, tx = tap - ty * 3;
            const int r0 = lo0 + (ty - 1) * a.W + (tx - 1);
            const unsigned e = at_lds + (unsigned)(r0 * (ROWH * 2) + (((4 * h + kq) ^ rw_swz(r0)) * 16));
            return ((top && ty == 0) || (bot && ty == 2)) ? z_lds : e;
        };
        auto frag_base = [&](int idx) -> unsigned {
            const int tap = idx / (NBLK * 2), rem = idx - tap * (NBLK * 2), blk = rem >> 1, hf = rem & 1, tx = tap % 3;
            unsigned e = tap_base(tap, hf ^ (blk & 1));
            if (blk == 0 && tx == 0) e = (pwl == 0 && dx == 0) ? z_lds : e;
            if (blk == NBLK - 1 && tx == 2) e = (pwl == 3 && dx == 1) ? z_lds : e;
            return e;
        };
        // Fragments travel in GROUPS of PD = 6 (three blocks x two channel halves) through two register sets: group g + 1 is read from
        // LDS during the FIRST half of group g's MFMAs, so that its youngest read is >= 8 MFMAs (128 cycles) old when group g + 1
        // starts.  The reads and their wait are INLINE ASM: with compiler-issued ds_read_b128 hipcc waited lgkmcnt(0) at a group's
        // first use ALSO for the reads it had just issued for the next group (seen in the ISA: one full drain per two groups), and
        // with ONE wavefront per SIMD nothing covers that latency.  After the wait a sched_barrier keeps every MFMA behind it
        // (cdna_hip_programming.md 5.4 rule 18); inside a group the compiler is free to interleave address arithmetic, reads and MFMAs.
        static_assert(NF % PD == 0 && PD == 6, "fragment groups");
        half8_t fr[2][PD];
#if (Y2_RW_ABL & 2)
        auto issue1 = [&](int g, int i) { asm volatile("v_mov_b32 %0, %1" : "=v"(fr[g & 1][i][0]) : "v"(frag_base(g * PD + i)) : "memory"); };   // diagnostic: no LDS read
#else
        auto issue1 = [&](int g, int i) {
            const int idx = g * PD + i, blk = (idx % (NBLK * 2)) >> 1;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[g & 1][i]) : "v"(frag_base(idx)), "i"(blk * 1024) : "memory");
        };
#endif
#pragma unroll
        for (int i = 0; i < PD; ++i) issue1(0, i);
#pragma clang loop unroll(full)
        for (int g = 0; g < NF / PD; ++g) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // group g's fragments (issued during group g - 1) have landed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                if (g + 1 < NF / PD && k < PD / 2) { issue1(g + 1, 2 * k); issue1(g + 1, 2 * k + 1); }
                const int i = (k % 3) * 2 + k / 3;                     // blocks first, channel halves second: six MFMAs between two on one accumulator
                const int idx = g * PD + i, tap = idx / (NBLK * 2), rem = idx - tap * (NBLK * 2), blk = rem >> 1, s = 2 * tap + (rem & 1);
                const half8_t af = fr[g & 1][i];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if constexpr (MODE == 1) acc[blk][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr[s][cb], acc[blk][cb], 0, 0, 0);
                    else acc[blk][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[s][cb], af, acc[blk][cb], 0, 0, 0);
                }
                if (k < PD / 2) __builtin_amdgcn_sched_barrier(0);     // (left alone, hipcc sinks the next group's reads to the end of this group)
            }
            __builtin_amdgcn_sched_barrier(0);
        }

#if (Y2_RW_ABL & 4)
        {   // diagnostic: no epilogue - one value per lane keeps the accumulators alive
            float keep = 0.f;
#pragma unroll
            for (int k = 0; k < NBLK; ++k) keep += acc[k][0][0] + acc[k][1][3];
            if (keep == 123.456f) out[tid] = (_Float16)keep;
            continue;
        }
#endif
        if constexpr (MODE == 1) {
            // D row = 4 kq + r = window kq of the block, member r: the pool is a max over the four registers
            _Float16 *Ct = Cts + (size_t)(ti & 1) * (BM / 4) * kCtP;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int ch = 32 * wave + 16 * cb + mi;
                const float bv = bias[ch];
#pragma unroll
                for (int k = 0; k < NBLK; ++k) {
                    float v = fmaxf(fmaxf(acc[k][cb][0], acc[k][cb][1]), fmaxf(acc[k][cb][2], acc[k][cb][3])) + bv;
                    if (a.leaky) v = fmaxf(v, v * 0.1f);
                    Ct[(size_t)(4 * k + kq) * kCtP + ch] = (_Float16)v;
                }
            }
        } else {
            // D row = channel 4 kq + r of the 16-channel block, column = pixel mi of the block: 4 consecutive channels per lane
            const int x0 = 2 * pwl + dx, y = y0 + dy;
            if constexpr (MODE == 0) {
                _Float16 *orow = out + ((size_t)kLead + (size_t)b * a.PL + (size_t)(y + 1) * a.Wp + x0) * a.Cp_out + a.out_ch_off;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int c0 = 32 * wave + 16 * cb + 4 * kq;
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + c0);
#pragma unroll
                    for (int k = 0; k < NBLK; ++k) {
                        float v0 = acc[k][cb][0] + bv.x, v1 = acc[k][cb][1] + bv.y, v2 = acc[k][cb][2] + bv.z, v3 = acc[k][cb][3] + bv.w;
                        if (a.leaky) { v0 = fmaxf(v0, v0 * 0.1f); v1 = fmaxf(v1, v1 * 0.1f); v2 = fmaxf(v2, v2 * 0.1f); v3 = fmaxf(v3, v3 * 0.1f); }   // == (v < 0 ? 0.1 v : v), one instruction less
                        const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                        if (c0 < a.n_store)
                            *reinterpret_cast<uint2 *>(orow + (size_t)(8 * k) * a.Cp_out + c0) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                    }
                }
            } else {
                // MODE 2: the 3x3's leaky'd fp16 result goes into the LDS buffer this tile was read from ([BM][kTP] halves, row = 16 block + mi)
                __syncthreads();                 // every wavefront has finished reading the input tile
                _Float16 *T = As + (size_t)(ti & 1) * lt_rows * ROWH;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int c0 = 32 * wave + 16 * cb + 4 * kq;
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + c0);
#pragma unroll
                    for (int k = 0; k < NBLK; ++k) {
                        float v0 = acc[k][cb][0] + bv.x, v1 = acc[k][cb][1] + bv.y, v2 = acc[k][cb][2] + bv.z, v3 = acc[k][cb][3] + bv.w;
                        if (a.leaky) { v0 = fmaxf(v0, v0 * 0.1f); v1 = fmaxf(v1, v1 * 0.1f); v2 = fmaxf(v2, v2 * 0.1f); v3 = fmaxf(v3, v3 * 0.1f); }   // == (v < 0 ? 0.1 v : v), one instruction less
                        const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                        *reinterpret_cast<uint2 *>(T + (size_t)(16 * k + mi) * kTP + c0) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                    }
                }
                __syncthreads();
                // the 1x1: out2[pixel][16 wave + 4 kq + r] = sum over 128 channels; operands swapped again (D rows = output channels)
                acc_t acc2[NBLK];
#pragma unroll
                for (int k = 0; k < NBLK; ++k) acc2[k] = acc_t{0.f, 0.f, 0.f, 0.f};
                // 4 k-slices x NBLK blocks = 52 fragments of the intermediate tile, in groups of 4 through two register sets, read by
                // inline asm like the input tile's (compiler-issued reads wait lgkmcnt(0) each: ~100 cycles x 52 with one wavefront per SIMD)
                {
                    constexpr int PD2 = 4, NF2 = 4 * NBLK;
                    static_assert(NF2 % PD2 == 0, "groups");
                    const unsigned t_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)T + (unsigned)(mi * (kTP * 2) + 16 * kq);
                    half8_t tf[2][PD2];
                    auto issue2 = [&](int g, int i) {
                        const int idx = g * PD2 + i, s2 = idx / NBLK, k = idx - s2 * NBLK;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tf[g & 1][i]) : "v"(t_lds), "i"(16 * k * (kTP * 2) + 64 * s2) : "memory");
                    };
#pragma unroll
                    for (int i = 0; i < PD2; ++i) issue2(0, i);
#pragma clang loop unroll(full)
                    for (int g = 0; g < NF2 / PD2; ++g) {
                        if (g + 1 < NF2 / PD2) {
#pragma unroll
                            for (int i = 0; i < PD2; ++i) issue2(g + 1, i);
                            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD2) : "memory");
                        } else
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < PD2; ++i) {
                            const int idx = g * PD2 + i, s2 = idx / NBLK, k = idx - s2 * NBLK;
                            acc2[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2fr[s2], tf[g & 1][i], acc2[k], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                _Float16 *orow = out + ((size_t)kLead + (size_t)b * a.PL + (size_t)(y + 1) * a.Wp + x0) * a.Cp_out + a.out_ch_off;
                const int c0 = 16 * wave + 4 * kq;
                const float4 bv = *reinterpret_cast<const float4 *>(bias2 + c0);
#pragma unroll
                for (int k = 0; k < NBLK; ++k) {
                    float v0 = acc2[k][0] + bv.x, v1 = acc2[k][1] + bv.y, v2 = acc2[k][2] + bv.z, v3 = acc2[k][3] + bv.w;
                    if (a.leaky) { v0 = fmaxf(v0, v0 * 0.1f); v1 = fmaxf(v1, v1 * 0.1f); v2 = fmaxf(v2, v2 * 0.1f); v3 = fmaxf(v3, v3 * 0.1f); }   // == (v < 0 ? 0.1 v : v), one instruction less
                    const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                    if (c0 < a.n_store)
                        *reinterpret_cast<uint2 *>(orow + (size_t)(8 * k) * a.Cp_out + c0) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                }
                // (the next iteration's barrier orders these reads of T before fill_a(ti + 2) overwrites the buffer)
            }
        }
    }
    if constexpr (MODE == 1) {
        __syncthreads();
        store_pooled_tile(my_n - 1);
    }
}

// ---- k_conv_f16_rwb: k_conv_f16_rw with its epilogue INSIDE the MFMA stream --------------------------------------------------
// k_conv_f16_rw multiplies a tile tap by tap with all 26 accumulators live and runs the epilogue afterwards, with one wavefront per
// SIMD and nothing to cover it: tools/rw_abl.sh prices that at 0.033 ms of layer 6's 0.176 (pool) and 0.113 ms of layers 4 + 5's
// 0.267 (intermediate tile, second GEMM, stores) per 128 frames, against 0.13 ms of MFMA issue.  Here the order is BLOCK-outer:
// the weights are in registers anyway, so one block of 16 pixels is finished (18 fragments x 2 channel blocks = 36 MFMAs) before
// the next begins, only two pairs of accumulators exist, and a finished block's epilogue is cut into pieces of <= 3 VALU
// instructions that are placed, one per MFMA, between the NEXT block's MFMAs (a 16x16x32 MFMA holds the issue port for half of its
// 16 cycles).  Every slot ends in a sched_barrier, so the order written here is the order issued.
//   * the input tile is staged WITH the layout's zero columns and rows (layout.hpp: out-of-image taps read stored zeros): four image
//     rows of 108 LDS rows each - x = -1 (the zero column before the row), x = 0 .. 103, x = 104 (the zero column after it), two unused -
//     so there is no zero region and no mask.  The pitch is 108 because of the swizzle: at 105 (the tensor's own pitch) no XOR key
//     over the row bits makes the 16x16x32 fragment reads conflict-free (best 2-way; (row >> 1) & 7 is 3-way and the kernel ran at
//     the speed of its LDS reads); at 108 the key {bit 1 -> slot bit 2, bit 2 -> slot bit 1} is conflict-free for every tap, both
//     channel halves and all four lane groups of ds_read_b128 (brute force over the bank model of MI355X_MICROARCH.md);
//   * MODE 1 (layer 6 + pool): block b's pooled values go to the LDS tile during block b + 1; the pooled row of tile t - 1 is
//     stored during block 0 of tile t;
//   * MODE 2 (layer 4 + layer 5): the leaky'd fp16 result T [208][128] has its own LDS region (XOR-swizzled 256-byte rows) and the
//     tile is worked in two halves (blocks 0-6 | 7-12) with a barrier between: while one half is multiplied, the 1x1 of the OTHER
//     half - written before the last barrier - runs from T (4 fragment reads, 4 MFMAs, bias + leaky + 8-byte stores per block),
//     also between the 3x3's MFMAs.  Two barriers per tile; T-writes of a half's last block and one 1x1 block are all that is
//     exposed.
// Same arithmetic and the same bits as k_conv_f16_rw (bias after the sum, leaky as max(v, 0.1 v), fp16 intermediate).
// Host: W = 104, 64-channel items in, leaky layers, every channel stored (build_f16_plan).
#ifndef Y2_RWB_ABL
#define Y2_RWB_ABL 0        // diagnostic builds of k_conv_f16_rwb (results wrong, only time matters): 1 = no per-group LDS waits, 2 = no staging,
#endif                      // 4 = no epilogue work between the MFMAs, 8 = no fragment reads; MODE 1: 16 = epilogue arithmetic without its LDS writes, 32 = the writes without the arithmetic, 64 = no pooled-row store (bit 4 also strips k_conv_f16_rwc's epilogue work)
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_f16_rwb(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh, const float *__restrict__ bias,
                                                       _Float16 *__restrict__ out, const _Float16 *__restrict__ wh2, const float *__restrict__ bias2,
                                                       const ConvF16Args a, const int n_tile_total)
{
    static_assert(MODE == 1 || MODE == 2, "pool or fused 1x1");
    constexpr int NBLK = 13, WP = 8 * NBLK + 1, LP = 108, ROWH = 64, PD = 6, NF = NBLK * 18, NG = NF / PD, H0 = 7;
    constexpr int LT = 4 * LP, kPieces = LT / 8, kFill = (kPieces + 3) / 4;   // 432 LDS rows per tile = 54 staged 8-row groups, 14 per wavefront
    constexpr int kCtP = 136;                                    // halves per row of the pooled tile (MODE 1)
    constexpr unsigned kBufBytes = LT * ROWH * 2, kCtBytes = (NBLK * 4) * kCtP * 2;
    typedef float acc_t __attribute__((ext_vector_type(4)));
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    _Float16 *As = smem_h;                                       // [2][LT][64]
    _Float16 *Xs = smem_h + (size_t)2 * LT * ROWH;               // MODE 1: pooled tiles [2][52][kCtP]; MODE 2: T [208][128], swizzled

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HH = a.H >> 1;
    const int xcd = (int)blockIdx.x & 7, S = (int)gridDim.x >> 3;
    const int t_first = (int)((long)n_tile_total * xcd / 8) + ((int)blockIdx.x >> 3);
    const int t_end = (int)((long)n_tile_total * (xcd + 1) / 8);
    const int my_n = t_first < t_end ? (t_end - t_first + S - 1) / S : 0;
    if (my_n == 0) return;

    // ---- staging: LDS row R = image row y0 - 1 + R / 108, x = R % 108 - 1; 16-byte chunks XOR-swizzled by rkey(R)
    auto rkey = [](int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1); };
    const int lrow = lane >> 3, lslot = lane & 7;
    // (piece wave + 4 it; the three wavefronts whose fourteenth piece would lie past the tile stage piece 52 again - the same bytes -
    //  so that no fill is conditional: a branch would end the basic block and with it the slot order the sched_barriers pin)
    int relb[kFill];                             // source byte offset of this lane's chunk relative to item (b, y0 - 1, 0)
#pragma unroll
    for (int it = 0; it < kFill; ++it) {
        const int row = min(wave + 4 * it, kPieces - 1) * 8 + lrow, ir = row / LP, c = row - ir * LP;
        relb[it] = (ir * WP + c - 1) * 128 + ((lslot ^ rkey(row)) * 16);
    }
    const char *abase = reinterpret_cast<const char *>(act);
    auto tile_item = [&](int ti) -> int {        // item index of (b, y0, 0) of this workgroup's ti-th tile
        const int t = t_first + ti * S, b = t / HH, y0 = 2 * (t - b * HH);
        return kLead + b * a.PL + (y0 + 1) * WP;
    };
    int fill_base = 0;                           // byte offset of item (b, y0 - 1, 0) of the tile being staged
    auto fill_piece = [&](int ti, int it) {
        const int g = min(wave + 4 * it, kPieces - 1);
        lds_dma16(abase, (unsigned)(fill_base + relb[it]), As + ((size_t)(ti & 1) * LT + g * 8) * ROWH);
    };

    // ---- resident weights: lane (i = lane & 15, kq = lane >> 4) holds k = 8 kq .. 8 kq + 7 of row / column i of every fragment
    const int mi = lane & 15, kq = lane >> 4;
    half8_t bfr[18][2];
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            bfr[s][cb] = *reinterpret_cast<const half8_t *>(wh + ((size_t)(32 * wave + 16 * cb + mi) * 9 + (s >> 1)) * 64 + 32 * (s & 1) + 8 * kq);
    half8_t b2fr[MODE == 2 ? 4 : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) b2fr[s2] = *reinterpret_cast<const half8_t *>(wh2 + (size_t)(16 * wave + mi) * 128 + 32 * s2 + 8 * kq);
    }
    // this lane's pixel inside a block: MFMA row / column mi = 4 (window in block) + 2 dy + dx
    const int dy = (mi >> 1) & 1, dx = mi & 1, pw = mi >> 2;
    const int lo0 = dy * LP + 2 * pw + dx;        // LDS row of tap (0, 0), block 0 (block k: + 8 k rows = + 1024 k bytes, the instruction's immediate)
    const unsigned as_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)As;
    const unsigned xs_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)Xs;
    // A fragment (tap, channel half h, block k): row r0 + 8 k, whose key is key(r0) (the key reads row bits 1-2 only): byte address
    // e[tap] ^ 64 h, + 1024 k
    unsigned erel[9], ecur[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int r0 = lo0 + (tap / 3) * LP + (tap % 3);
        erel[tap] = as_lds + (unsigned)(r0 * 128 + ((kq ^ rkey(r0)) * 16));
    }

    // ---- epilogue constants
    float bv1[2];                                 // MODE 1: bias of this lane's channel in each channel block
    acc_t bv2[2], b2v;                            // MODE 2: bias of this lane's 4 channels per channel block; of its 4 output channels of the 1x1
    unsigned ctw[2], twr[2], trb = 0, ct_cur[2] = {0, 0};
    unsigned o_lane = 0;                          // MODE 2: byte offset of this lane's pixel / channels relative to item (b, y0, 0) of the 1x1's tensor
    if constexpr (MODE == 1) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            bv1[cb] = bias[32 * wave + 16 * cb + mi];
            ctw[cb] = xs_lds + (unsigned)((kq * kCtP + 32 * wave + 16 * cb + mi) * 2);      // pooled pixel 4 k + kq, channel
        }
    } else {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * wave + 16 * cb + 4 * kq);
            bv2[cb] = acc_t{b4.x, b4.y, b4.z, b4.w};
            // T row 16 k + mi, channels 32 wave + 16 cb + 4 kq .. + 3: 16-byte slot (4 wave + 2 cb + (kq >> 1)) ^ mi, upper or lower 8 bytes
            twr[cb] = xs_lds + (unsigned)(mi * 256 + (((4 * wave + 2 * cb + (kq >> 1)) ^ mi) * 16) + (kq & 1) * 8);
        }
        const float4 b4 = *reinterpret_cast<const float4 *>(bias2 + 16 * wave + 4 * kq);
        b2v = acc_t{b4.x, b4.y, b4.z, b4.w};
        trb = xs_lds + (unsigned)(mi * 256 + ((kq ^ mi) * 16));       // fragment of k-slice s2: slot (4 s2 + kq) ^ mi = byte address ^ 64 s2
        o_lane = (unsigned)((((dy * WP + 2 * pw + dx) * a.Cp_out) + a.out_ch_off + 16 * wave + 4 * kq) * 2);
    }

    half8_t fr[2][PD];
    acc_t acc[2][2];
    // MODE 2 state carried between slots
    acc_t acc2[2];
    half8_t tf[2][4];
    float tv[2][4], tu[2][4], ov[4], ou[4];
    unsigned tpk[2][2], opk[2];
    // MODE 1 state
    float pm[2], pt[2];
    unsigned ph[2];
    half8_t pst[4];
    char *o_cur = nullptr, *o_prev = nullptr;    // MODE 2: item (b, y0, 0) of the 1x1's tensor for this / the previous tile; MODE 1: pooled row of the previous tile
#ifdef Y2_STAMPS
    // diagnostic builds: wavefront 0 of workgroup 0 records the shader clock at the start of every group of its fourth tile (y2_stamps[0 .. NG], tools/rwb_stamps.py)
    unsigned tstamp[NG + 1];
    unsigned long long tprev = 0;
#endif

    auto issue_a = [&](int g, int i) {
        const int F = g * PD + i, blk = F / 18, s = F % 18;
#if (Y2_RWB_ABL & 8)
        asm volatile("v_mov_b32 %0, %1" : "=v"(fr[g & 1][i][0]) : "v"(ecur[s >> 1]) : "memory");
        return;
#endif
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[g & 1][i]) : "v"(ecur[s >> 1] ^ (unsigned)(64 * (s & 1))), "i"(blk * 1024) : "memory");
    };
    auto leaky = [](float v) { return fmaxf(v, v * 0.1f); };      // == (v < 0 ? 0.1 v : v), one instruction less
    auto pack2 = [](float x, float y) -> unsigned { const half2_t h = {(_Float16)x, (_Float16)y}; return __builtin_bit_cast(unsigned, h); };

    // ---- the epilogue as MICRO-OPS (one VALU instruction or one memory access each).  A 16x16x32 MFMA holds the issue port for 8 of its
    // 16 cycles and a VALU instruction for 4: a slot carries its MFMA and at most TWO micro-ops; the first three slots of a group carry
    // the next group's six fragment reads instead.  So a block of 36 slots has 27 free slots = 54 micro-op positions k.
    // MODE 1, block pb: positions 0..5 / 6..11 (channel blocks 0 / 1): pool = max over the four registers (one window), bias, leaky,
    // convert; 18 / 19: the 2-byte LDS writes (early in the next group: an LDS access must be old at the wait that ends its group)
    auto drain1 = [&](int pb, int k) {
        const int set = pb & 1;
#if (Y2_RWB_ABL & 32)
        if (k < 12) { if (k % 6 == 0) ph[k / 6] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc[set][k / 6][0]); } else
#elif (Y2_RWB_ABL & 16)
        if (k >= 12) { if (k == 19) asm volatile("" ::"v"(ph[0]), "v"(ph[1])); } else
#endif
        if (k < 12) {
            const int cb = k / 6, j = k % 6;
            // (the two maxima as instructions: fmaxf() on values the compiler cannot prove canonical - MFMA results - first runs each
            //  operand through a v_max x, x, which made these two micro-ops five instructions; an MFMA on finite inputs gives no signalling NaN)
            if (j == 0) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(pm[cb]) : "v"(acc[set][cb][0]), "v"(acc[set][cb][1]), "v"(acc[set][cb][2]));
            else if (j == 1) asm("v_max_f32 %0, %1, %2" : "=v"(pm[cb]) : "v"(pm[cb]), "v"(acc[set][cb][3]));
            else if (j == 2) pm[cb] = pm[cb] + bv1[cb];
            else if (j == 3) pt[cb] = pm[cb] * 0.1f;
            else if (j == 4) pm[cb] = fmaxf(pm[cb], pt[cb]);         // leaky == (v < 0 ? 0.1 v : v)
            else ph[cb] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)pm[cb]);
        } else if (k == 18 || k == 19) {
            const int cb = k - 18;
            asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(ct_cur[cb]), "v"(ph[cb]), "i"(4 * pb * kCtP * 2) : "memory");
        }
    };
    // bias + leaky + fp16 of four accumulator values in 14 micro-ops: v[4] in, two packed registers out
    auto epi14 = [&](int j, const acc_t &v, const acc_t &bv, float (&t)[4], float (&u)[4], unsigned (&pk)[2]) {
        const int r = j < 7 ? j / 3 : 2 + (j - 7) / 3, ph3 = j < 7 ? j % 3 : (j - 7) % 3;
        if (j == 6) pk[0] = pack2(t[0], t[1]);
        else if (j == 13) pk[1] = pack2(t[2], t[3]);
        else if (ph3 == 0) t[r] = v[r] + bv[r];
        else if (ph3 == 1) u[r] = t[r] * 0.1f;
        else t[r] = fmaxf(t[r], u[r]);
    };
    // MODE 2, T-write of block pb: positions 4..17 / 20..33 the values of channel blocks 0 / 1, 18 / 36 their 8-byte LDS writes
    auto drain2 = [&](int pb, int k) {
        const int set = pb & 1;
        if (k >= 4 && k < 18) epi14(k - 4, acc[set][0], bv2[0], tv[0], tu[0], tpk[0]);
        else if (k >= 20 && k < 34) epi14(k - 20, acc[set][1], bv2[1], tv[1], tu[1], tpk[1]);
        else if (k == 18 || k == 36) {
            const int cb = k == 18 ? 0 : 1;
            const uint2v d = {tpk[cb][0], tpk[cb][1]};
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(twr[cb]), "v"(d), "i"(pb * 4096) : "memory");
        }
    };
    // MODE 2: the 1x1 of T block q (16 pixels x this wavefront's 16 output channels), register set z: fragment reads, MFMAs, epilogue
    auto g2_read = [&](int z, int q, int s2) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tf[z][s2]) : "v"(trb ^ (unsigned)(64 * s2)), "i"(q * 4096) : "memory");
    };
    auto g2_mfma = [&](int z, int s2) {
        const acc_t c0 = s2 == 0 ? acc_t{0.f, 0.f, 0.f, 0.f} : acc2[z];
        acc2[z] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2fr[s2], tf[z][s2], c0, 0, 0, 0);
    };
    auto g2_epi = [&](int z, int q, int j, char *obase) {      // j = 0..13: the values; 14: the 8-byte store
        if (j < 14) epi14(j, acc2[z], b2v, ov, ou, opk);
        else if (j == 14) *reinterpret_cast<uint2v *>(obase + (o_lane + (unsigned)(q * 8 * a.Cp_out * 2))) = uint2v{opk[0], opk[1]};
    };
    // micro-op k of main block blk (MODE 2): the T-write of block blk - 1, the 1x1 of the other half's block, and in the last block the reads
    // of the 1x1 that has no block of its own (T block 6)
    auto side2 = [&](int blk, int k) {
        if (blk != 0 && blk != H0) drain2(blk - 1, k);
        const bool has = blk < H0 ? blk < NBLK - H0 : true;
        const int q = blk < H0 ? H0 + blk : blk - H0;         // T block 7 + blk of the previous tile (first half), blk - 7 of this tile (second half)
        if (has) {
            if (k < 4) g2_read(0, q, k);
            if (k >= 38 && k < 53) g2_epi(0, q, k - 38, blk < H0 ? o_prev : o_cur);
        }
        if (blk == NBLK - 1) {
            if (k == 19) g2_read(1, H0 - 1, 0);
            if (k == 34) g2_read(1, H0 - 1, 1);
            if (k == 35) g2_read(1, H0 - 1, 2);
            if (k == 37) g2_read(1, H0 - 1, 3);
        }
    };
    auto side1 = [&](int blk, int k, int ti) {
        if (blk > 0) drain1(blk - 1, k);
#if (Y2_RWB_ABL & 64)
        else if (k < 0) {
#else
        else {
#endif
            // the pooled row of tile ti - 1: 52 pixels x 16 pieces of 16 bytes over 256 threads (the fourth round is wavefront 0's;
            // the others repeat piece 831 - same bytes, no branch): LDS reads at positions 0..3, stores at 18..21
            const unsigned cprev = xs_lds + ((ti & 1) ? 0u : kCtBytes);
            if (k < 4) {
                const int piece = min(tid + k * 256, NBLK * 4 * 16 - 1), px = piece >> 4, ck = piece & 15;
                asm volatile("ds_read_b128 %0, %1" : "=v"(pst[k]) : "v"(cprev + (unsigned)((px * kCtP + ck * 8) * 2)) : "memory");
            } else if (k >= 18 && k < 22) {
                const int piece = min(tid + (k - 18) * 256, NBLK * 4 * 16 - 1), px = piece >> 4, ck = piece & 15;
                *reinterpret_cast<half8_t *>(o_prev + (size_t)((px * a.Cp_out + ck * 8) * 2)) = pst[k - 18];
            }
        }
    };

    {   // first tile
        fill_base = (tile_item(0) - WP) * 128;
#pragma unroll
        for (int it = 0; it < kFill; ++it) fill_piece(0, it);
    }

    for (int ti = 0; ti < my_n; ++ti) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tile ti is staged; everybody has left tile ti - 1 (its input buffer, its T / pooled tile are complete)
        // No side work is conditional.  After the last tile the "next" tile staged is this one again (into the idle buffer); before
        // the first, the "previous" tile's output goes where THIS tile's will: garbage (from LDS nobody has written) that the same
        // lanes overwrite with the real values later - one thread's stores to one address keep their order.
        {
            const int item = tile_item(ti);
            if constexpr (MODE == 2) {
                char *oc = reinterpret_cast<char *>(out) + (size_t)item * a.Cp_out * 2;
                o_prev = ti > 0 ? o_cur : oc;
                o_cur = oc;
            } else if (ti == 0) {
                const int t = t_first, b = t / HH, oy = t - b * HH;
                o_prev = reinterpret_cast<char *>(out) + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp) * a.Cp_out * 2 + (size_t)a.out_ch_off * 2;
            }
            fill_base = (tile_item(ti + 1 < my_n ? ti + 1 : ti) - WP) * 128;
        }
        const unsigned boff = (ti & 1) ? kBufBytes : 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) ecur[tap] = erel[tap] + boff;
        if constexpr (MODE == 1) {
            ct_cur[0] = ctw[0] + ((ti & 1) ? kCtBytes : 0u);
            ct_cur[1] = ctw[1] + ((ti & 1) ? kCtBytes : 0u);
        }
#pragma unroll
        for (int i = 0; i < PD; ++i) issue_a(0, i);

#pragma clang loop unroll(full)
        for (int g = 0; g < NG; ++g) {
            if (MODE == 2 && g == H0 * 3) {      // the first half's last block, then the barrier between the halves (LDS only: the next tile's fill stays in flight)
#pragma unroll
                for (int k = 0; k < 54; ++k) drain2(H0 - 1, k);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
#if !(Y2_RWB_ABL & 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // group g's fragments (issued during group g - 1) and every other LDS access of that group have landed
#endif
#ifdef Y2_STAMPS
            if (g > 0) tstamp[g - 1] = (unsigned)tprev;
            tprev = __builtin_amdgcn_s_memtime();                      // (read one group later, behind that group's wait: the counter's return is never waited for on its own)
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int F = g * PD + i, blk = F / 18, s = F % 18, m = 2 * s + cb, qs = 2 * i + cb;
                    const acc_t c0 = s == 0 ? acc_t{0.f, 0.f, 0.f, 0.f} : acc[blk & 1][cb];
                    if constexpr (MODE == 1) acc[blk & 1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[g & 1][i], bfr[s][cb], c0, 0, 0, 0);
                    else acc[blk & 1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[s][cb], fr[g & 1][i], c0, 0, 0, 0);
                    if (qs < PD / 2) {                                  // the next group's fragments: two reads in each of the first three slots
                        if (g + 1 < NG) { issue_a(g + 1, 2 * qs); issue_a(g + 1, 2 * qs + 1); }
                    } else {
                        const int k0 = 2 * ((m / 12) * 9 + qs - PD / 2);        // this slot's two micro-op positions
#if !(Y2_RWB_ABL & 4)
                        if constexpr (MODE == 1) { side1(blk, k0, ti); side1(blk, k0 + 1, ti); }   // (one micro-op per slot instead of two, or one fragment read per slot over six slots: -1.5 .. -3 % on layer 6, within what boxes differ by)
                        else { side2(blk, k0); side2(blk, k0 + 1); }
#endif
                    }
                    if constexpr (MODE == 2) {
#if !(Y2_RWB_ABL & 4)
                        // the 1x1's MFMAs ride behind main MFMAs of the block's second group (its fragments were read in the first)
                        if ((blk < H0 ? blk < NBLK - H0 : true) && m >= 14 && m <= 23 && (m - 14) % 3 == 0) g2_mfma(0, (m - 14) / 3);
                        if (blk == NBLK - 1 && (m == 26 || m == 30 || m == 34)) g2_mfma(1, (m - 26) / 4);
#endif
                    }
#if !(Y2_RWB_ABL & 2)
                    if (blk < 4 && m >= 20 && m < 24 && 4 * blk + (m - 20) < kFill) fill_piece(ti + 1, 4 * blk + (m - 20));
#endif
#if (Y2_RWB_ABL & 4)
                    if (m == 35) asm volatile("" ::"v"(acc[blk & 1][0]), "v"(acc[blk & 1][1]));      // (keeps the MFMAs alive without their consumers)
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#ifdef Y2_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tstamp[NG - 1] = (unsigned)tprev;
        tstamp[NG] = (unsigned)__builtin_amdgcn_s_memtime();
        if (a.stamp && ti == 3 && blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int g = 0; g <= NG; ++g) y2_stamps[g] = tstamp[g];
        }
#endif
        // the tile's last block (nothing left to hide it behind)
        if constexpr (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 20; ++k) drain1(NBLK - 1, k);
            {   // the pooled row of THIS tile is stored during the next tile (or after the loop): remember where it goes
                const int t = t_first + ti * S, b = t / HH, oy = t - b * HH;
                o_prev = reinterpret_cast<char *>(out) + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp) * a.Cp_out * 2 + (size_t)a.out_ch_off * 2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 54; ++k) drain2(NBLK - 1, k);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fourth fragment of T block 6
            g2_mfma(1, 3);
#pragma unroll
            for (int j = 0; j < 15; ++j) g2_epi(1, H0 - 1, j, o_cur);
        }
    }

    // ---- after the last tile: its pooled row (MODE 1) / the 1x1 of its second half (MODE 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (MODE == 1) {
        const _Float16 *Ct = Xs + (size_t)((my_n - 1) & 1) * (NBLK * 4) * kCtP;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int piece = tid + it * 256, px = piece >> 4, ck = piece & 15;
            if (piece < NBLK * 4 * 16)
                *reinterpret_cast<half8_t *>(o_prev + (size_t)((px * a.Cp_out + ck * 8) * 2)) = *reinterpret_cast<const half8_t *>(Ct + (size_t)px * kCtP + ck * 8);
        }
    } else {
#pragma unroll
        for (int q = H0; q < NBLK; ++q) {
            acc_t c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const half8_t f = *reinterpret_cast<const half8_t *>(Xs + (size_t)(16 * q + mi) * 128 + (((4 * s2 + kq) ^ mi) * 8));
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2fr[s2], f, c2, 0, 0, 0);
            }
            const uint2v d = {pack2(leaky(c2[0] + b2v[0]), leaky(c2[1] + b2v[1])), pack2(leaky(c2[2] + b2v[2]), leaky(c2[3] + b2v[3]))};
            *reinterpret_cast<uint2v *>(o_cur + (o_lane + (unsigned)(q * 8 * a.Cp_out * 2))) = d;
        }
    }
}

// ---- k_conv_f16_rwc: layer 2 (32 -> 64 channels at 208 x 208, + its 2x2 pool) with the weights in registers ---------------------
// The layer's whole weight set - 64 channels x 9 taps x 32 input channels = 36 B fragments of v_mfma_f32_16x16x32_f16 - fits the
// registers of ONE wavefront, so here EVERY wavefront holds all of it and the four wavefronts of a workgroup (one per SIMD) split the
// PIXELS of a tile: a tile is two image rows = 26 blocks of 16 pixels (pool-window order m = 4 w + 2 dy + dx as in k_conv_f16_rw);
// wavefront w takes blocks 6 w .. 6 w + 5 whole (9 fragment reads, 36 MFMAs each) and HALF of block 24 + (w >> 1) - two of the four
// channel blocks, 18 MFMAs: 234 MFMAs per wavefront and tile, the same for all four.  (The channel blocks are numbered per wavefront,
// register block i = channel block (i + 2 (w & 1)) & 3, so that the half block is always register blocks 0 and 1.)  An A fragment
// feeds four MFMAs and is read by one wavefront only: 252 ds_read_b128 per tile where k_conv_f16_rw needs 936.
//   * Input: a workgroup walks a RUN of consecutive row pairs of one image and keeps a RING of eight image rows in LDS (row = 212
//     items of 64 bytes: x = -1 .. 210, the layout's zero columns included - no masks; pitch 212 with the key {row bit 2 -> slot
//     bit 1} makes the fragment reads conflict-free, brute-forced like k_conv_f16_rwb's).  Tile t reads ring rows 2 t .. 2 t + 3 and
//     stages rows 2 t + 6, 2 t + 7 - two rows per tile instead of four, two tiles ahead (a fill has a whole tile to land).
//   * ONE barrier per tile, in the MIDDLE of it (after three blocks): what was requested before the previous barrier is complete
//     and visible after this one, so the fragment reads run on across tile boundaries (block 0 of tile t + 1 is prefetched during
//     the half block of tile t) and no wavefront ever waits at a tile's start.  Rows requested after barrier t are used from tile
//     t + 2 on; they overwrite rows 2 t - 2, 2 t - 1, which nobody reads once everybody has arrived at barrier t.
//   * Epilogue as micro-ops between the MFMAs like k_conv_f16_rwb MODE 1 (max over the four registers = one pool window, bias,
//     leaky, fp16, 2-byte write into a pooled LDS tile); the pooled row of tile t - 1 is stored after barrier t; three pooled
//     tiles, so that a tile's writes (from its second block on) can never meet the reads of the tile three before it.
// Host: W = 208, 32-channel items in, 64-channel items out, leaky, H / 2 divisible by the run length (build_f16_plan).
__global__ __launch_bounds__(256) void k_conv_f16_rwc(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh, const float *__restrict__ bias,
                                                       _Float16 *__restrict__ out, const ConvF16Args a, const int run_len, const int n_runs)
{
    constexpr int NBLK = 26, WP = 8 * NBLK + 1, LP = 212, PD = 9, NFULL = 6, NGRP = NFULL + 1, RING = 8;
    constexpr unsigned kRowBytes = LP * 64;                     // one ring row
    constexpr int kPairPieces = (2 * LP + 15) / 16, kFill = (kPairPieces + 3) / 4;   // 27 pieces of 16 LDS rows per two image rows, 7 per wavefront
    constexpr int kCtP = 72;                                    // halves per pixel of a pooled tile
    constexpr unsigned kCtBytes = NBLK * 4 * kCtP * 2;
    typedef float acc_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    _Float16 *As = smem_h;                                      // ring: [8][LP][32]
    _Float16 *Xs = smem_h + (size_t)RING * LP * 32;             // pooled tiles [3][104][kCtP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wsel = wave & 1;
    const int mi = lane & 15, kq = lane >> 4;
    const int dy = (mi >> 1) & 1, dx = mi & 1, pw = mi >> 2;
    const int rpi = (a.H >> 1) / run_len;                       // runs per image

    // ---- resident weights and epilogue constants
    half8_t bfr[9][4];
    float bv[4];
    int chn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        chn[i] = 16 * ((i + 2 * wsel) & 3) + mi;
        bv[i] = bias[chn[i]];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) bfr[tap][i] = *reinterpret_cast<const half8_t *>(wh + ((size_t)chn[i] * 9 + tap) * 32 + 8 * kq);
    }
    const unsigned as_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)As;
    const unsigned xs_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) _Float16 *)Xs;
    // A fragment (tap, block): ring row ((2 t + dy + ty) & 7), LDS row c0 + 8 block of it, c0 = 2 pw + dx + tx; the 16-byte slot is
    // kq ^ key, key = 2 ((ring row ^ (c0 >> 2)) & 1) - and the ring row's parity is (dy + ty) & 1 whatever t is
    unsigned cbase[9], ecur[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ty = tap / 3, tx = tap % 3, c0 = 2 * pw + dx + tx, key = 2 * (((dy + ty) ^ (c0 >> 2)) & 1);
        cbase[tap] = as_lds + (unsigned)(c0 * 64 + ((kq ^ key) * 16) + 6 * wave * 512);
    }
    const int ehd = (24 + (wave >> 1) - 6 * wave) * 512;        // from this wavefront's block 0 to its half block
    unsigned ctw[4], cth[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ctw[i] = xs_lds + (unsigned)(((4 * 6 * wave + kq) * kCtP + chn[i]) * 2);        // pooled pixel 4 block + kq
#pragma unroll
    for (int i = 0; i < 2; ++i) cth[i] = xs_lds + (unsigned)(((4 * (24 + (wave >> 1)) + kq) * kCtP + chn[i]) * 2);

    // ---- staging of two image rows (424 LDS rows) as 27 pieces of 16 rows; the last piece starts at row 408 (it repeats eight rows of
    // the one before rather than run past the pair)
    int relp[kFill];
#pragma unroll
    for (int it = 0; it < kFill; ++it) {
        const int p = min(wave + 4 * it, kPairPieces - 1), r0 = min(p * 16, 2 * LP - 16), row = r0 + (lane >> 2), ir = row / LP, c = row - ir * LP;
        relp[it] = (ir * WP + c - 1) * 64 + (((lane & 3) ^ (2 * ((ir ^ (c >> 2)) & 1))) * 16);
    }
    const char *abase = reinterpret_cast<const char *>(act);
    int run_base = 0;                                            // byte offset of item (b, y_start - 1, 0)
    auto fill_piece = [&](int pair, int it) {                   // rows 2 pair, 2 pair + 1 of the run
        const int p = min(wave + 4 * it, kPairPieces - 1), r0 = min(p * 16, 2 * LP - 16);
        lds_dma16(abase, (unsigned)(run_base + pair * (2 * WP * 64) + relp[it]), As + ((size_t)((2 * pair) & 7) * LP + r0) * 32);
    };

    half8_t fr[2][PD], frh[PD];                  // fragments of the full blocks (two sets) and of the half block
    acc_t acc[2][4], acch[2];
    float pm[4], pt[4];
    unsigned ph[4];
    half8_t pst[4];
    char *o_prev = nullptr, *o_this = nullptr;
    unsigned ct_cur = 0, ct_prev = 0;                            // byte offsets of this / the previous tile's pooled LDS tile

    auto issue_a = [&](int g, int tap) {                        // group g < 6: block 6 wave + g; group 6: the half block
        if (g < NFULL) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[g & 1][tap]) : "v"(ecur[tap]), "i"(g * 512) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=v"(frh[tap]) : "v"(ecur[tap] + (unsigned)ehd) : "memory");
    };
    auto set_ecur = [&](int ti) {
        const int s0 = 2 * ti + dy;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) ecur[tap] = cbase[tap] + (unsigned)((s0 + tap / 3) & 7) * kRowBytes;
    };
    // epilogue micro-ops of one accumulator (register block i): 0..5 values, 6 the LDS write
    auto epi1 = [&](int j, const acc_t &v, int i, unsigned dst, int imm_blocks) {
        // (maxima of MFMA results as instructions: see k_conv_f16_rwb's drain1)
        if (j == 0) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(pm[i]) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
        else if (j == 1) asm("v_max_f32 %0, %1, %2" : "=v"(pm[i]) : "v"(pm[i]), "v"(v[3]));
        else if (j == 2) pm[i] = pm[i] + bv[i];
        else if (j == 3) pt[i] = pm[i] * 0.1f;
        else if (j == 4) pm[i] = fmaxf(pm[i], pt[i]);            // leaky == (v < 0 ? 0.1 v : v)
        else if (j == 5) ph[i] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)pm[i]);
        else asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(dst), "v"(ph[i]), "i"(imm_blocks * 4 * kCtP * 2) : "memory");   // (block index inside this wavefront's six)
    };
    // micro-op position k of the drain of full block pj (its accumulators in set pj & 1): 0..23 values, 24..27 writes
    auto drain_full = [&](int pj, int k, unsigned cto) {
        if (k < 24) epi1(k % 6, acc[pj & 1][k / 6], k / 6, 0u, 0);
        else if (k < 28) epi1(6, acc[pj & 1][k - 24], k - 24, ctw[k - 24] + cto, pj);
    };
    auto drain_half = [&](int k, unsigned cto) {                // 0..11 values, 12, 13 writes
        if (k < 12) epi1(k % 6, acch[k / 6], k / 6, 0u, 0);
        else if (k < 14) epi1(6, acch[k - 12], k - 12, cth[k - 12] + cto, 0);
    };

    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const int b = run / rpi, ys = (run - b * rpi) * 2 * run_len;
        run_base = (kLead + b * a.PL + ys * WP) * 64;           // item (b, ys - 1, 0)
        char *orow0 = reinterpret_cast<char *>(out) + ((size_t)kLead + (size_t)b * a.oPL + (size_t)((ys >> 1) + 1) * a.oWp) * a.Cp_out * 2 + (size_t)a.out_ch_off * 2;
        __syncthreads();                                         // (a further run: everybody has left the previous one)
#pragma unroll
        for (int pair = 0; pair < 3; ++pair)
#pragma unroll
            for (int it = 0; it < kFill; ++it) fill_piece(pair, it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        set_ecur(0);
#pragma unroll
        for (int tap = 0; tap < PD; ++tap) issue_a(0, tap);
        o_this = orow0;
        o_prev = orow0;                                          // (tile -1: garbage that tile 0's own row overwrites later, see k_conv_f16_rwb)
        ct_cur = 0; ct_prev = 2 * kCtBytes;

        for (int ti = 0; ti < run_len; ++ti) {
            // pair index staged during this tile: rows 2 ti + 6, 2 ti + 7; past the run's last rows the last pair is staged again
            const int fpair = min(ti + 3, run_len);
#pragma clang loop unroll(full)
            for (int g = 0; g < NGRP; ++g) {
                if (g == 3) {                                    // the tile's barrier: fills requested during the previous tile have landed for everybody
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int ncb = g < NFULL ? 4 : 2;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i >= ncb) continue;
                        const int m = ncb * tap + i;
                        if (g < NFULL) {
                            const acc_t c0 = tap == 0 ? acc_t{0.f, 0.f, 0.f, 0.f} : acc[g & 1][i];
                            acc[g & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[g & 1][tap], bfr[tap][i], c0, 0, 0, 0);
                        } else {
                            const acc_t c0 = tap == 0 ? acc_t{0.f, 0.f, 0.f, 0.f} : acch[i];
                            acch[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frh[tap], bfr[tap][i], c0, 0, 0, 0);
                        }
                        if (m < 5) {                             // the next group's nine fragments: two per slot
                            // (group 6 prefetches block 0 of the NEXT tile: ecur was advanced during group 5; after the run's last tile
                            //  it reads rows nobody will use)
                            const int ng = g + 1 < NGRP ? g + 1 : 0;
                            issue_a(ng, 2 * m);
                            if (2 * m + 1 < PD) issue_a(ng, 2 * m + 1);
                        } else {
                            const int k0 = 2 * (m - 5);
#if (Y2_RWB_ABL & 4)
                            if (m == ncb * 9 - 1) { if (g < NFULL) asm volatile("" ::"v"(acc[g & 1][0]), "v"(acc[g & 1][1]), "v"(acc[g & 1][2]), "v"(acc[g & 1][3])); else asm volatile("" ::"v"(acc[0][0]), "v"(acch[0]), "v"(acch[1])); }
                            for (int k = k0; k < k0; ++k) {      // diagnostic: no epilogue work between the MFMAs (they are kept alive)
#else
#pragma unroll
                            for (int k = k0; k < k0 + 2; ++k) {
#endif
                                if (g == 0) {
                                    // the previous tile's half block, and the last two writes of its sixth block (the half block's 13 free slots hold 26 of its 28 micro-ops)
                                    if (k < 14) drain_half(k, ct_prev);
                                    else if (k < 16) drain_full(NFULL - 1, 26 + (k - 14), ct_prev);
                                } else if (g < NGRP - 1) {
                                    if (k < 28) drain_full(g - 1, k, ct_cur);
                                } else {
                                    if (k < 26) drain_full(NFULL - 1, k, ct_cur);
                                }
                                if (g == 3 && k >= 30 && k < 34) {         // the pooled row of tile ti - 1: LDS reads here, stores one block later
                                    const int piece = min(tid + (k - 30) * 256, NBLK * 4 * 8 - 1), px = piece >> 3, ck = piece & 7;
                                    asm volatile("ds_read_b128 %0, %1" : "=v"(pst[k - 30]) : "v"(xs_lds + ct_prev + (unsigned)((px * kCtP + ck * 8) * 2)) : "memory");
                                }
                                if (g == 4 && k >= 30 && k < 34) {
                                    const int piece = min(tid + (k - 30) * 256, NBLK * 4 * 8 - 1), px = piece >> 3, ck = piece & 7;
                                    *reinterpret_cast<half8_t *>(o_prev + (size_t)((px * a.Cp_out + ck * 8) * 2)) = pst[k - 30];
                                }
                                if (g == 5 && k == 30) set_ecur(ti + 1);   // (the half block's fragments were requested in this group's first slots)
                            }
                        }
                        if ((g == 3 && m >= 20 && m < 24) || (g == 4 && m >= 20 && m < 23)) fill_piece(fpair, g == 3 ? m - 20 : 4 + (m - 20));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            o_prev = o_this;
            o_this += (size_t)a.oWp * a.Cp_out * 2;
            ct_prev = ct_cur;
            ct_cur = ct_cur == 2 * kCtBytes ? 0u : ct_cur + kCtBytes;
        }
        // ---- after the run's last tile: what the next tile would have hidden
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 14; ++k) drain_half(k, ct_prev);
#pragma unroll
        for (int k = 26; k < 28; ++k) drain_full(NFULL - 1, k, ct_prev);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const _Float16 *Ct = Xs + (size_t)(ct_prev / 2);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int piece = tid + it * 256, px = piece >> 3, ck = piece & 7;
                if (piece < NBLK * 4 * 8)
                    *reinterpret_cast<half8_t *>(o_prev + (size_t)((px * a.Cp_out + ck * 8) * 2)) = *reinterpret_cast<const half8_t *>(Ct + (size_t)px * kCtP + ck * 8);
            }
        }
    }
}

// ---- 1x1 layers: persistent workgroups over a ring of staged K-steps --------------------------
// A 1x1 layer is a plain GEMM [pixels x Cin] x [Cin x Cout] with 2 (layer 5) to 16 (layers 19/21/30) K-steps of 64
// channels per tile: in the one-tile-per-workgroup kernels above its time is the per-tile set-up, the prologue fill
// latency and the LDS-transposed epilogue, not the MFMAs (layer 26: 0.27 ms for 17 us of HBM traffic).  Here a
// workgroup is PERSISTENT: it walks its share of the tiles, and the (tile, K-step) stages of all of them form ONE
// sequence through a ring of NS LDS stage buffers, filled NS-1 stages ahead by LDS-DMA - the next tile's first stages
// are in flight while this tile is multiplied and stored.  The epilogue uses no LDS (so it cannot collide with the
// ring): the MFMA operands are swapped (weights as the row operand), which leaves every lane with 4 CONSECUTIVE
// channels of one pixel per accumulator group; v_permlane32_swap pairs the two lane halves' groups into 8 channels and
// the lane stores 16 bytes.  One barrier per K-step; counted s_waitcnt vmcnt(P (NS-2)) instead of a full drain.
// Tiles are dealt XCD-aware like xcd_logical_id(): XCD k owns a contiguous range of the (pixel tile, n-tile) sequence.
template <int BM, int BN, int NS>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void k_gemm1_f16_p(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                                            const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                                            float *__restrict__ out_f32, const ConvF16Args a, const int n_tile_total)
{
    constexpr int BK = 64, ROWH = BK, WM = BM / 64, WN = BN / 64, NW = WM * WN;
    constexpr int AG = BM / 8 / NW, BG = BN / 8 / NW, P = AG + BG;     // LDS-DMA pieces per wavefront and stage
    constexpr int STAGE = (BM + BN) * ROWH;                            // halves per ring stage
    static_assert(AG >= 1 && BG >= 1 && NS >= 2 && NS <= 4 && P * (NS - 2) <= 63, "ring shape");
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int HW = a.H * a.W;
    // this workgroup's tiles: t_first, t_first + S, ... below t_end
    const int xcd = (int)blockIdx.x & 7, S = (int)gridDim.x >> 3;
    const int t_first = (int)((long)n_tile_total * xcd / 8) + ((int)blockIdx.x >> 3);
    const int t_end = (int)((long)n_tile_total * (xcd + 1) / 8);
    const int my_n = t_first < t_end ? (t_end - t_first + S - 1) / S : 0;
    if (my_n == 0) return;
    const int ksteps = a.Cp_in / BK, total = my_n * ksteps;

    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned a_src[AG], b_src[BG];       // 32-bit byte offsets from wave-uniform bases (tensors below 4 GiB: host check)
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave * BG + i) * 8 + lrow;
        b_src[i] = (unsigned)(((size_t)row * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8)) * 2);
    }
    // ---- producer side: stage (f_i, f_k) = K-step f_k of this workgroup's f_i-th tile
    int f_i = 0, f_k = 0, f_buf = 0;
    const char *f_wbase = nullptr;
    auto fill_tile_setup = [&](int ord) {
        const int t = t_first + ord * S, pt = t / a.n_tiles, nt = t - pt * a.n_tiles;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int row = (wave * AG + i) * 8 + lrow;
            const int q = min(pt * BM + row, a.npix - 1);           // rows past the last pixel re-read it (never stored)
            a_src[i] = (unsigned)((((size_t)kLead + flat_of_fast(a, q)) * a.Cp_in + (size_t)((lslot ^ ((row >> 1) & 7)) * 8)) * 2);
        }
        f_wbase = reinterpret_cast<const char *>(wh + (size_t)nt * BN * a.Cp_in);
    };
    auto fill = [&]() {
        const char *ab = reinterpret_cast<const char *>(act + f_k * BK);
        const char *bb = f_wbase + (size_t)f_k * BK * 2;
        _Float16 *As = smem_h + (size_t)f_buf * STAGE, *Bs = As + BM * ROWH;
#pragma unroll
        for (int i = 0; i < AG; ++i)
            lds_dma16(ab, a_src[i], As + (size_t)(wave * AG + i) * 8 * ROWH);
#pragma unroll
        for (int i = 0; i < BG; ++i)
            lds_dma16(bb, b_src[i], Bs + (size_t)(wave * BG + i) * 8 * ROWH);
        f_buf = f_buf == NS - 1 ? 0 : f_buf + 1;
        if (++f_k == ksteps) {
            f_k = 0;
            if (++f_i < my_n) fill_tile_setup(f_i);
        }
    };
    fill_tile_setup(0);
    int issued = 0;
    for (; issued < NS - 1 && issued < total; ++issued) fill();

    const int frow = lane & 31, fhalf = lane >> 5;
    int sg = 0, cur = 0;
    for (int ti = 0; ti < my_n; ++ti) {
        const int t = t_first + ti * S, pt = t / a.n_tiles, nt = t - pt * a.n_tiles;
        const int q0 = pt * BM, n0 = nt * BN;
        float16_t acc[2][2];             // [pixel 32-tile i][channel 32-tile j]: D rows = channels, columns = pixels
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int k = 0; k < ksteps; ++k, ++sg) {
            // stage sg has landed once at most the NS-2 younger stages' pieces of this wavefront are outstanding
            if (issued - sg - 1 >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();             // ... for every wavefront, and all of them are done reading stage sg-1
            if (issued < total) { fill(); ++issued; }   // into the buffer stage sg-1 used
            const _Float16 *As = smem_h + (size_t)cur * STAGE, *Bs = As + BM * ROWH;
            half8_t af[2][2], bf[2][2];
            auto read_frags = [&](int kk, int set) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = wm * 64 + u * 32 + frow;
                    af[set][u] = *reinterpret_cast<const half8_t *>(As + (size_t)row * ROWH + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) * 8));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = wn * 64 + u * 32 + frow;
                    bf[set][u] = *reinterpret_cast<const half8_t *>(Bs + (size_t)row * ROWH + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) * 8));
                }
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (kk + 1 < BK / 16) read_frags(kk + 1, (kk + 1) & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk & 1][j], af[kk & 1][i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int kk = 0; kk + 1 < BK / 16; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            cur = cur == NS - 1 ? 0 : cur + 1;
        }

        // ---- epilogue straight from the accumulators.  Lane: pixel column lane & 31 of each pixel 32-tile; accumulator
        // register r of channel 32-tile j = channel j*32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3).
        if (out_f32) {                   // the region layer: dense [B][N][H][W] fp32 (32 lanes = 32 consecutive pixels of a channel)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = q0 + wm * 64 + i * 32 + frow;
                if (q >= a.npix) continue;
                const int b = (int)fast_div((unsigned)q, a.mHW, a.sHW), rem = q - b * HW;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = n0 + wn * 64 + j * 32 + 8 * (r >> 2) + 4 * fhalf + (r & 3);
                        if (ch >= a.N) continue;
                        float v = acc[i][j][r] + bias[ch];
                        if (a.leaky && v < 0.f) v *= 0.1f;
                        out_f32[((size_t)b * a.N + ch) * HW + rem] = v;
                    }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = q0 + wm * 64 + i * 32 + frow;
            const bool qok = q < a.npix;
            _Float16 *orow = out + ((size_t)kLead + flat_of_fast(a, min(q, a.npix - 1))) * a.Cp_out + a.out_ch_off;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cb = n0 + wn * 64 + j * 32;
                unsigned pk[4][2];       // [group g = r >> 2][dword]: this lane's 4 channels 8 g + 4 (lane >> 5) ..+3 as halves
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + cb + 8 * g + 4 * fhalf);
                    float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
                    float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                    if (a.leaky) {
                        v0 = v0 < 0.f ? v0 * 0.1f : v0; v1 = v1 < 0.f ? v1 * 0.1f : v1;
                        v2 = v2 < 0.f ? v2 * 0.1f : v2; v3 = v3 < 0.f ? v3 * 0.1f : v3;
                    }
                    const half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                    pk[g][0] = __builtin_bit_cast(unsigned, h01);
                    pk[g][1] = __builtin_bit_cast(unsigned, h23);
                }
                // lanes 0-31 end up with channels 8 g' .. 8 g' + 7 of group g' = 2 pr, lanes 32-63 with those of 2 pr + 1
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                    const uint2v s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                    const uint2v s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                    const int c0 = cb + 8 * (2 * pr + fhalf);
                    if (qok && c0 < a.n_store) *reinterpret_cast<uint4 *>(orow + c0) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
    }
}

// ---- the 32-channel 3x3 layer (layer 2: 208x208, 32 -> 64, K = 288) + its 2x2 pool ---------------------
// With 32-channel items a K-step of one tap is two MFMA k-slices: in the per-tap kernels above this layer is one
// barrier per 4 MFMAs and runs at 0.5 PF.  Here nothing is staged per tap: one workgroup owns a 16 x 16 tile of conv
// outputs x all 64 channels; its 18 x 18 input patch (324 rows of 64 B; pixels outside the image are the layout's
// zero pad items, no masking) and ALL nine taps' weights (9 x 64 rows of 64 B = 36 KB) go into LDS once by LDS-DMA,
// then four wavefronts (four tile rows each) run 72 MFMAs apiece with no barrier in between.  57 KB of LDS: two
// workgroups per CU, one stages or stores while the other multiplies.
// LDS rows are 64 B, four to a bank row: a ds_read_b128 lane group is conflict-free when its 16 rows are distinct
// mod 16 (slot = chunk ^ ((row >> 2) & 3)).  A 32-row MFMA block is two tile rows (patch rows 18 apart): its second
// half is rotated - MFMA row 16 + j = pixel (j - 2) mod 16 of the lower tile row - which makes both hardware lane
// groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} cover all 16 residues for every tap shift.
// Operands swapped as in k_gemm1_f16_p (D rows = channels): a lane leaves with 4 consecutive channels of a pixel per
// accumulator group, written as 8-byte pieces into a [pixel][channel] tile in LDS (over the dead patch / weights); the
// 2x2 pool is a max over four rows of that tile (same bits as store_pooled: bias + leaky + fp16 rounding are
// monotonic), stored in 16-byte pieces.  K order tap-major like k_conv_f16: the same bits as that kernel.
__global__ __launch_bounds__(256) void k_conv_f16_c32_pool(const _Float16 *__restrict__ act, const _Float16 *__restrict__ wh,
                                                          const float *__restrict__ bias, _Float16 *__restrict__ out,
                                                          const ConvF16Args a, const int n_tile_total)
{
    constexpr int TS = 16, PW = TS + 2;                         // tile side, patch pitch (pixels); the patch is staged as 21 DMA pieces of 16 rows = 336 LDS rows
    constexpr int ROWB = 64;                                    // bytes per LDS row (32 halves)
    constexpr int kWRows = 9 * 64, kCtPitch = 144;              // weight rows; bytes per pixel row of the epilogue tile (64 ch + pad)
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_h[];
    char *lds = reinterpret_cast<char *>(smem_h);
    char *Wl = lds;                                             // [9 * 64][64 B]
    char *Pl = lds + kWRows * ROWB;                             // [336][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.W / TS, tiles_y = a.H / TS;

    // ---- the weights are staged ONCE per workgroup (36 pieces: piece p covers LDS rows 16 p .. 16 p + 15, lane = (row, slot));
    // the workgroup then walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    const int lr = lane >> 2, slot = lane & 3;
    {
        const char *wb = reinterpret_cast<const char *>(wh);
        for (int p = wave; p < 36; p += 4) {
            const int row = p * 16 + lr, tap = row >> 6, n = row & 63;
            const unsigned src = (unsigned)(((n * 9 + tap) * 32 + ((slot ^ ((row >> 2) & 3)) * 8)) * 2);
            lds_dma16(wb, src, reinterpret_cast<_Float16 *>(Wl + p * 1024));
        }
    }
    // ---- this lane's fragment rows
    const int c = lane & 31, h = lane >> 5;
    const int pxl = c < 16 ? c : ((c - 18) & 15);               // pixel column of MFMA row c inside its tile row
    int arow[2];                                                // patch row of the centre tap, pixel block i = tile rows (4 wave + 2 i, +1)
#pragma unroll
    for (int i = 0; i < 2; ++i) arow[i] = (4 * wave + 2 * i + (c >> 4) + 1) * PW + pxl + 1;
    const int bkey = (c >> 2) & 3;                              // weight row = tap * 64 + j * 32 + c: its swizzle key depends on c only
    int boff[2];                                                // byte offset of this lane's weight fragment, k-slice 0 / 1, at tap 0, j 0
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) boff[kk] = c * ROWB + (((kk * 2 + h) ^ bkey) * 16);

    for (int tile = (int)blockIdx.x; tile < n_tile_total; tile += (int)gridDim.x) {
    const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
    const int ty0 = (tr / tiles_x) * TS, tx0 = (tr % tiles_x) * TS;
    {   // stage the patch (21 pieces); item of patch pixel (py, px): b PL + (ty0 + py) Wp + tx0 - 1 + px   (pixel (ty0 - 1 + py, tx0 - 1 + px))
        const char *ab = reinterpret_cast<const char *>(act + ((size_t)kLead + (size_t)b * a.PL + (size_t)ty0 * a.Wp + tx0 - 1) * 32);
        for (int p = wave; p < 21; p += 4) {
            const int row = min(p * 16 + lr, PW * PW - 1), py = row / PW, px = row - py * PW;
            const unsigned src = (unsigned)(((py * a.Wp + px) * 32 + ((slot ^ (((p * 16 + lr) >> 2) & 3)) * 8)) * 2);
            lds_dma16(ab, src, reinterpret_cast<_Float16 *>(Pl + p * 1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    half8_t af[2][2], bf[2][2];
    auto read_frags = [&](int tap, int kk, int set) {
        const int toff = (tap / 3 - 1) * PW + (tap % 3 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = arow[i] + toff;
            af[set][i] = *reinterpret_cast<const half8_t *>(Pl + R * ROWB + (((kk * 2 + h) ^ ((R >> 2) & 3)) * 16));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[set][j] = *reinterpret_cast<const half8_t *>(Wl + tap * (64 * ROWB) + j * (32 * ROWB) + boff[kk]);
    };
    read_frags(0, 0, 0);
#pragma unroll
    for (int s = 0; s < 18; ++s) {                              // slice s = (tap s / 2, k-slice s % 2); the next slice's fragments are read ahead
        if (s + 1 < 18) read_frags((s + 1) >> 1, (s + 1) & 1, (s + 1) & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                            // every wavefront is done with the patch and the weights

    // ---- epilogue: bias + leaky + fp16; the horizontal half of the 2x2 pool is a max with the neighbouring lane (MFMA
    // columns 2p, 2p + 1 are the pixels 2p, 2p + 1 of one tile row in both halves of a block), the even lanes write
    // [tile row][pooled column][channel] into LDS over the dead patch (8-byte pieces), the vertical half is a max over
    // two rows of that tile; 16-byte stores.  Same bits as store_pooled: bias + leaky + fp16 rounding are monotonic.
    char *Ct = Pl;                                              // [16][8][kCtPitch]: 18 KB of the patch's 21 KB
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int prow = (4 * wave + 2 * i + (c >> 4)) * 8 + (pxl >> 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {                       // register r = channel j * 32 + 8 (r >> 2) + 4 h + (r & 3)
                const int ch = j * 32 + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4 *>(bias + ch);
                float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
                float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                if (a.leaky) {
                    v0 = v0 < 0.f ? v0 * 0.1f : v0; v1 = v1 < 0.f ? v1 * 0.1f : v1;
                    v2 = v2 < 0.f ? v2 * 0.1f : v2; v3 = v3 < 0.f ? v3 * 0.1f : v3;
                }
                half2_t h01 = {(_Float16)v0, (_Float16)v1}, h23 = {(_Float16)v2, (_Float16)v3};
                const unsigned n01 = (unsigned)__builtin_amdgcn_mov_dpp((int)__builtin_bit_cast(unsigned, h01), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                const unsigned n23 = (unsigned)__builtin_amdgcn_mov_dpp((int)__builtin_bit_cast(unsigned, h23), 0xB1, 0xF, 0xF, true);
                h01 = __builtin_elementwise_max(h01, __builtin_bit_cast(half2_t, n01));
                h23 = __builtin_elementwise_max(h23, __builtin_bit_cast(half2_t, n23));
                if (!(c & 1))
                    *reinterpret_cast<uint2 *>(Ct + prow * kCtPitch + ch * 2) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            }
    }
    __syncthreads();
    {   // 64 pooled pixels x 4 pieces of 16 channels: one per thread
        const int pp = tid >> 2, ck = tid & 3, py = pp >> 3, px = pp & 7;
        half8_t m[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char *p00 = Ct + ((2 * py) * 8 + px) * kCtPitch + ck * 32 + q * 16;
            m[q] = __builtin_elementwise_max(*reinterpret_cast<const half8_t *>(p00), *reinterpret_cast<const half8_t *>(p00 + 8 * kCtPitch));
        }
        const int oy = ty0 / 2 + py, ox = tx0 / 2 + px;
        _Float16 *o = out + ((size_t)kLead + (size_t)b * a.oPL + (size_t)(oy + 1) * a.oWp + ox) * a.Cp_out + a.out_ch_off + ck * 16;
        *reinterpret_cast<half8_t *>(o) = m[0];
        *reinterpret_cast<half8_t *>(o + 8) = m[1];
    }
    __syncthreads();                                            // the pooled tile is read before the next patch lands on it
    }
}

// Layer 0 + layer 1 fused (conv 3->32 3x3 + leaky + 2x2 max pool) straight from the float frames:
// K = 27 is too thin for the matrix cores, and the 416x416x32 intermediate is never needed again
// (yolov2.cfg: layer 1 is its only consumer), so this kernel keeps it in registers.  One lane owns
// one pooled pixel = a 2x2 block of conv outputs x 16 channels (blockIdx.y picks the channel half)
// = 64 fp32 accumulators; the 27x32 weights are read from LDS as broadcast 16-byte reads;
// output = layer-1 items (32 halves).
// w0: [27][32] fp32 (k = c*9 + i*3 + j), bias0: [32] fp32.
// SPLIT (fp32tol mode): nothing is rounded to fp16 before the pool; the pooled fp32 value leaves as (hi, lo) in an item of 128 halves
// [hi 32 | lo 32 | hi 32 | 32 zero].
template <bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_conv0_pool_f16(const float *__restrict__ frames, const float *__restrict__ w0,
                                                         const float *__restrict__ bias0, _Float16 *__restrict__ out,
                                                         int B, int H, int W, int oWp, int oPL)
{
    __shared__ __attribute__((aligned(16))) float ws[27 * 32 + 32];
    for (int i = threadIdx.x; i < 27 * 32 + 32; i += 256) ws[i] = i < 27 * 32 ? w0[i] : bias0[i - 27 * 32];
    __syncthreads();
    const int OH = H / 2, OW = W / 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B * OH * OW) return;
    const int b = q / (OH * OW), r = q - b * (OH * OW), oy = r / OW, ox = r - oy * OW;
    const int nh = blockIdx.y * 16;   // channel half
    float acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[p][n] = ws[27 * 32 + nh + n];
    const size_t HWs = (size_t)H * W;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        float in[4][4];   // the 4x4 input patch of this 2x2 output block
        // branch-free: clamped addresses are always valid, out-of-image taps are zeroed afterwards,
        // so the 16 loads issue back to back instead of one latency-exposed load per guarded branch
        const float *plane = frames + ((size_t)b * 3 + c) * HWs;
#pragma unroll
        for (int yy = 0; yy < 4; ++yy)
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const int sy = 2 * oy + yy - 1, sx = 2 * ox + xx - 1;
                in[yy][xx] = plane[(size_t)min(max(sy, 0), H - 1) * W + min(max(sx, 0), W - 1)];
            }
#pragma unroll
        for (int yy = 0; yy < 4; ++yy)
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const int sy = 2 * oy + yy - 1, sx = 2 * ox + xx - 1;
                in[yy][xx] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? in[yy][xx] : 0.f;
            }
        // weights of tap t+1 are read from LDS while tap t is multiplied (one tap of lookahead:
        // a fence per tap keeps hipcc from hoisting all 27 taps' reads and blowing the register file)
        float4 wcur[4], wnxt[4];
        {
            const float4 *wr = reinterpret_cast<const float4 *>(&ws[(c * 9) * 32 + nh]);
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) wcur[n4] = wr[n4];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int i = t / 3, j = t % 3;
            if (t < 8) {
                const float4 *wr = reinterpret_cast<const float4 *>(&ws[(c * 9 + t + 1) * 32 + nh]);
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) wnxt[n4] = wr[n4];
            }
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) {
                const float4 wv = wcur[n4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xv = in[(p >> 1) + i][(p & 1) + j];
                    acc[p][n4 * 4 + 0] = fmaf(wv.x, xv, acc[p][n4 * 4 + 0]);
                    acc[p][n4 * 4 + 1] = fmaf(wv.y, xv, acc[p][n4 * 4 + 1]);
                    acc[p][n4 * 4 + 2] = fmaf(wv.z, xv, acc[p][n4 * 4 + 2]);
                    acc[p][n4 * 4 + 3] = fmaf(wv.w, xv, acc[p][n4 * 4 + 3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) wcur[n4] = wnxt[n4];
        }
    }
    half8_t *dst = reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * (SPLIT ? 128 : 32) + nh);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        half8_t o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = k * 8 + e;
            float m = -3.0e38f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float v = acc[p][n];
                v = v < 0.f ? v * 0.1f : v;          // leaky, then pool (same order as the layer pipeline)
                // the layer-0 tensor is fp16 in the unfused pipeline: round before the max like it does
                m = fmaxf(m, SPLIT ? v : (float)(_Float16)v);
            }
            if constexpr (SPLIT) { _Float16 h, l; split_f32(m, h, l); o[e] = h; ol[e] = l; }
            else o[e] = (_Float16)m;
        }
        dst[k] = o;
        if constexpr (SPLIT) { dst[k + 4] = ol; dst[k + 8] = o; }     // parts at +32 and +64 halves
    }
}

// ------------------------------------------------------------------ split-fp16 ("fp32tol") helpers

// 2x2/2 max pool on split items [hi | lo | hi] of PS channels per part: the max of the fp32 values hi + lo (an exact sum), whose
// own (hi, lo) pair is copied.  One thread per (output pixel, 8-channel chunk of a part).
__global__ void k_maxpool2_split(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int PS, int Cp, int B, int OH, int OW,
                                 int iWp, int iPL, int oWp, int oPL)
{
    const int chunks = PS / 8;
    const long n = (long)B * OH * OW * chunks;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int ck = (int)(t % chunks);
    const long p = t / chunks;
    const int x = (int)(p % OW), y = (int)((p / OW) % OH), b = (int)(p / ((long)OW * OH));
    const size_t s = ((size_t)kLead + (size_t)b * iPL + (size_t)(2 * y + 1) * iWp + 2 * x) * Cp + ck * 8;
    const size_t offs[4] = {0, (size_t)Cp, (size_t)iWp * Cp, (size_t)iWp * Cp + Cp};
    half8_t bh = *reinterpret_cast<const half8_t *>(in + s), bl = *reinterpret_cast<const half8_t *>(in + s + PS);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const half8_t h = *reinterpret_cast<const half8_t *>(in + s + offs[k]), l = *reinterpret_cast<const half8_t *>(in + s + offs[k] + PS);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if ((float)h[e] + (float)l[e] > (float)bh[e] + (float)bl[e]) { bh[e] = h[e]; bl[e] = l[e]; }
    }
    _Float16 *o = out + ((size_t)kLead + (size_t)b * oPL + (size_t)(y + 1) * oWp + x) * Cp + ck * 8;
    *reinterpret_cast<half8_t *>(o) = bh;
    *reinterpret_cast<half8_t *>(o + PS) = bl;
    *reinterpret_cast<half8_t *>(o + 2 * PS) = bh;
}

// Darknet legacy reorg on split items: part p of the 64-channel input goes to channels [0, 256) of part p of the concat items.  Like
// k_reorg_f16 below: one thread per (frame, output pixel, part, PAIR of output channels), consecutive lanes write consecutive 4-byte pieces.
__global__ void k_reorg_split(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int B, int iPS, int iCp, int iWp, int iPL,
                              int oPS, int oCp, int oWp, int oPL)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 169 * 3 * 128) return;
    const int cp = t & 127, q = t >> 7, p = q % 3, r = q / 3, pix = r % 169, b = r / 169;
    const int oy = pix / 13, ox = pix - oy * 13;
    half2_t v;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int o = (2 * cp + e) * 169 + pix;        // the reference's flat output index [256][13][13]
        const int k = o / (26 * 416), rem = o - k * (26 * 416), j = rem / 26, i = rem - j * 26;
        const int sidx = (2 * i + (k & 1)) + 52 * (2 * j + (k >> 1));
        const int sc = sidx / 676, sr = sidx - sc * 676, sy = sr / 26, sx = sr - sy * 26;
        v[e] = in[((size_t)kLead + (size_t)b * iPL + (size_t)(sy + 1) * iWp + sx) * iCp + p * iPS + sc];
    }
    *reinterpret_cast<half2_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * oCp + p * oPS + 2 * cp) = v;
}

// weights_reorg (fp32 stream of one layer) -> wh[N_pad][KK][Cp] halves for the split mode: Cp = three parts of PS channels
// (+ zero padding), part 0 and part 1 hold w_hi = fp16(w), part 2 holds w_lo = fp16(w - w_hi): against activations [a_hi | a_lo | a_hi].
__global__ void k_pack_weights_split(const float *__restrict__ src, _Float16 *__restrict__ dst, float *__restrict__ bias_dst,
                                     const float *__restrict__ bias_src, int C, int N, int KK, int PS, int Cp, int Npad)
{
    const long n = (long)Npad * KK * Cp;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Npad) bias_dst[t] = t < N ? bias_src[t] : 0.f;
    if (t >= n) return;
    const int cc = (int)(t % Cp), part = cc / PS, ci = cc - part * PS;
    const int tap = (int)((t / Cp) % KK);
    const int m = (int)(t / ((long)Cp * KK));
    float v = 0.f;
    if (m < N && ci < C && part < 3) {
        const int m0 = m / kTm * kTm, tm = m - m0, tm_min = min(kTm, N - m0);
        const int n0 = ci / kTn * kTn, tn = ci - n0, tn_min = min(kTn, C - n0);
        v = src[(long)m0 * C * KK + (long)tm_min * n0 * KK + (long)tap * tm_min * tn_min + tm * tn_min + tn];
    }
    _Float16 h, l;
    split_f32(v, h, l);
    dst[t] = part == 2 ? l : h;
}

// ------------------------------------------------------------------ small fp16 kernels

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

// Darknet legacy reorg (yolo2_model.cpp:112-129) into channels [0,256) of the 1280-channel concat items.  One thread per (frame,
// output pixel, PAIR of output channels): consecutive lanes write consecutive 4-byte pieces of one item (coalesced), the two halves
// are gathered from the 64-channel source items.  (Round 3's one-thread-per-element form in [channel][pixel] order wrote 2-byte
// pieces 2560 bytes apart: 0.052 ms per 128 frames, as much as a 1x1 conv layer.)
__global__ void k_reorg_f16(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int B, int iCp, int iWp, int iPL,
                            int oCp, int oWp, int oPL)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 169 * 128) return;
    const int cp = t & 127, pix = (t >> 7) % 169, b = t / (169 * 128);
    const int oy = pix / 13, ox = pix - oy * 13;
    half2_t v;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int o = (2 * cp + e) * 169 + pix;        // the reference's flat output index [256][13][13]
        const int k = o / (26 * 416), rem = o - k * (26 * 416), j = rem / 26, i = rem - j * 26;
        const int sidx = (2 * i + (k & 1)) + 52 * (2 * j + (k >> 1));
        const int sc = sidx / 676, sr = sidx - sc * 676, sy = sr / 26, sx = sr - sy * 26;
        v[e] = in[((size_t)kLead + (size_t)b * iPL + (size_t)(sy + 1) * iWp + sx) * iCp + sc];
    }
    *reinterpret_cast<half2_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * oCp + 2 * cp) = v;
}

// weights_reorg (fp32 stream of one layer) -> wh[N_pad][KK][Cp] halves (zero padded).
// im2col_first: pack layer 0 as a 1x1 conv over k = c*9 + tap (kept for experiments; the pass uses w0f).
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

// Layers 0 + 1 on the matrix cores.  K = 27 is thin, but the fp32-VALU kernel above still needs
// 76 GFLOP of plain FMAs per 256 frames (1.25 ms, 12 % of the fp16 pass); as an im2col GEMM with K
// padded to 32 it is two v_mfma_f32_32x32x16_f16 per 32 pixels x 32 channels.  One workgroup owns a
// 16 x 32 tile of conv outputs (8 x 16 pooled pixels): the 18 x 34 x 3 input patch is converted to
// fp16 into LDS once, and each A fragment is gathered from it with 16-bit LDS reads (k = c*9 + tap
// -> patch[c][y + tap/3][x + tap%3]; rows k >= 27 of B are zero, so their A entries may be any
// finite patch value).  A 32-row MFMA block is 2 conv rows x 16 conv columns ordered
// row = 4*pooled_col + 2*dy + dx, which puts the four members of every 2x2 pool window into
// registers r&3 = 0..3 of ONE lane (C layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5)): the pool is an
// in-lane max, then bias + leaky (monotonic, so it commutes with the max).
// w0: [27][32] fp32 (k = c*9 + tap), bias0: [32] fp32; out = layer-1 items of 32 halves.
// SPLIT (the fp32-tolerance pass, 4.6 of DESIGN.md): frame values and weights as (hi, lo) fp16 pairs - a second patch and a second pair of
// B fragments -, every product as three MFMAs (hi hi + lo hi + hi lo) into the same fp32 accumulators, the pooled fp32 value split again
// into the layer-1 item's [hi | lo | hi] parts of 32 channels (items of 128 halves).  Replaces the fp32-VALU form k_conv0_pool_f16<true>
// there (0.59 ms per 64 frames, 9 % of that pass).
template <bool SPLIT = false>
__global__ __launch_bounds__(256) void k_conv0_pool_mfma(const float *__restrict__ frames, const float *__restrict__ w0,
                                                          const float *__restrict__ bias0, _Float16 *__restrict__ out, int H,
                                                          int W, int oWp, int oPL, int n_tile_total)
{
    // Patch = rows ty0-1 .. ty0+16, image columns tx0-4 .. tx0+35 (the tile's 34 plus three on either side so that every piece is
    // a 16-byte aligned float4 in the frame: tx0 is a multiple of 32): column tx0-1+p is stored at index p + PSH of its row.
    constexpr int TR = 16, TC = 32, PR = TR + 2, PCS = 40, PSH = 3, PV = 10;   // patch rows / row stride (halves) / shift / float4 pieces per row
    constexpr int PLO = 3 * PR * PCS;                                          // SPLIT: the lo patch follows the hi patch, the lo tile the hi tile
    __shared__ __attribute__((aligned(16))) _Float16 patch[(SPLIT ? 2 : 1) * 3 * PR * PCS];
    __shared__ __attribute__((aligned(16))) _Float16 otile[(SPLIT ? 2 : 1) * 8 * 16][40];   // pooled tile [pixel][32 ch + pad]: leaves in 16-byte stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = W / TC, tiles_y = H / TR;

    // B fragments (weights), constant for the whole kernel: lane (n = lane & 31, h = lane >> 5) holds B[16kk + 8h + j][n]
    const int n = lane & 31, h = lane >> 5;
    half8_t bfrag[2], bfragl[SPLIT ? 2 : 1];
    int aoff[2][8];   // patch offset (halves) of A element (kk, j) relative to the pixel's top-left tap
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * kk + 8 * h + j;
            const float wv = k < 27 ? w0[k * 32 + n] : 0.f;
            bfrag[kk][j] = (_Float16)wv;
            if constexpr (SPLIT) bfragl[kk][j] = (_Float16)(wv - (float)bfrag[kk][j]);
            const int c = k / 9, tap = k - c * 9;
            aoff[kk][j] = k < 27 ? (c * PR + tap / 3) * PCS + tap % 3 + PSH : PSH;
        }
    const float bv = bias0[n];
    // this lane's pixel inside an MFMA block: row r = 4*pc + 2*dy + dx
    const int r = lane & 31, dx = r & 1, dy = (r >> 1) & 1, pc = r >> 2;

    // The workgroup is PERSISTENT (round 3): it walks tiles blockIdx.x, + gridDim.x, ... and requests tile t + 1's patch
    // (registers) right after tile t's has been written to LDS, so the loads are in flight during tile t's gathers, MFMAs and
    // stores.  One tile per workgroup kept only a quarter of a workgroup's life's worth of bytes in flight: 2.2 TB/s.
    // Staging (round 4): 3 x 18 rows x 10 aligned float4 pieces = 540 pieces per tile, up to three per thread, each converted and
    // written to LDS as ONE 8-byte piece.  (Round 3 loaded the 1836 patch elements as 4-byte pieces that started 4 bytes before a
    // 128-byte line and wrote them with 2-byte LDS stores; with every request aimed at one cache-hot patch - Y2_C0_ABL = 2 - the
    // kernel took 0.160 instead of 0.237 ms per 128 frames: the loads, not the gathers, were what it waited for.)
    constexpr int NPC = 3 * PR * PV, NIT = (NPC + 255) / 256;
    int pel_off[NIT], pel_pk[NIT], pel_src[NIT];   // LDS offset (halves) of the piece; py | j << 8 | c << 16; its offset (floats) from the patch's first element
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(tid + it * 256, NPC - 1);
        const int row = i / PV, j = i - row * PV, c = row / PR, py = row - c * PR;
        pel_pk[it] = py | (j << 8) | (c << 16);
        pel_off[it] = row * PCS + 4 * j;
        pel_src[it] = (c * H + py) * W + 4 * j;
    }
    // Two tiles' pieces in flight: the set a tile is converted from was requested TWO tiles earlier (one tile ahead left ~0.05 of
    // 0.19 ms per 128 frames waiting for loads: Y2_C0_ABL = 2).
    float4 pvA[NIT], pvB[NIT];
    auto request = [&](int tile, float4 (&pv)[NIT]) {   // clamped addresses (always inside the plane), masked when written
#if (Y2_C0_ABL & 2)
        tile = (int)blockIdx.x;          // diagnostic: every request re-reads the workgroup's first (cache-hot) patch
#endif
        const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * TR, tx0 = (tr % tiles_x) * TC;
        const float *fb = frames + (size_t)b * 3 * H * W;
        if (ty0 > 0 && ty0 + TR < H && tx0 > 0 && tx0 + TC < W) {      // an interior tile (wave-uniform): the whole patch is inside the plane, no clamps
            const float *tb = fb + (size_t)(ty0 - 1) * W + (tx0 - 4);
#pragma unroll
            for (int it = 0; it < NIT; ++it) pv[it] = *reinterpret_cast<const float4 *>(tb + pel_src[it]);
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int sy = ty0 + (pel_pk[it] & 255) - 1, sx = tx0 - 4 + 4 * ((pel_pk[it] >> 8) & 255);
            pv[it] = *reinterpret_cast<const float4 *>(fb + ((size_t)(pel_pk[it] >> 16) * H + min(max(sy, 0), H - 1)) * W + min(max(sx, 0), W - 4));
        }
    };
    // Which tiles: round k of the grid works on tiles k G .. (k + 1) G - 1, and workgroups are dealt to the 8 XCDs round-robin, so in launch
    // order the two tiles either side of a tile - whose patches share its first and last 128-byte line of every row - and the tile rows
    // above and below (two shared image rows) run on OTHER XCDs and each L2 fetches those lines again: the PMC passes of
    // tools/f16_traffic.sh counted 848 MB read per 128 frames for 266 MB of frames (profiles/r04_f16_traffic.json), and with the 354 MB
    // of stores that is the fabric's whole rate (6.3 TB/s in 0.19 ms).  XCD x therefore takes the contiguous eighth x G/8 .. of every round:
    // ten tile rows of one image meet in one L2.  Same tiles, same bits.
    const int G = (int)gridDim.x;
#if defined(Y2_C0_NO_XCD)
    const int wg = (int)blockIdx.x;
#else
    const int wg = (G & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
#endif
    if (wg < n_tile_total) request(wg, pvA);
    if (wg + G < n_tile_total) request(wg + G, pvB);
    auto do_tile = [&](int tile, float4 (&pv)[NIT]) {
        const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * TR, tx0 = (tr % tiles_x) * TC;
        const bool interior = ty0 > 0 && ty0 + TR < H && tx0 > 0 && tx0 + TC < W;   // wave-uniform: nothing of the patch is outside the image
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < NPC) {
                bool keep = true;
                if (!interior) {          // (uniform branch: 78 % of the tiles skip the masks)
                    const int sy = ty0 + (pel_pk[it] & 255) - 1, sx = tx0 - 4 + 4 * ((pel_pk[it] >> 8) & 255);
                    keep = sy >= 0 && sy < H && sx >= 0 && sx < W;      // (a piece is wholly inside or wholly outside the image: W % 4 == 0)
                }
                typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                half4_t hv = {(_Float16)pv[it].x, (_Float16)pv[it].y, (_Float16)pv[it].z, (_Float16)pv[it].w};
                if (!keep) hv = half4_t{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                *reinterpret_cast<half4_t *>(&patch[pel_off[it]]) = hv;
                if constexpr (SPLIT) {
                    half4_t lv = {(_Float16)(pv[it].x - (float)hv[0]), (_Float16)(pv[it].y - (float)hv[1]), (_Float16)(pv[it].z - (float)hv[2]),
                                  (_Float16)(pv[it].w - (float)hv[3])};
                    if (!keep) lv = half4_t{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    *reinterpret_cast<half4_t *>(&patch[PLO + pel_off[it]]) = lv;
                }
            }
        }
        __syncthreads();   // the patch is complete; everybody has stored the previous tile's pooled rows
        if (tile + 2 * G < n_tile_total) request(tile + 2 * G, pv);

#if !(Y2_C0_ABL & 4)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int prow = wave * 2 + (mb >> 1), chalf = mb & 1;            // pooled row 0..7, column half 0..1 of the tile
            const int base = (2 * prow + dy) * PCS + (chalf * 16 + 2 * pc + dx);   // top-left tap of this lane's conv pixel
            float16_t acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8_t af;
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = patch[base + aoff[kk][j]];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag[kk], acc, 0, 0, 0);
                if constexpr (SPLIT) {
                    half8_t al;
#pragma unroll
                    for (int j = 0; j < 8; ++j) al[j] = patch[PLO + base + aoff[kk][j]];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bfrag[kk], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfragl[kk], acc, 0, 0, 0);
                }
            }
            // lane holds channel n for pooled columns 2g + h (g = 0..3): registers 4g .. 4g+3 are one pool window
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // (fmaxf() on MFMA results - which the compiler cannot prove canonical - first runs every operand through a v_max x, x: the
                //  pool's maxima were nine instructions.  Inline-asm maxima as in k_conv_f16_rwb are NOT an option here: they would read the
                //  accumulators right behind the MFMAs, and the compiler does not see an asm's operands when it places the wait states an MFMA
                //  result needs - measured: wrong values.  Adding the bias FIRST gives canonical operands - four adds, v_max3, v_max - and the
                //  same bits, rounding being monotonic; leaky as max(v, 0.1 v) == (v < 0 ? 0.1 v : v).  The kernel issues ~450 VALU
                //  instructions per tile and wavefront at four wavefronts per SIMD: since its loads stopped being fetched three times over,
                //  that issue stream is what it runs at.)
                float v = fmaxf(fmaxf(acc[4 * g] + bv, acc[4 * g + 1] + bv), fmaxf(acc[4 * g + 2] + bv, acc[4 * g + 3] + bv));
                v = fmaxf(v, v * 0.1f);
                if constexpr (SPLIT) {
                    _Float16 vh, vl;
                    split_f32(v, vh, vl);
                    otile[prow * 16 + chalf * 8 + 2 * g + h][n] = vh;
                    otile[128 + prow * 16 + chalf * 8 + 2 * g + h][n] = vl;
                } else
                    otile[prow * 16 + chalf * 8 + 2 * g + h][n] = (_Float16)v;
            }
        }
#endif
        __syncthreads();   // the pooled tile is complete; nobody reads the patch any more
        if constexpr (SPLIT) {
            // items of 128 halves, parts [hi | lo | hi] of 32 channels: 128 pooled pixels x 3 parts x 4 chunks = 1536 16-byte stores, six per thread
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int e = tid + it * 256, pp = e / 12, rest = e - pp * 12, part = rest >> 2, ck = rest & 3;
                const int oy = ty0 / 2 + (pp >> 4), ox = tx0 / 2 + (pp & 15);
                *reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * 128 + part * 32 + ck * 8) =
                    *reinterpret_cast<const half8_t *>(&otile[(part == 1 ? 128 : 0) + pp][ck * 8]);
            }
            return;
        }
        // 128 pooled pixels x 4 chunks of 8 channels = 512 16-byte stores, two per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + it * 256, pp = e >> 2, ck = e & 3;
            const int oy = ty0 / 2 + (pp >> 4), ox = tx0 / 2 + (pp & 15);
#if (Y2_C0_ABL & 1)
            if (oy < 0)                  // diagnostic: no global stores
#endif
            *reinterpret_cast<half8_t *>(out + ((size_t)kLead + (size_t)b * oPL + (size_t)(oy + 1) * oWp + ox) * 32 + ck * 8) =
                *reinterpret_cast<const half8_t *>(&otile[pp][ck * 8]);
        }
    };
    for (int tile = wg; tile < n_tile_total; tile += 2 * G) {
        do_tile(tile, pvA);
        if (tile + G < n_tile_total) do_tile(tile + G, pvB);
    }
}

// (Round 4 also built a second form with the im2col expansion on the B side - an MFMA row = four adjacent conv pixels, K = (channel,
//  patch row, 8-column window), so that a fragment is two ds_read_b64 instead of eight ds_read_u16, 20 MFMAs per block instead of 8:
//  0.241 against 0.231 ms per 128 frames, i.e. NO gain - the kernel was waiting for its global loads, not for its gathers; removed.
//  What paid was the staging above: 0.237 -> 0.189 ms.)

// layer-0 weights for k_conv0_pool_f16: weights_reorg fp32 (C=3, N=32, 3x3) -> w0[k = c*9 + tap][n]
__global__ void k_pack_w0_f32(const float *__restrict__ src, const float *__restrict__ bias_src, float *__restrict__ w0,
                              float *__restrict__ b0)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t < 32) b0[t] = bias_src[t];
    if (t >= 27 * 32) return;
    const int n = t & 31, k = t >> 5, c = k / 9, tap = k - c * 9;
    // block (m0 = 0, n0 = 0): [tap][32][3]
    w0[t] = src[(long)tap * 32 * 3 + n * 3 + c];
}

}  // namespace y2
