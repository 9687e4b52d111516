// kernels_int16.hpp -- hand-written gfx950 kernels for the int16 fixed-point YOLOv2 path.
//
// What is computed (bit-exact target, SURVEY.md 8a "equivalent per-output definition",
// restating hls/core/core_compute.cpp:22-120 + :175-210):
//   acc = shift(bias[m], Qb-Qa_out)                               (not saturated)
//   for each 4-input-channel group, for each tap (i,j):
//       p   = sum_{t<4} w[m][4g+t][i][j] * in[4g+t][y+i-pad][x+j-pad]
//       acc = sat16(acc + shift_round(p, Qa_in+Qw-Qa_out))
//   out = leaky ? (acc<0 ? acc/10 : acc) : acc
// The per-(group,tap) round+saturate makes this an integer-VALU problem, not a GEMM: there is
// no int16 MFMA on CDNA4 and a wider contraction would change results.  The kernel is built
// around the per-(4 channels x 1 tap x 1 output) step (forms A/B below) with every operand
// already in registers:
//   * weights are wave-uniform -> scalar loads (s_load_dwordx16 per tap), SGPR operands;
//   * the input tile of one channel group is staged once per workgroup in LDS and shared by
//     the 4 wavefronts, each of which owns 8 of the workgroup's 32 output channels;
//   * lanes own consecutive pixels (conflict-free ds_read_b64, coalesced 8-byte stores),
//     P pixels per lane x 8 channels = 8P int32 accumulators in VGPRs.
#pragma once

#include "conv_common.hpp"
#include "layout.hpp"

namespace y2 {


// ---- the requantise-and-saturate step, 32-bit forms ------------------------------------------
//
// Issue cost on gfx950 (profiles/r01_ubench_valu_issue_cost.txt): v_dot2*, shifts, v_med3,
// v_and_or take 4 SIMD cycles per wave64; v_add_u32 (VGPR operands), v_mov, v_and/or take 2.
//
// Form A (valid whenever no intermediate leaves int32): t = (dot4(w,x) + round) >> shift,
// acc = med3(acc + t).  hipcc's sdot2 builtin selects the VOP2 v_dot2c form, which accumulates in
// place, so the rounding constant is seeded with a (2-cycle) v_mov: 2+4+4+4+2+4 = 20 cycles/step.
// The VOP3P v_dot2_i32_i16 would take the constant as src2 (18 cycles) but is only reachable
// through inline asm, which hipcc can neither pad for the DOT->VALU wait states of gfx950 nor
// schedule -- it spilled hundreds of VGPRs around such blocks -- so the builtin form is used.
__device__ __forceinline__ int stepA(int acc, const int2 x, const int2 w, const int r, const int s)
{
    int d = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.x), __builtin_bit_cast(short2_t, w.x), r, false);
    d = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.y), __builtin_bit_cast(short2_t, w.y), d, false);
    return clamp16(acc + (d >> s));
}

// Form C (3.5 instructions = 14 cycles per step, tightest bound): two int16 accumulators share a
// VGPR and are updated by ONE saturating packed add, v_pk_add_i16 clamp == sat16(acc + t) for both
// halves -- legal when every t = (p + round) >> s and the shifted bias fit int16, which the host
// proves from the weights.  The shift of the second value writes the upper half of the first's
// register directly (SDWA dst_sel:WORD_1, dst_unused:UNUSED_PRESERVE), so packing costs nothing.
// Hand-ordered for gfx950's software-visible hazards (hipcc pads nothing inside asm):
//   DOT result -> non-DOT reader: 3 instructions in between; sub-dword (SDWA) write -> reader: 1;
//   DOT -> same-opcode DOT through src2: 0.
// One statement = 4 output channels (2 packed accumulators) x 1 (pixel, tap).
__device__ __forceinline__ void stepC4(int &acc01, int &acc23, const int2 x, const int2 w0, const int2 w1, const int2 w2,
                                       const int2 w3, const int r, const int s)
{
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %2, %6, %14, %16\n\t"
        "v_dot2_i32_i16 %3, %7, %14, %16\n\t"
        "v_dot2_i32_i16 %4, %8, %14, %16\n\t"
        "v_dot2_i32_i16 %5, %9, %14, %16\n\t"
        "v_dot2_i32_i16 %2, %10, %15, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %15, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %15, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %15, %5\n\t"
        "v_ashrrev_i32 %2, %17, %2\n\t"
        "v_ashrrev_i32_sdwa %2, %17, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32 %4, %17, %4\n\t"
        "v_ashrrev_i32_sdwa %4, %17, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_pk_add_i16 %0, %0, %2 clamp\n\t"
        "v_pk_add_i16 %1, %1, %4 clamp"
        : "+v"(acc01), "+v"(acc23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "s"(w0.x), "s"(w1.x), "s"(w2.x), "s"(w3.x), "s"(w0.y), "s"(w1.y), "s"(w2.y), "s"(w3.y), "v"(x.x), "v"(x.y),
          "v"(r), "s"(s));
}

// Form D (3 instructions = 12 cycles per step): form C with the shift folded into the weights.
// If a block's weights leave k = 16 - s bits of int16 headroom, the loader stores them as w * 2^k;
// then p' = p * 2^k and  (p' + 2^15) >> 16  ==  (p + 2^(s-1)) >> s  exactly, i.e. every increment t
// is simply the HIGH half of the dot result.  One v_perm_b32 gathers the high halves of two
// channels into a packed pair (replacing two shifts) and v_pk_add_i16 clamp accumulates it.
// Hazards as in form C (DOT result -> non-DOT reader: 3 instructions in between), which is why one
// statement covers 8 channels: every v_perm is at least 3 instructions behind its second operand's
// last dot.  `sel` = 0x07060302: bytes 2,3 of src1 (even channel) then bytes 2,3 of src0 (odd channel).
// (Two asm statements because one may name at most 30 operands; hipcc can only insert instructions
// between them, which lengthens the spacing.)
__device__ __forceinline__ void stepD8(int &acc01, int &acc23, int &acc45, int &acc67, const int2 x, const int2 *w,
                                       const int r, const int sel)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %0, %8, %24, %26\n\t"
        "v_dot2_i32_i16 %1, %9, %24, %26\n\t"
        "v_dot2_i32_i16 %2, %10, %24, %26\n\t"
        "v_dot2_i32_i16 %3, %11, %24, %26\n\t"
        "v_dot2_i32_i16 %4, %12, %24, %26\n\t"
        "v_dot2_i32_i16 %5, %13, %24, %26\n\t"
        "v_dot2_i32_i16 %6, %14, %24, %26\n\t"
        "v_dot2_i32_i16 %7, %15, %24, %26\n\t"
        "v_dot2_i32_i16 %0, %16, %25, %0\n\t"
        "v_dot2_i32_i16 %1, %17, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %25, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %25, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %25, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %25, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %25, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %25, %7"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "s"(w[0].x), "s"(w[1].x), "s"(w[2].x), "s"(w[3].x), "s"(w[4].x), "s"(w[5].x), "s"(w[6].x), "s"(w[7].x),
          "s"(w[0].y), "s"(w[1].y), "s"(w[2].y), "s"(w[3].y), "s"(w[4].y), "s"(w[5].y), "s"(w[6].y), "s"(w[7].y),
          "v"(x.x), "v"(x.y), "v"(r));
    asm("v_perm_b32 %4, %5, %4, %12\n\t"
        "v_perm_b32 %6, %7, %6, %12\n\t"
        "v_perm_b32 %8, %9, %8, %12\n\t"
        "v_perm_b32 %10, %11, %10, %12\n\t"
        "v_pk_add_i16 %0, %0, %4 clamp\n\t"
        "v_pk_add_i16 %1, %1, %6 clamp\n\t"
        "v_pk_add_i16 %2, %2, %8 clamp\n\t"
        "v_pk_add_i16 %3, %3, %10 clamp"
        : "+v"(acc01), "+v"(acc23), "+v"(acc45), "+v"(acc67), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5),
          "+v"(t6), "+v"(t7)
        : "v"(sel));
}

// Form D without the v_perm ("MODE 5"): every channel keeps its accumulator in the HIGH half of a register of its own and takes
// its increment straight from the high half of its dot result with one v_pk_add_i16 clamp (the low halves add up the dot results'
// fraction bits with saturation - never read).  Same instruction count as form D (16 dots + 8 packed adds instead of 16 + 4 + 4)
// and twice the accumulator registers, but no v_perm and no perm -> add dependency: tools/ubench_step.hip measures 12.75 against
// 13.62 issue cycles per step for the bare sequences at eight wavefronts per SIMD (profiles/r03_ubench_step.txt).  Same weights,
// same legality and the same bits as form D; the host picks per layer by timing / the plan table.
__device__ __forceinline__ void stepE8(int (&acc)[8], const int2 x, const int2 *w, const int r)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %0, %8, %24, %26\n\t"
        "v_dot2_i32_i16 %1, %9, %24, %26\n\t"
        "v_dot2_i32_i16 %2, %10, %24, %26\n\t"
        "v_dot2_i32_i16 %3, %11, %24, %26\n\t"
        "v_dot2_i32_i16 %4, %12, %24, %26\n\t"
        "v_dot2_i32_i16 %5, %13, %24, %26\n\t"
        "v_dot2_i32_i16 %6, %14, %24, %26\n\t"
        "v_dot2_i32_i16 %7, %15, %24, %26\n\t"
        "v_dot2_i32_i16 %0, %16, %25, %0\n\t"
        "v_dot2_i32_i16 %1, %17, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %25, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %25, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %25, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %25, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %25, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %25, %7"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "s"(w[0].x), "s"(w[1].x), "s"(w[2].x), "s"(w[3].x), "s"(w[4].x), "s"(w[5].x), "s"(w[6].x), "s"(w[7].x),
          "s"(w[0].y), "s"(w[1].y), "s"(w[2].y), "s"(w[3].y), "s"(w[4].y), "s"(w[5].y), "s"(w[6].y), "s"(w[7].y),
          "v"(x.x), "v"(x.y), "v"(r));
    // (DOT result -> non-DOT reader needs 3 instructions in between: t0's last dot is 8 instructions back, t7's 7 by the time it is read)
    asm("v_pk_add_i16 %0, %0, %8 clamp\n\t"
        "v_pk_add_i16 %1, %1, %9 clamp\n\t"
        "v_pk_add_i16 %2, %2, %10 clamp\n\t"
        "v_pk_add_i16 %3, %3, %11 clamp\n\t"
        "v_pk_add_i16 %4, %4, %12 clamp\n\t"
        "v_pk_add_i16 %5, %5, %13 clamp\n\t"
        "v_pk_add_i16 %6, %6, %14 clamp\n\t"
        "v_pk_add_i16 %7, %7, %15 clamp"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7));
}

// Form B (4 instructions per step, needs the tighter bound checked by the host): keep the
// accumulator pre-shifted, Bv = acc*2^s + round.  Then
//     Bv' = clamp( ((Bv + p) & ~(2^s-1)) | round )        with bounds  {-32768,32767}*2^s + round
// equals (sat16(acc + ((p + round) >> s)))*2^s + round exactly (floor division distributes over the
// multiple-of-2^s part), and p is accumulated straight into Bv by two v_dot2c: no seed, no shift.
__device__ __forceinline__ int stepB(int Bv, const int2 x, const int2 w, const int nmask, const int r, const int lo, const int hi)
{
    Bv = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.x), __builtin_bit_cast(short2_t, w.x), Bv, false);
    Bv = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.y), __builtin_bit_cast(short2_t, w.y), Bv, false);
    Bv = (Bv & nmask) | r;
    // min(max()) only folds to v_med3_i32 when hipcc can prove lo <= hi; say it directly.
    // (plain VALU -> VALU: no wait states involved; lo in an SGPR, hi in a VGPR)
    int o;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(Bv), "s"(lo), "v"(hi));
    return o;
}


__device__ __forceinline__ long step64(long acc, int2 x, int2 w, const ConvArgs &a)
{
    const int x0 = (short)(x.x & 0xffff), x1 = x.x >> 16, x2 = (short)(x.y & 0xffff), x3 = x.y >> 16;
    const int w0 = (short)(w.x & 0xffff), w1 = w.x >> 16, w2 = (short)(w.y & 0xffff), w3 = w.y >> 16;
    // each int16 x int16 product fits int32; widen once per product like core_compute.cpp:102-106
    const long p = (long)(w0 * x0) + (long)(w1 * x1) + (long)(w2 * x2) + (long)(w3 * x3);
    long v = acc + shift64(p, a.sh_right, a.sh_left, a.shift, a.sh_right && a.shift > 0 ? (1L << (a.shift - 1)) : 0);
    v = v > 32767 ? 32767 : v;
    v = v < -32768 ? -32768 : v;
    return v;
}


// Conv KSxKS, stride 1, 'same' padding, on the item layout.
//   grid.x = ceil(npix / (64*P)) pixel tiles, grid.y = output-channel blocks of 32
//   block  = 256 threads = 4 wavefronts; wavefront w owns channels [32*by + 8w, +8)
// MODE 0: 32-bit form A, valid when the host proved no intermediate leaves int32.
// MODE 1: 32-bit form B (pre-shifted accumulator), tighter bound, 4 instructions per step.
// MODE 2: 64-bit path, any Q / any weights (reference arithmetic verbatim).
// MODE 3: 32-bit form C (packed int16 accumulators, saturating packed add), 3.5 instructions per step.
// MODE 4: form D = form C on weights pre-scaled by 2^(16-s): shift-free, 3 instructions per step.
// MODE 5: form D with one accumulator register per channel (value in the high half) and no v_perm (stepE8).
// NST: staging registers per thread, 256*NST >= LDS tile items.
// GRP (1x1 convs only): channel groups staged and consumed per barrier.  A 1x1 conv has one tap per
//      group, i.e. only 8*P steps between barriers; with GRP = 8 the loop body looks like a 3x3
//      group (the weight slices of consecutive groups are contiguous, exactly like taps).

#ifndef PREX_MAX_P
#define PREX_MAX_P 2
#endif
#ifndef Y2_I16_ABL
#define Y2_I16_ABL 0   // diagnostic builds only (tools/build_variant.sh): 1 = every group re-reads group 0's weight slice (scalar-cache hits:
                       // what the per-tap scalar loads cost), 2 = no input staging inside the loop, 4 = no workgroup barrier inside the loop,
                       // 8 = weights loaded once per workgroup (no scalar loads in the loop), 16 = one input item per lane (no LDS reads in the loop).
                       // Results are wrong with any of them; only the timing is of interest.
#endif
template <int KS, int P, int MODE, int NST, int GRP = 1>
__global__ __launch_bounds__(256, 2) void k_conv_i16(const int2 *__restrict__ in, int2 *__restrict__ out,
                                                   const int2 *__restrict__ wpk,
                                                   const short *__restrict__ bias, const ConvArgs a)
{
    extern __shared__ int2 lds[];
    constexpr bool X64 = MODE == 2;
    typedef typename std::conditional<X64, long, int>::type acc_t;
    constexpr int T = 64 * P;
    constexpr int KT = KS * KS;         // spatial taps per channel group
    constexpr int KK = KT * GRP;        // "taps" per barrier (taps of GRP consecutive channel groups)
    constexpr int PFG = KS == 1 ? GRP : 1;   // channel groups whose input values are fetched from LDS in one batch
    static_assert(GRP == 1 || KS == 1 || (KS == 3 && GRP == 2), "grouping: 1x1 convs (8 groups per barrier); 3x3 with 2 groups works but measured slower, not launched");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x, yb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, yb);
    const int mb = a.mb_list ? a.mb_list[yb] : yb;
    const int q0 = tile * T;
    const int qlast = min(q0 + T, a.npix) - 1;
    const int halo = (KS == 3) ? a.Wp + 1 : 0;
    const int fmin = flat_of(a, q0);
    const int fmax = flat_of(a, qlast);
    const int tile_start = fmin - halo;
    const int Lt = min(fmax - fmin + 1 + 2 * halo, a.lt_max);

    int fo[P];        // flat item offset of each owned pixel (also its output address)
    int rowaddr[P][KS];  // LDS byte address of (row i-1, col -1) relative tap origin per pixel
    bool valid[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = q0 + p * 64 + lane;
        valid[p] = q <= qlast;
        fo[p] = flat_of(a, min(q, qlast));
        const int lo = fo[p] - tile_start;
#pragma unroll
        for (int i = 0; i < KS; ++i)
            rowaddr[p][i] = (KS == 3) ? (lo + (i - 1) * a.Wp - 1) * 8 : lo * 8;
    }

    // bias moved to the Qa_out domain, not saturated (core_compute.cpp:86-97)
    acc_t acc[P][8];
    {
        const short *bp = bias + mb * 32 + wave * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc_t b0;
            if (X64) {
                b0 = (acc_t)shift64((long)bp[m], a.bs_right, a.bs_left, a.bs_mag,
                                    a.bs_right && a.bs_mag > 0 ? (1L << (a.bs_mag - 1)) : 0);
            } else {
                const int b = bp[m];
                b0 = a.bs_right ? (acc_t)((b + (a.bs_mag > 0 ? (1 << (a.bs_mag - 1)) : 0)) >> a.bs_mag)
                                : (a.bs_left ? (acc_t)(b << a.bs_mag) : (acc_t)b);
                if (MODE == 1) b0 = (acc_t)(((int)b0 << a.shift) + a.round);
            }
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p][m] = b0;
        }
        if (MODE == 5) {   // value in the high half (the host proved |bias0| <= 32767)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[p][m] = (acc_t)((int)acc[p][m] << 16);
        }
        if (MODE == 3 || MODE == 4) {  // pack channel pairs (2j, 2j+1) into acc[p][j]; the host proved |bias0| <= 32767
            int pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[j] = ((int)acc[0][2 * j] & 0xffff) | ((int)acc[0][2 * j + 1] << 16);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] = (acc_t)pk[j];
        }
    }

    const int r = a.round, s = a.shift;
    const int nmask = ~((1 << s) - 1), lo_b = (int)(0xffff8000u << s) + r, hi_b = (32767 << s) + r;
    // Form B wants v_and_or_b32 (mask in an SGPR, round in a VGPR: one constant-bus operand);
    // hide the uniformity of `round` so hipcc keeps it in a VGPR instead of splitting and/or.
    int r_vgpr = r;
    if (MODE != 2) asm volatile("" : "+v"(r_vgpr));
    const char *lds_b = reinterpret_cast<const char *>(lds);
    const int2 *src = in + kLead + tile_start;
    const int2 *wq = wpk + ((long)mb * a.CGin * KT * 32 + wave * 8);   // [mb][cg][tap][32]: KT taps per group

    // Input tiles are double-buffered in LDS: the global loads of group cg+1 are issued before the
    // compute on group cg and written to the other buffer after it, so HBM/L2 latency hides behind
    // ~KK*P*8 requant steps and there is ONE barrier per channel group.
    int2 stage[NST];
    const int buf_items = a.lt_max * GRP;
    // staging index i -> (group g, item j): g = i / Lt for GRP > 1 (one division per staged item, outside the step loop)
    auto src_off = [&](int i) -> long { if (GRP == 1) return i; const int g = i / Lt; return (long)g * a.in_cg_stride + (i - g * Lt); };
    auto lds_off = [&](int i) -> int { if (GRP == 1) return i; const int g = i / Lt; return g * a.lt_max + (i - g * Lt); };
    const int LtG = Lt * GRP;
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < LtG) stage[k] = src[src_off(i)];
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < LtG) lds[lds_off(i)] = stage[k];
    }
    __syncthreads();

    int chain = 0;
#if (Y2_I16_ABL & 8)
    int2 abl_w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) abl_w[m] = wq[m];
#endif
#if (Y2_I16_ABL & 16)
    int2 abl_x = lds[tid & 63];
#endif
    const int niter = a.CGin / GRP;   // the host only selects GRP > 1 when it divides CGin
    for (int cg = 0; cg < niter; ++cg) {
        // branch-free: the last group re-fetches its own tile instead of testing `cg + 1 < niter`
        src += (cg + 1 < niter) ? a.in_cg_stride * GRP : 0;
#if !(Y2_I16_ABL & 2)
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 256;
            if (i < LtG) stage[k] = src[src_off(i)];
        }
#endif
        const char *tile = lds_b + (cg & 1) * buf_items * 8;
        // Small tiles (P <= 2): fetch the whole group's input values up front, so the LDS latency of a
        // tap is not exposed when few wavefronts are resident (the tail of a launch at modest batch).
        constexpr bool PREX = (MODE == 3 || MODE == 4 || MODE == 5) && P <= PREX_MAX_P;
        // LDS address of "tap" t of pixel p: group t / KT of this barrier interval, spatial tap t % KT
        auto xaddr = [&](int t, int p) -> const int2 * {
            return reinterpret_cast<const int2 *>(tile + (t / KT) * (a.lt_max * 8) + rowaddr[p][(t % KT) / KS] + ((t % KT) % KS) * 8);
        };
#pragma unroll
        for (int gb = 0; gb < GRP / PFG; ++gb) {
        int2 xv[PREX ? KT * PFG : 1][P];
        if (PREX && !(Y2_I16_ABL & 16)) {
#pragma unroll
            for (int tt = 0; tt < KT * PFG; ++tt)
#pragma unroll
                for (int p = 0; p < P; ++p) xv[tt][p] = *xaddr(gb * KT * PFG + tt, p);
        }
#pragma unroll
        for (int tt = 0; tt < KT * PFG; ++tt) {
            const int tap = gb * KT * PFG + tt;
            int2 w[8];
#if (Y2_I16_ABL & 8)
#pragma unroll
            for (int m = 0; m < 8; ++m) { w[m] = abl_w[m]; asm volatile("" : "+s"(w[m].x), "+s"(w[m].y)); }
#else
#pragma unroll
            for (int m = 0; m < 8; ++m) w[m] = wq[tap * 32 + m];  // wave-uniform: scalar loads
#endif
#pragma unroll
            for (int p = 0; p < P; ++p) {
                int2 x;
#if (Y2_I16_ABL & 16)
                x = abl_x; asm volatile("" : "+v"(x.x), "+v"(x.y));
#else
                if (PREX) x = xv[tt][p];
                else x = *xaddr(tap, p);
#endif
                // Form A's dot products do not depend on the accumulators, so hipcc would compute
                // all 72x8 of a group up front (hundreds of live registers, SGPR spills).  An empty
                // asm ties this pixel's x to the previous pixel's last accumulator: same order as
                // form B gets naturally from accumulating into acc.  No instruction is emitted.
                if (MODE == 0) asm volatile("" : "+v"(x.x) : "v"(chain));
                if (MODE == 5) {
                    int ae[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) ae[m] = (int)acc[p][m];
                    stepE8(ae, x, w, r_vgpr);
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[p][m] = (acc_t)ae[m];
                } else if (MODE == 4) {
                    int a01 = (int)acc[p][0], a23 = (int)acc[p][1], a45 = (int)acc[p][2], a67 = (int)acc[p][3];
                    stepD8(a01, a23, a45, a67, x, w, r_vgpr, 0x07060302);
                    acc[p][0] = (acc_t)a01; acc[p][1] = (acc_t)a23; acc[p][2] = (acc_t)a45; acc[p][3] = (acc_t)a67;
                } else if (MODE == 3) {
                    int a01 = (int)acc[p][0], a23 = (int)acc[p][1], a45 = (int)acc[p][2], a67 = (int)acc[p][3];
                    stepC4(a01, a23, x, w[0], w[1], w[2], w[3], r_vgpr, s);
                    stepC4(a45, a67, x, w[4], w[5], w[6], w[7], r_vgpr, s);
                    acc[p][0] = (acc_t)a01; acc[p][1] = (acc_t)a23; acc[p][2] = (acc_t)a45; acc[p][3] = (acc_t)a67;
                } else
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (MODE == 2) acc[p][m] = (acc_t)step64((long)acc[p][m], x, w[m], a);
                    else if (MODE == 1) acc[p][m] = (acc_t)stepB((int)acc[p][m], x, w[m], nmask, r_vgpr, lo_b, hi_b);
                    else acc[p][m] = (acc_t)stepA((int)acc[p][m], x, w[m], r_vgpr, s);
                }
                if (MODE == 0) chain = (int)acc[p][7];
            }
            // form A: without a fence between taps hipcc hoists the scalar weight loads of all
            // nine taps (144 SGPRs) to the top of the group and spills them
            if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
        }
        }
#if !(Y2_I16_ABL & 2)
        {
            int2 *nxt = lds + ((cg + 1) & 1) * buf_items;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = tid + k * 256;
                if (i < LtG) nxt[lds_off(i)] = stage[k];
            }
        }
#endif
#if !(Y2_I16_ABL & 1)
        wq += KK * 32;
#else
        asm volatile("" : "+s"(wq));
#endif
#if !(Y2_I16_ABL & 4)
        __syncthreads();
#endif
    }

    // write-back with integer leaky (core_compute.cpp:175-264): 2 items (8 channels) per pixel
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int cgo = mb * 8 + wave * 2 + g;
        if (cgo >= a.CGout) continue;
        int2 *dst = out + a.out_base + (long)cgo * a.out_cg_stride;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int e = (int)acc[p][g * 4 + t];
                if (MODE == 1) e >>= s;  // back from the pre-shifted domain
                if (MODE == 5) e >>= 16;
                if (MODE == 3 || MODE == 4) {   // unpack channel g*4+t from its pair register
                    const int pr = (int)acc[p][(g * 4 + t) >> 1];
                    e = (t & 1) ? (pr >> 16) : (int)(short)(pr & 0xffff);
                }
                v[t] = a.leaky ? leaky_i16(e) : e;
            }
            int2 o;
            o.x = (v[0] & 0xffff) | (v[1] << 16);
            o.y = (v[2] & 0xffff) | (v[3] << 16);
            if (valid[p]) dst[fo[p]] = o;
        }
    }
}

// ------------------------------------------------------------------ 16 output channels per wavefront (2-wave workgroups)
//
// Ablation of k_conv_i16 at batch 256 (profiles/r03_i16_ablation_b256.txt): removing the barrier changes nothing, removing the
// input staging gains 1 % on the 3x3 layers - what is left between its ~93 % and the issue ceiling is SIMD time of the NON-step
// instructions of a wavefront: above all the ds_read of the input items (9 x P reads of 8 bytes per lane and channel group) and
// the per-tap wait for its scalar weight load, both of which scale with the number of (pixel, tap) pairs a wavefront visits, not
// with the steps it does there.  The conv + pool kernel, whose four pixels share a 4 x 4 patch (16 reads for 36 taps), runs the
// same layer shapes at 95.5 %.  Here a wavefront owns SIXTEEN output channels instead of eight: every input item read from LDS
// and every visit of a tap feeds two 8-channel step sequences (the second with the next 16 SGPRs of the same 32-channel weight
// slice), i.e. half the LDS reads, half the staged bytes per step and twice the work behind every scalar-load wait.  The
// workgroup stays one 32-channel block x 64 P pixels - it is two wavefronts instead of four, sixteen of them per CU - so grids,
// the block-wise arithmetic forms, the XCD numbering and the tile geometry are those of k_conv_i16.  Forms C / D only.
template <int KS, int P, int MODE, int NST>
__global__ __launch_bounds__(128, 4) void k_conv_i16_w16(const int2 *__restrict__ in, int2 *__restrict__ out, const int2 *__restrict__ wpk,
                                                          const short *__restrict__ bias, const ConvArgs a)
{
    static_assert(MODE == 3 || MODE == 4, "packed-accumulator forms only");
    extern __shared__ int2 lds[];
    constexpr int T = 64 * P, KT = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0 / 1: channels [16 wave, 16 wave + 16) of the block
    int tile = blockIdx.x, yb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, yb);
    const int mb = a.mb_list ? a.mb_list[yb] : yb;
    const int q0 = tile * T, qlast = min(q0 + T, a.npix) - 1;
    const int halo = (KS == 3) ? a.Wp + 1 : 0;
    const int fmin = flat_of(a, q0), fmax = flat_of(a, qlast);
    const int tile_start = fmin - halo;
    const int Lt = min(fmax - fmin + 1 + 2 * halo, a.lt_max);

    int fo[P], rowaddr[P][KS];
    bool valid[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = q0 + p * 64 + lane;
        valid[p] = q <= qlast;
        fo[p] = flat_of(a, min(q, qlast));
        const int lo = fo[p] - tile_start;
#pragma unroll
        for (int i = 0; i < KS; ++i) rowaddr[p][i] = (KS == 3) ? (lo + (i - 1) * a.Wp - 1) * 8 : lo * 8;
    }
    int acc[P][8];    // [pixel][channel pair]: pairs 0-3 = channels 0-7 of this wavefront, pairs 4-7 = channels 8-15
    {
        const short *bp = bias + mb * 32 + wave * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int b2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int b = bp[2 * j + h];
                b2[h] = a.bs_right ? ((b + (a.bs_mag > 0 ? (1 << (a.bs_mag - 1)) : 0)) >> a.bs_mag) : (a.bs_left ? (b << a.bs_mag) : b);
            }
            const int pk = (b2[0] & 0xffff) | (b2[1] << 16);   // the host proved |shifted bias| <= 32767
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p][j] = pk;
        }
    }
    int r_vgpr = a.round;
    asm volatile("" : "+v"(r_vgpr));
    const int s = a.shift;
    const char *lds_b = reinterpret_cast<const char *>(lds);
    const int2 *src = in + kLead + tile_start;
    const int2 *wq = wpk + ((long)mb * a.CGin * KT * 32 + wave * 16);   // [mb][cg][tap][32]

    int2 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 128;
        if (i < Lt) stage[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 128;
        if (i < Lt) lds[i] = stage[k];
    }
    __syncthreads();

    for (int cg = 0; cg < a.CGin; ++cg) {
        src += (cg + 1 < a.CGin) ? a.in_cg_stride : 0;     // branch-free: the last group re-fetches its own tile
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 128;
            if (i < Lt) stage[k] = src[i];
        }
        const char *tl = lds_b + (cg & 1) * a.lt_max * 8;
        int2 xv[KT][P];
#pragma unroll
        for (int tt = 0; tt < KT; ++tt)
#pragma unroll
            for (int p = 0; p < P; ++p) xv[tt][p] = *reinterpret_cast<const int2 *>(tl + rowaddr[p][tt / KS] + (tt % KS) * 8);
#pragma unroll
        for (int tt = 0; tt < KT; ++tt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {       // the two 8-channel halves take their 16 SGPRs one after the other
                int2 w[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) w[m] = wq[tt * 32 + h * 8 + m];   // wave-uniform: scalar loads
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (MODE == 4) {
                        stepD8(acc[p][4 * h], acc[p][4 * h + 1], acc[p][4 * h + 2], acc[p][4 * h + 3], xv[tt][p], w, r_vgpr, 0x07060302);
                    } else {
                        stepC4(acc[p][4 * h], acc[p][4 * h + 1], xv[tt][p], w[0], w[1], w[2], w[3], r_vgpr, s);
                        stepC4(acc[p][4 * h + 2], acc[p][4 * h + 3], xv[tt][p], w[4], w[5], w[6], w[7], r_vgpr, s);
                    }
                }
            }
        }
        {
            int2 *nxt = lds + ((cg + 1) & 1) * a.lt_max;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = tid + k * 128;
                if (i < Lt) nxt[i] = stage[k];
            }
        }
        wq += KT * 32;
        __syncthreads();
    }

    // write-back with integer leaky: 4 items (16 channels) per pixel
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cgo = mb * 8 + wave * 4 + g;
        if (cgo >= a.CGout) continue;
        int2 *dst = out + a.out_base + (long)cgo * a.out_cg_stride;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int p01 = acc[p][2 * g], p23 = acc[p][2 * g + 1];
            int v[4] = {(int)(short)(p01 & 0xffff), p01 >> 16, (int)(short)(p23 & 0xffff), p23 >> 16};
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = a.leaky ? leaky_i16(v[t]) : v[t];
            int2 o;
            o.x = (v[0] & 0xffff) | (v[1] << 16);
            o.y = (v[2] & 0xffff) | (v[3] << 16);
            if (valid[p]) dst[fo[p]] = o;
        }
    }
}

// ------------------------------------------------------------------ conv 3x3 + leaky + 2x2/2 max pool, fused
//
// Five of the 3x3 convs (layers 0, 2, 6, 10, 16) feed a 2x2 stride-2 max pool (pool_yolo2, core_compute.cpp:266-305),
// and for four of them the pool is the ONLY consumer: the full-resolution tensor need never reach memory.  Here a lane
// owns one pool WINDOW = 2 x 2 conv outputs (x 8 channels per wavefront), computes them with exactly the step sequence
// of k_conv_i16 (same order group -> tap, same saturating chain: bit-exact) and stores max(leaky(.)) - the integer leaky
// is monotonic, so max and leaky commute exactly.  What changes around the steps:
//   * a tile = 64 consecutive windows in raster order over (b, oy, ox); the input of all their pixels (+ halo) is still
//     ONE contiguous run of the flat item layout (whole rows between the first window's top-left halo and the last
//     window's bottom-right halo), so staging is unchanged;
//   * the four pixels of a window share a 4 x 4 input patch: 16 ds_read_b64 per channel group feed 36 (pixel, tap) steps
//     (the 1-pixel-per-lane form reads 9 per 9), fetched ahead of the group's tap loop;
//   * FULL (layer 16, which also feeds the route to layer 26): the full-resolution tensor is stored as well;
//   * SINGLE (one input channel group: layer 0): nothing is staged inside the loop, so the staging registers die
//     after the prologue.
// MODE 3 / 4 only (packed int16 accumulators); layers whose blocks need another form keep conv + k_maxpool2.
__device__ __forceinline__ int pkmax16(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b)));
}

template <int MODE, int NST, bool FULL, bool SINGLE = false>
__global__ __launch_bounds__(256) void k_conv_i16_pool(const int2 *__restrict__ in, int2 *__restrict__ out_full,
                                                        int2 *__restrict__ out_pool, const int2 *__restrict__ wpk,
                                                        const short *__restrict__ bias, const ConvArgs a)
{
    static_assert(MODE == 3 || MODE == 4, "packed-accumulator forms only");
    extern __shared__ int2 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x, yb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, yb);
    const int mb = a.mb_list ? a.mb_list[yb] : yb;
    const int OW = a.W >> 1, OHW = (a.H >> 1) * OW;
    // flat item offset of the top-left pixel of window wq
    auto win_tl = [&](int wq, int &b, int &oy, int &ox) -> int {
        b = div_c(wq, a.mOHW, a.sOHW);
        const int r = wq - b * OHW;
        oy = div_c(r, a.mOW, a.sOW);
        ox = r - oy * OW;
        return b * a.PL + (2 * oy + 1) * a.Wp + 2 * ox;
    };
    const int w0 = tile * 64, wlast = min(w0 + 63, a.nwin - 1);
    int tb, toy, tox;
    const int f_first = win_tl(w0, tb, toy, tox);
    const int f_last = win_tl(wlast, tb, toy, tox) + a.Wp + 1;      // bottom-right pixel of the last window
    const int tile_start = f_first - a.Wp - 1;
    const int Lt = min(f_last + a.Wp + 1 - tile_start + 1, a.lt_max);

    const int wq = w0 + lane;
    const bool valid = wq <= wlast;
    int wb, woy, wox;
    const int ftl = win_tl(min(wq, wlast), wb, woy, wox);
    // LDS byte address of patch row r (image rows 2oy-1 .. 2oy+2), column 2ox-1
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (ftl - tile_start + (r - 1) * a.Wp - 1) * 8;

    int acc[4][4];   // [pixel dy*2+dx][channel pair]
    {
        const short *bp = bias + mb * 32 + wave * 8;
        int pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int b2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int b = bp[2 * j + h];
                b2[h] = a.bs_right ? ((b + (a.bs_mag > 0 ? (1 << (a.bs_mag - 1)) : 0)) >> a.bs_mag) : (a.bs_left ? (b << a.bs_mag) : b);
            }
            pk[j] = (b2[0] & 0xffff) | (b2[1] << 16);   // the host proved |shifted bias| <= 32767
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[p][j] = pk[j];
    }
    int r_vgpr = a.round;
    asm volatile("" : "+v"(r_vgpr));
    const int s = a.shift;
    const char *lds_b = reinterpret_cast<const char *>(lds);
    const int2 *src = in + kLead + tile_start;
    const int2 *wq_p = wpk + ((long)mb * a.CGin * 9 * 32 + wave * 8);

    int2 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < Lt) stage[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < Lt) lds[i] = stage[k];
    }
    __syncthreads();

    for (int cg = 0; cg < a.CGin; ++cg) {
        const bool more = !SINGLE && cg + 1 < a.CGin;     // wave-uniform
        if (more) {
            src += a.in_cg_stride;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = tid + k * 256;
                if (i < Lt) stage[k] = src[i];
            }
        }
        const char *tl = lds_b + (cg & 1) * a.lt_max * 8;
        int2 patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) patch[r][c] = *reinterpret_cast<const int2 *>(tl + prow[r] + c * 8);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            int2 w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) w[m] = wq_p[tap * 32 + m];   // wave-uniform: scalar loads
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int2 x = patch[(p >> 1) + tap / 3][(p & 1) + tap % 3];
                if (MODE == 4) {
                    stepD8(acc[p][0], acc[p][1], acc[p][2], acc[p][3], x, w, r_vgpr, 0x07060302);
                } else {
                    stepC4(acc[p][0], acc[p][1], x, w[0], w[1], w[2], w[3], r_vgpr, s);
                    stepC4(acc[p][2], acc[p][3], x, w[4], w[5], w[6], w[7], r_vgpr, s);
                }
            }
        }
        if (more) {
            int2 *nxt = lds + ((cg + 1) & 1) * a.lt_max;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = tid + k * 256;
                if (i < Lt) nxt[i] = stage[k];
            }
        }
        wq_p += 9 * 32;
        __syncthreads();
    }

    // epilogue: max over the window on packed pairs, then the integer leaky (monotonic: commutes with max)
    int mx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mx[j] = pkmax16(pkmax16(acc[0][j], acc[1][j]), pkmax16(acc[2][j], acc[3][j]));
    const int fpool = wb * a.oPL + (woy + 1) * a.oWp + wox;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int cgo = mb * 8 + wave * 2 + g;
        if (cgo >= a.CGout) continue;
        auto pack4 = [&](int p01, int p23) -> int2 {
            int v[4] = {(int)(short)(p01 & 0xffff), p01 >> 16, (int)(short)(p23 & 0xffff), p23 >> 16};
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = a.leaky ? leaky_i16(v[t]) : v[t];
            int2 o;
            o.x = (v[0] & 0xffff) | (v[1] << 16);
            o.y = (v[2] & 0xffff) | (v[3] << 16);
            return o;
        };
        if (valid) out_pool[a.pool_base + (long)cgo * a.pool_cg_stride + fpool] = pack4(mx[2 * g], mx[2 * g + 1]);
        if (FULL) {
            int2 *dst = out_full + a.out_base + (long)cgo * a.out_cg_stride;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (valid) dst[ftl + (p >> 1) * a.Wp + (p & 1)] = pack4(acc[p][2 * g], acc[p][2 * g + 1]);
        }
    }
}

// ------------------------------------------------------------------ split-K variant (small batches)
//
// With one frame, a 13x13 layer has 169 pixels: the tiled kernel above leaves most SIMDs idle and
// every output is a serial chain of CGin*KK saturating steps (up to 2880).  The chain can still be
// split bit-exactly: each step is the map x -> clamp(x + t, -32768, 32767), and such clamp-affine
// maps  f(x) = clamp(x + a, l, h)  are closed under composition:
//     (f2 o f1)(x) = clamp(x + a1 + a2, clamp(l1 + a2, l2, h2), clamp(h1 + a2, l2, h2)).
// A wavefront here is T = 64/S pixels x S K-splits (S = 4, or 8 on 1x1 layers; lane = split*T + pixel,
// optionally PP = 2 or 4 pixels per lane): split s runs channel groups [s*CGin/S, (s+1)*CGin/S) and
// carries the triple (a, l, h) of its sub-chain per output channel instead of a value; at the end the S
// triples are combined IN ORDER with log2(S) rounds of wavefront shuffles (__shfl_down T, 2T, ..) and
// applied to the shifted bias.  Because the splits use different channel groups, the weights are
// lane-dependent: the S 32-channel weight slices are staged through LDS next to the S input tiles and
// read into VGPRs.
// Legal when the host proved |t| < 2^29 and no int32 overflow (form A bound); CGin % S == 0.
constexpr int kSplitBig = 1 << 29;

// PACK (form D layers, shift = 16): the triples live in packed int16 pairs like form D's accumulators -
// every increment pair is one v_perm_b32 of two dot results, l and h advance with one v_pk_add_i16 clamp
// each and a with a WRAPPING packed add (16 cycles per step instead of ~30).  The true a is recovered at
// the end: for x in [-32768, 32767], f(-32768) = l and f(32767) = h give  h - 32767 <= a <= l + 32768,
// an interval shorter than 65536, so a mod 2^16 determines it (and when l == h, a no longer matters).
// PP (1 or 2): pixels per lane.  The per-iteration cost (stage, barrier, LDS round trip) dominates at one
// frame, so two pixel tiles per wavefront share one staged set of weight slices and input tiles.
template <int KS, int NST, bool PACK = false, int S = 4, int PP = 1>
__global__ __launch_bounds__(256) void k_conv_i16_splitk(const int2 *__restrict__ in, int2 *__restrict__ out,
                                                          const int2 *__restrict__ wpk,
                                                          const short *__restrict__ bias, const ConvArgs a)
{
    extern __shared__ int2 lds[];
    constexpr int KK = KS * KS, WITEMS = KK * 32, T = 64 / S;   // S K-splits x T pixels per wavefront (x PP per lane)
    constexpr int TT = T * PP;                                  // pixels per workgroup tile
    // LDS stride of a weight slice: +4 items, so that the S slices a wavefront reads from at once (same
    // offset, one per split) start 32 bytes apart instead of on the same banks (2304 = 9 x 256 bytes)
    constexpr int WSTR = WITEMS + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = lane / T, pix = lane % T;
    // same XCD grid as the tiled kernel: at one frame the input is tiny and the weights are everything, so
    // the host picks Xm = 8 - all pixel tiles of a channel block run on ONE XCD and its weight slices cross
    // the fabric once instead of once per XCD
    int tile = blockIdx.x, yb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, yb);
    const int mb = a.mb_list ? a.mb_list[yb] : yb;
    const int q0 = tile * TT;
    const int qlast = min(q0 + TT, a.npix) - 1;
    const int halo = (KS == 3) ? a.Wp + 1 : 0;
    const int fmin = flat_of(a, q0), fmax = flat_of(a, qlast);
    const int tile_start = fmin - halo;
    const int Lt = min(fmax - fmin + 1 + 2 * halo, a.lt_max);
    const int Q = a.CGin / S;                               // channel groups per split
    // one LDS buffer = S input tiles (lt_max items each) followed by S weight slices (WSTR each)
    const int buf_items = S * (a.lt_max + WSTR);
    bool valid[PP];
    int fo[PP], rowaddr[PP][KS];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int qp = q0 + pix + p * T;
        valid[p] = (qp <= qlast) && split == 0;             // split-0 lanes hold the combined result
        fo[p] = flat_of(a, min(qp, qlast));
        const int lo = fo[p] - tile_start;
#pragma unroll
        for (int i = 0; i < KS; ++i) rowaddr[p][i] = (split * a.lt_max + ((KS == 3) ? (lo + (i - 1) * a.Wp - 1) : lo)) * 8;
    }
    const int waddr = (S * a.lt_max + split * WSTR + wave * 8) * 8;   // this lane's weight slice, tap 0

    int ta[PP][8], tl[PP][8], th[PP][8];   // the clamp-affine triple of this lane's sub-chain, per pixel and output channel
    short2_t pa[PP][4], pl[PP][4], ph[PP][4];   // PACK: the same triples for channel pairs (2j, 2j+1)
#pragma unroll
    for (int p = 0; p < PP; ++p) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { ta[p][m] = 0; tl[p][m] = -kSplitBig; th[p][m] = kSplitBig; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pa[p][j] = short2_t{0, 0};
            pl[p][j] = short2_t{(short)-32768, (short)-32768};
            ph[p][j] = short2_t{(short)32767, (short)32767};
        }
    }

    const int r = a.round, s = a.shift;
    const int LtS = Lt * S, WS = WITEMS * S, total = LtS + WS;
    // staging item i: i < LtS -> input tile (split g = i / Lt), else weight slice (split g = (i-LtS) / WITEMS)
    const int2 *xsrc = in + kLead + tile_start;                               // group 0 of split 0
    const int2 *wsrc = wpk + (long)mb * a.CGin * KK * 32;                     // this block's 32 channels, group 0
    // the (split, item) decomposition of a staging index does not depend on the iteration: resolve it
    // once per thread (pointer at iteration 0, LDS slot, per-iteration stride)
    const int2 *p0[NST];
    int slot[NST];
    long pstep[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        p0[k] = xsrc; slot[k] = 0; pstep[k] = 0;
        if (i < LtS) {
            const int g = i / Lt, j = i - g * Lt;
            p0[k] = xsrc + (long)g * Q * a.in_cg_stride + j;
            pstep[k] = a.in_cg_stride;
            slot[k] = g * a.lt_max + j;
        } else if (i < total) {
            const int jj = i - LtS, g = jj / WITEMS, j = jj - g * WITEMS;
            p0[k] = wsrc + (long)g * Q * WITEMS + j;
            pstep[k] = WITEMS;
            slot[k] = S * a.lt_max + g * WSTR + j;
        }
    }
    int2 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) { const int i = tid + k * 256; if (i < total) stage[k] = p0[k][0]; }
#pragma unroll
    for (int k = 0; k < NST; ++k) { const int i = tid + k * 256; if (i < total) lds[slot[k]] = stage[k]; }
    __syncthreads();

    const char *lds_b = reinterpret_cast<const char *>(lds);
    // Two iterations of global loads are in flight: at one frame a layer has only one or two workgroups per
    // CU, so nothing else hides the latency of fetching the next tiles and weight slices; the loads of
    // iteration it+2 are issued before the compute of it, those of it+1 (issued one iteration earlier) are
    // committed to the other LDS buffer after it.  Two register sets alternate.
    int2 stage_b[NST];
    auto issue = [&](int2 (&dst)[NST], int itx) {
        const int itc = min(itx, Q - 1);   // branch-free: past the end re-fetch the last iteration
#pragma unroll
        for (int k = 0; k < NST; ++k) { const int i = tid + k * 256; if (i < total) dst[k] = p0[k][(long)itc * pstep[k]]; }
    };
    auto commit = [&](const int2 (&src)[NST], int bufidx) {
        int2 *nxt = lds + (size_t)bufidx * buf_items;
#pragma unroll
        for (int k = 0; k < NST; ++k) { const int i = tid + k * 256; if (i < total) nxt[slot[k]] = src[k]; }
    };
    auto compute = [&](int it) {
        const char *buf = lds_b + (size_t)(it & 1) * buf_items * 8;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            int2 x[PP];
#pragma unroll
            for (int p = 0; p < PP; ++p) x[p] = *reinterpret_cast<const int2 *>(buf + rowaddr[p][tap / KS] + (tap % KS) * 8);
            int2 w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) w[m] = *reinterpret_cast<const int2 *>(buf + waddr + (tap * 32 + m) * 8);
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                if (PACK) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int d0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].x), __builtin_bit_cast(short2_t, w[2 * j].x), r, false);
                        int d1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].x), __builtin_bit_cast(short2_t, w[2 * j + 1].x), r, false);
                        d0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].y), __builtin_bit_cast(short2_t, w[2 * j].y), d0, false);
                        d1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].y), __builtin_bit_cast(short2_t, w[2 * j + 1].y), d1, false);
                        // shift = 16: the increments are the high halves; gather both into one register
                        const short2_t t = __builtin_bit_cast(short2_t, __builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x07060302u));
                        pa[p][j] = pa[p][j] + t;                                  // wraps mod 2^16 per half
                        pl[p][j] = __builtin_elementwise_add_sat(pl[p][j], t);    // sat16(l + t)
                        ph[p][j] = __builtin_elementwise_add_sat(ph[p][j], t);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        int d = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].x), __builtin_bit_cast(short2_t, w[m].x), r, false);
                        d = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x[p].y), __builtin_bit_cast(short2_t, w[m].y), d, false);
                        const int t = d >> s;
                        ta[p][m] += t;
                        tl[p][m] = clamp16(tl[p][m] + t);
                        th[p][m] = clamp16(th[p][m] + t);
                    }
                }
            }
        }
    };
    issue(stage, 1);
    for (int it = 0; it < Q; it += 2) {
        issue(stage_b, it + 2);
        compute(it);
        commit(stage, (it + 1) & 1);
        __syncthreads();
        if (it + 1 < Q) {
            issue(stage, it + 3);
            compute(it + 1);
            commit(stage_b, it & 1);
            __syncthreads();
        }
    }

    const short *bp = bias + mb * 32 + wave * 8;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        if (PACK) {   // unpack, recover the true sums (see the note above the kernel)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int a16 = m & 1 ? pa[p][m >> 1].y : pa[p][m >> 1].x;
                tl[p][m] = m & 1 ? pl[p][m >> 1].y : pl[p][m >> 1].x;
                th[p][m] = m & 1 ? ph[p][m >> 1].y : ph[p][m >> 1].x;
                const int lo = th[p][m] - 32767;
                ta[p][m] = lo + ((a16 - lo) & 0xffff);
            }
        }
        // ordered combine across the splits with wavefront shuffles (tree: neighbours first)
#pragma unroll
        for (int delta = T; delta < 64; delta <<= 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int a2 = __shfl_down(ta[p][m], delta), l2 = __shfl_down(tl[p][m], delta), h2 = __shfl_down(th[p][m], delta);
                // this lane's map runs first, the partner's (higher split) second: f = f2 o f1
                tl[p][m] = min(max(tl[p][m] + a2, l2), h2);
                th[p][m] = min(max(th[p][m] + a2, l2), h2);
                ta[p][m] += a2;
            }
        }
        // apply to the shifted (unsaturated) bias, integer leaky, store 2 items (8 channels) per pixel
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int cgo = mb * 8 + wave * 2 + g;
            int v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = g * 4 + t;
                const int b = bp[m];
                const int b0 = a.bs_right ? ((b + (a.bs_mag > 0 ? (1 << (a.bs_mag - 1)) : 0)) >> a.bs_mag) : (a.bs_left ? (b << a.bs_mag) : b);
                int e = min(max(b0 + ta[p][m], tl[p][m]), th[p][m]);
                v[t] = a.leaky ? leaky_i16(e) : e;
            }
            int2 o;
            o.x = (v[0] & 0xffff) | (v[1] << 16);
            o.y = (v[2] & 0xffff) | (v[3] << 16);
            if (valid[p] && cgo < a.CGout) out[a.out_base + (long)cgo * a.out_cg_stride + fo[p]] = o;
        }
    }
}

// ------------------------------------------------------------------ K-split across WORKGROUPS, wave-uniform weights (single frames)
//
// k_conv_i16_splitk above splits the saturating chain across the LANES of a wavefront, which makes the weights lane-dependent
// (four weight slices staged through LDS per iteration) and leaves a 13x13 layer of one frame with 192 workgroups of one
// wavefront per SIMD: 159 us for the 1024 -> 1024 layer where the chip's issue rate allows ~40.  Here the split is across
// workgroups: a workgroup is exactly k_conv_i16's (64 pixels x 32 channels, input tile in LDS, weights by scalar loads straight
// into SGPR operands) but walks only channel groups [z Q, (z + 1) Q) and carries, instead of the accumulator, the clamp-affine
// triple (a, l, h) of that sub-chain per output channel - packed int16 pairs exactly like the lane-split kernel's PACK form
// (a wraps, l and h saturate; form D layers only: every increment is the high half of a dot result).  The S triples of an output
// are applied in order to the shifted bias by k_ks_finalize (x -> clamp(x + a, l, h), S times), which also does the leaky and the
// store.  S x more workgroups at the same wavefront efficiency: 13x13 at one frame becomes 3 tiles x 32 blocks x 8 splits = 768
// workgroups.  Bit-exact by the same argument as the lane-split kernel (tests: test_fullnet_ksplit_*).
__device__ __forceinline__ void stepT8(int (&pa)[4], int (&pl)[4], int (&ph)[4], const int2 x, const int2 *w, const int r, const int sel)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %0, %8, %24, %26\n\t"
        "v_dot2_i32_i16 %1, %9, %24, %26\n\t"
        "v_dot2_i32_i16 %2, %10, %24, %26\n\t"
        "v_dot2_i32_i16 %3, %11, %24, %26\n\t"
        "v_dot2_i32_i16 %4, %12, %24, %26\n\t"
        "v_dot2_i32_i16 %5, %13, %24, %26\n\t"
        "v_dot2_i32_i16 %6, %14, %24, %26\n\t"
        "v_dot2_i32_i16 %7, %15, %24, %26\n\t"
        "v_dot2_i32_i16 %0, %16, %25, %0\n\t"
        "v_dot2_i32_i16 %1, %17, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %25, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %25, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %25, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %25, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %25, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %25, %7"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "s"(w[0].x), "s"(w[1].x), "s"(w[2].x), "s"(w[3].x), "s"(w[4].x), "s"(w[5].x), "s"(w[6].x), "s"(w[7].x),
          "s"(w[0].y), "s"(w[1].y), "s"(w[2].y), "s"(w[3].y), "s"(w[4].y), "s"(w[5].y), "s"(w[6].y), "s"(w[7].y),
          "v"(x.x), "v"(x.y), "v"(r));
    // increments of the four channel pairs (high halves), then a += t (wrapping), l = sat16(l + t), h = sat16(h + t)
    asm("v_perm_b32 %12, %13, %12, %20\n\t"
        "v_perm_b32 %14, %15, %14, %20\n\t"
        "v_perm_b32 %16, %17, %16, %20\n\t"
        "v_perm_b32 %18, %19, %18, %20\n\t"
        "v_pk_add_u16 %0, %0, %12\n\t"
        "v_pk_add_i16 %4, %4, %12 clamp\n\t"
        "v_pk_add_i16 %8, %8, %12 clamp\n\t"
        "v_pk_add_u16 %1, %1, %14\n\t"
        "v_pk_add_i16 %5, %5, %14 clamp\n\t"
        "v_pk_add_i16 %9, %9, %14 clamp\n\t"
        "v_pk_add_u16 %2, %2, %16\n\t"
        "v_pk_add_i16 %6, %6, %16 clamp\n\t"
        "v_pk_add_i16 %10, %10, %16 clamp\n\t"
        "v_pk_add_u16 %3, %3, %18\n\t"
        "v_pk_add_i16 %7, %7, %18 clamp\n\t"
        "v_pk_add_i16 %11, %11, %18 clamp"
        : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]), "+v"(pl[3]), "+v"(ph[0]), "+v"(ph[1]),
          "+v"(ph[2]), "+v"(ph[3]), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7)
        : "v"(sel));
}

template <int KS, int NST>
__global__ __launch_bounds__(256, 2) void k_conv_i16_ks(const int2 *__restrict__ in, const int2 *__restrict__ wpk, const ConvArgs a)
{
    extern __shared__ int2 lds[];
    constexpr int KT = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x, yb = blockIdx.y;
    if (a.xcd_remap) xcd_partition(a.xcd_remap - 1, tile, yb);
    const int z = yb / a.ks_mb, ybm = yb - z * a.ks_mb;        // split, block slot
    const int mb = a.mb_list ? a.mb_list[ybm] : ybm;
    const int q0 = tile * 64, qlast = min(q0 + 64, a.npix) - 1;
    const int halo = (KS == 3) ? a.Wp + 1 : 0;
    const int fmin = flat_of(a, q0), fmax = flat_of(a, qlast);
    const int tile_start = fmin - halo;
    const int Lt = min(fmax - fmin + 1 + 2 * halo, a.lt_max);
    const int q = q0 + lane;
    const bool valid = q <= qlast;
    const int lo = flat_of(a, min(q, qlast)) - tile_start;
    int rowaddr[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) rowaddr[i] = (KS == 3) ? (lo + (i - 1) * a.Wp - 1) * 8 : lo * 8;

    int pa[4], pl[4], ph[4];     // the sub-chain's map per channel pair (2j, 2j + 1): identity on int16
#pragma unroll
    for (int j = 0; j < 4; ++j) { pa[j] = 0; pl[j] = (int)0x80008000u; ph[j] = 0x7fff7fff; }
    int r_vgpr = a.round;
    asm volatile("" : "+v"(r_vgpr));
    const char *lds_b = reinterpret_cast<const char *>(lds);
    const int g0 = z * a.ks_Q;
    const int2 *src = in + kLead + tile_start + (long)g0 * a.in_cg_stride;
    const int2 *wq = wpk + (((long)mb * a.CGin + g0) * KT * 32 + wave * 8);

    int2 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < Lt) stage[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        if (i < Lt) lds[i] = stage[k];
    }
    __syncthreads();
    for (int cg = 0; cg < a.ks_Q; ++cg) {
        src += (cg + 1 < a.ks_Q) ? a.in_cg_stride : 0;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 256;
            if (i < Lt) stage[k] = src[i];
        }
        const char *tl = lds_b + (cg & 1) * a.lt_max * 8;
        int2 xv[KT];
#pragma unroll
        for (int tt = 0; tt < KT; ++tt) xv[tt] = *reinterpret_cast<const int2 *>(tl + rowaddr[tt / KS] + (tt % KS) * 8);
#pragma unroll
        for (int tt = 0; tt < KT; ++tt) {
            int2 w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) w[m] = wq[tt * 32 + m];   // wave-uniform: scalar loads
            stepT8(pa, pl, ph, xv[tt], w, r_vgpr, 0x07060302);
        }
        {
            int2 *nxt = lds + ((cg + 1) & 1) * a.lt_max;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = tid + k * 256;
                if (i < Lt) nxt[i] = stage[k];
            }
        }
        wq += KT * 32;
        __syncthreads();
    }
    // triples out: [split][channel item][pixel][pair 0: a l h, pair 1: a l h]
    if (valid) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int cgo = mb * 8 + wave * 2 + g;
            if (cgo >= a.CGout) continue;
            int *dst = a.ks_trip + (((long)z * a.CGout + cgo) * a.npix + q) * 6;
            *reinterpret_cast<int2 *>(dst) = make_int2(pa[2 * g], pl[2 * g]);
            *reinterpret_cast<int2 *>(dst + 2) = make_int2(ph[2 * g], pa[2 * g + 1]);
            *reinterpret_cast<int2 *>(dst + 4) = make_int2(pl[2 * g + 1], ph[2 * g + 1]);
        }
    }
}

// applies the S sub-chain maps of every output in order to its shifted bias, then leaky + store (one thread per pixel and item)
__global__ __launch_bounds__(256) void k_ks_finalize(const int *__restrict__ trip, int2 *__restrict__ out, const short *__restrict__ bias,
                                                      const ConvArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.npix * a.CGout) return;
    const int cgo = t / a.npix, q = t - cgo * a.npix;
    int e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = bias[cgo * 4 + k];       // (bias_pk is padded to whole blocks of 32)
        e[k] = a.bs_right ? ((b + (a.bs_mag > 0 ? (1 << (a.bs_mag - 1)) : 0)) >> a.bs_mag) : (a.bs_left ? (b << a.bs_mag) : b);
    }
    for (int z = 0; z < a.ks_S; ++z) {
        const int *src = trip + (((long)z * a.CGout + cgo) * a.npix + q) * 6;
        const int2 v0 = *reinterpret_cast<const int2 *>(src), v1 = *reinterpret_cast<const int2 *>(src + 2), v2 = *reinterpret_cast<const int2 *>(src + 4);
        const int pa[2] = {v0.x, v1.y}, pl[2] = {v0.y, v2.x}, ph[2] = {v1.x, v2.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = k >> 1, hi = k & 1;
            const int a16 = hi ? (pa[j] >> 16) : (int)(short)(pa[j] & 0xffff);
            const int l = hi ? (pl[j] >> 16) : (int)(short)(pl[j] & 0xffff);
            const int h = hi ? (ph[j] >> 16) : (int)(short)(ph[j] & 0xffff);
            // the true sum of the increments: f(-32768) = l and f(32767) = h confine it to [h - 32767, l + 32768] (see k_conv_i16_splitk)
            const int lo = h - 32767;
            const int at = lo + ((a16 - lo) & 0xffff);
            e[k] = min(max(e[k] + at, l), h);
        }
    }
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = a.leaky ? leaky_i16(e[k]) : e[k];
    int2 o;
    o.x = (v[0] & 0xffff) | (v[1] << 16);
    o.y = (v[2] & 0xffff) | (v[3] << 16);
    out[a.out_base + (long)cgo * a.out_cg_stride + flat_of(a, q)] = o;
}

// ------------------------------------------------------------------ small kernels

// float [B][3][416][416] -> quantised items (C=3, 4th lane 0).  yolo2_model.cpp:257-273.
__global__ void k_pack_input(const float *__restrict__ frames, int2 *__restrict__ out, int B, int H, int W,
                             int Wp, int PL, float scale)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (q >= B * HW) return;
    const int b = q / HW, r = q - b * HW;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = frames[((long)b * 3 + c) * HW + r] * scale;
        f = fminf(fmaxf(f, -32768.f), 32767.f);
        v[c] = clamp16((int)roundf(f));  // llround: half away from zero
    }
    const int y = r / W, x = r - y * W;
    int2 o;
    o.x = (v[0] & 0xffff) | (v[1] << 16);
    o.y = (v[2] & 0xffff);
    out[kLead + (long)b * PL + (long)(y + 1) * Wp + x] = o;
}

__device__ __forceinline__ int pkmax(int a, int b)
{
    short2_t r = __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b));
    return __builtin_bit_cast(int, r);
}

// 2x2 stride-2 max pool on items (pool_yolo2, core_compute.cpp:266-305; even H, W: no edge).
__global__ void k_maxpool2(const int2 *__restrict__ in, int2 *__restrict__ out, int CG, int B, int OH, int OW,
                           int iWp, int iPL, int oWp, int oPL)
{
    const long n = (long)CG * B * OH * OW;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int x = (int)(t % OW);
    const int y = (int)((t / OW) % OH);
    const long pb = t / ((long)OW * OH);  // cg*B + b
    const int2 *s = in + kLead + pb * iPL + (long)(2 * y + 1) * iWp + 2 * x;
    const int2 a = s[0], b = s[1], c = s[iWp], d = s[iWp + 1];
    int2 o;
    o.x = pkmax(pkmax(a.x, b.x), pkmax(c.x, d.x));
    o.y = pkmax(pkmax(a.y, b.y), pkmax(c.y, d.y));
    out[kLead + pb * oPL + (long)(y + 1) * oWp + x] = o;
}

// Darknet legacy reorg (stride 2) of the 64x26x26 tensor + route-28 Q alignment shift
// (yolo2_model.cpp:112-129, 358-403): out[i + 26*(j + 416*k)] = x[(2i + k%2) + 52*(2j + k/2)]
// over flat [64*26*26]; destination is channel groups [0,64) of the 1280-channel concat tensor.
__global__ void k_reorg(const short *__restrict__ in, short *__restrict__ out, int B, int iWp, int iPL,
                        long i_cg_stride, int oWp, int oPL, long o_cg_stride, int shift)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;  // over B * 256*169 output elements
    if (t >= B * 256 * 169) return;
    const int b = t / (256 * 169);
    const int o = t - b * (256 * 169);  // flat [256][13][13] == flat i + 26*(j + 416*k)
    const int k = o / (26 * 416);
    const int rem = o - k * (26 * 416);
    const int j = rem / 26, i = rem - j * 26;
    const int sidx = (2 * i + (k & 1)) + 52 * (2 * j + (k >> 1));  // flat [64][26][26]
    const int sc = sidx / 676, sr = sidx - sc * 676;
    const int sy = sr / 26, sx = sr - sy * 26;
    int v = in[(kLead + (long)(sc >> 2) * i_cg_stride + (long)b * iPL + (long)(sy + 1) * iWp + sx) * 4 + (sc & 3)];
    if (shift > 0) v >>= shift;                 // arithmetic, no rounding (yolo2_model.cpp:387-388)
    else if (shift < 0) v = (int)((unsigned)v << (-shift));
    v = clamp16(v);
    const int oc = o / 169, orr = o - oc * 169;
    const int oy = orr / 13, ox = orr - oy * 13;
    out[(kLead + (long)(oc >> 2) * o_cg_stride + (long)b * oPL + (long)(oy + 1) * oWp + ox) * 4 + (oc & 3)] = (short)v;
}

// items (C channels) -> dense [B][C][H][W] int16 (the 13-of-16 region gather, yolo2_model.cpp:406-414)
__global__ void k_unpack_dense(const short *__restrict__ in, short *__restrict__ out, int B, int C, int H, int W,
                               int Wp, int PL, long cg_stride)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)B * C * H * W;
    if (t >= n) return;
    const int x = (int)(t % W);
    const int y = (int)((t / W) % H);
    const int c = (int)((t / ((long)W * H)) % C);
    const int b = (int)(t / ((long)W * H * C));
    out[t] = in[(kLead + (long)(c >> 2) * cg_stride + (long)b * PL + (long)(y + 1) * Wp + x) * 4 + (c & 3)];
}

// reference layout [C][H][W8] -> items (B = 1), and back (pad columns of the destination untouched,
// like the reference's row write-back of TC_MIN elements, core_compute.cpp:212-220)
__global__ void k_ref_to_items(const short *__restrict__ in, short *__restrict__ out, int C, int H, int W, int W8,
                               int Wp, long cg_stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * H * W) return;
    const int x = t % W, y = (t / W) % H, c = t / (W * H);
    out[(kLead + (long)(c >> 2) * cg_stride + (long)(y + 1) * Wp + x) * 4 + (c & 3)] = in[((long)c * H + y) * W8 + x];
}

__global__ void k_items_to_ref(const short *__restrict__ in, short *__restrict__ out, int C, int H, int W, int W8,
                               int Wp, int PL, long cg_stride, int b)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * H * W) return;
    const int x = t % W, y = (t / W) % H, c = t / (W * H);
    out[((long)c * H + y) * W8 + x] = in[(kLead + (long)(c >> 2) * cg_stride + (long)b * PL + (long)(y + 1) * Wp + x) * 4 + (c & 3)];
}


// max over (m, group, tap) of sum_t |w_t| on the packed weights: the 32-bit exactness bound.
__global__ void k_weight_bound(const short *__restrict__ wpk, long n_quads, int *__restrict__ result)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int best = 0;
    for (; t < n_quads; t += (long)gridDim.x * blockDim.x) {
        const short *w = wpk + t * 4;
        int s = abs((int)w[0]) + abs((int)w[1]) + abs((int)w[2]) + abs((int)w[3]);
        best = max(best, s);
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) atomicMax(result, best);
}

// Same bound per output-channel block of 32 (one workgroup per block): lets the loader pick the
// arithmetic form per block, so a few large-weight channels do not slow the whole layer down.
// result_abs: the block's largest |w| (the int16 headroom form D needs to fold the shift into the weights).
__global__ void k_weight_bound_mb(const short *__restrict__ wpk, long quads_per_mb, int *__restrict__ result,
                                  int *__restrict__ result_abs)
{
    __shared__ int red[4], red_abs[4];
    const short *base = wpk + (long)blockIdx.x * quads_per_mb * 4;
    int best = 0, big = 0;
    for (long t = threadIdx.x; t < quads_per_mb; t += blockDim.x) {
        const short *w = base + t * 4;
        const int a0 = abs((int)w[0]), a1 = abs((int)w[1]), a2 = abs((int)w[2]), a3 = abs((int)w[3]);
        best = max(best, a0 + a1 + a2 + a3);
        big = max(big, max(max(a0, a1), max(a2, a3)));
    }
    for (int o = 32; o > 0; o >>= 1) { best = max(best, __shfl_xor(best, o)); big = max(big, __shfl_xor(big, o)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = best; red_abs[threadIdx.x >> 6] = big; }
    __syncthreads();
    if (threadIdx.x == 0) {
        result[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
        result_abs[blockIdx.x] = max(max(red_abs[0], red_abs[1]), max(red_abs[2], red_abs[3]));
    }
}

// Multiplies the packed weights of output-channel block `mb` by 2^shl[mb] (shl < 0: exact arithmetic
// right shift back).  Used once per (re)resolution of the arithmetic forms: blocks that run form D
// keep their weights pre-scaled by 2^(16-s).  grid = (chunks, MB).
__global__ void k_scale_weight_blocks(short *__restrict__ wpk, long elems_per_mb, const signed char *__restrict__ shl)
{
    const int k = shl[blockIdx.y];
    if (k == 0) return;
    short *base = wpk + (long)blockIdx.y * elems_per_mb;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < elems_per_mb; t += (long)gridDim.x * blockDim.x) {
        const int v = base[t];
        base[t] = (short)(k > 0 ? v * (1 << k) : v >> (-k));
    }
}

// ------------------------------------------------------------------ generic reference-layout kernels
// One thread per output element, reference arithmetic verbatim (64-bit), any K<=3, stride<=2,
// padding<=4.  Used by the per-layer driver calls for shapes the tiled kernel does not cover,
// and as an independent second implementation in the GPU tests.
__global__ void k_conv_ref_i16(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ w,
                               const short *__restrict__ bias, int C, int N, int K, int stride, int W, int H,
                               int OW, int OH, int pad, int leaky, int so, int sb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * OH * OW) return;
    const int x = t % OW, y = (t / OW) % OH, m = t / (OW * OH);
    const int W8 = (W + 7) & ~7, OW8 = (OW + 7) & ~7, KK = K * K;
    const int so_r = so > 0, so_l = so < 0, so_m = min(so_r ? so : -so, 30);
    const int sb_r = sb > 0, sb_l = sb < 0, sb_m = min(sb_r ? sb : -sb, 30);
    long acc = shift64((long)bias[m], sb_r, sb_l, sb_m, sb_r && sb_m > 0 ? (1L << (sb_m - 1)) : 0);
    const int m0 = m / kTm * kTm, tm = m - m0, tm_min = min(kTm, N - m0);
    for (int n0 = 0; n0 < C; n0 += kTn) {
        const int tn_min = min(kTn, C - n0);
        const short *wb = w + (long)m0 * C * KK + (long)tm_min * n0 * KK;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                const int sy = y * stride + i - pad, sx = x * stride + j - pad;
                const bool inb = sy >= 0 && sy < H && sx >= 0 && sx < W;
                long p = 0;
                for (int tt = 0; tt < tn_min; ++tt) {
                    const int wv = wb[(long)(i * K + j) * tm_min * tn_min + tm * tn_min + tt];
                    const int xv = inb ? in[((long)(n0 + tt) * H + sy) * W8 + sx] : 0;
                    p += (long)(wv * xv);
                }
                long v = acc + shift64(p, so_r, so_l, so_m, so_r && so_m > 0 ? (1L << (so_m - 1)) : 0);
                acc = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
            }
    }
    int e = (int)acc;
    if (leaky) e = leaky_i16(e);
    out[((long)m * OH + y) * OW8 + x] = (short)clamp16(e);
}



}  // namespace y2
