// ubench_valu.hip -- per-instruction issue cost on gfx950 for the integer ops the int16 conv step
// can be built from.  Each kernel runs ITER x 64 instructions of one kind on 8 independent
// register chains per lane; cycles are read with s_memtime inside the kernel (shader clock), so
// the result is cycles per wave-instruction per SIMD at a given number of waves per SIMD.
//   build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu ubench_valu.hip ; run: ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ITER 512

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS)

#define KERNEL(NAME, ASM_LINE)                                                                     \
    __global__ void NAME(int *out, unsigned long long *cyc, int sa, int sb)                        \
    {                                                                                              \
        int v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13,   \
            v6 = v0 * 17, v7 = v0 * 19;                                                            \
        int x = threadIdx.x * 0x10003 + 77, y = x ^ 0x5555;                                        \
        unsigned long long t0, t1;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                 \
        for (int it = 0; it < ITER; ++it) {                                                        \
            asm volatile(BODY8(ASM_LINE)                                                           \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(x), "v"(y), "s"(sa), "s"(sb));                                      \
        }                                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;        \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                           \
    }

#define I_DOT2C(n) "v_dot2c_i32_i16 %" #n ", %10, %8\n\t"
#define I_DOT2(n) "v_dot2_i32_i16 %" #n ", %10, %8, %" #n "\n\t"
#define I_DOT2V(n) "v_dot2_i32_i16 %" #n ", %9, %8, %" #n "\n\t"
#define I_ADD(n) "v_add_u32 %" #n ", %8, %" #n "\n\t"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", %10, %" #n "\n\t"
#define I_MED3(n) "v_med3_i32 %" #n ", %" #n ", %10, %9\n\t"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %10, %9\n\t"
#define I_PKADD(n) "v_pk_add_i16 %" #n ", %" #n ", %8 clamp\n\t"
#define I_ADDCL(n) "v_add_i32 %" #n ", %" #n ", %8 clamp\n\t"
#define I_SDWA(n) "v_ashrrev_i32_sdwa %" #n ", %10, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
#define I_MAD24(n) "v_mad_i32_i24 %" #n ", %8, %9, %" #n "\n\t"
#define I_MADI16(n) "v_mad_i32_i16 %" #n ", %8, %9, %" #n "\n\t"
#define I_PKMAD(n) "v_pk_mad_i16 %" #n ", %8, %9, %" #n "\n\t"
#define I_DOT4(n) "v_dot4_i32_i8 %" #n ", %8, %9, %" #n "\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %10\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", %10, %9\n\t"
#define I_MAX(n) "v_max_i32 %" #n ", %8, %" #n "\n\t"
#define I_PKMAX(n) "v_pk_max_i16 %" #n ", %8, %" #n "\n\t"
#define I_BFE(n) "v_bfe_i32 %" #n ", %" #n ", %10, 16\n\t"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, %10\n\t"

KERNEL(k_dot2c, I_DOT2C)
KERNEL(k_dot2, I_DOT2)
KERNEL(k_dot2v, I_DOT2V)
KERNEL(k_add, I_ADD)
KERNEL(k_ashr, I_ASHR)
KERNEL(k_med3, I_MED3)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_addcl, I_ADDCL)
KERNEL(k_sdwa, I_SDWA)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_madi16, I_MADI16)
KERNEL(k_pkmad, I_PKMAD)
KERNEL(k_dot4, I_DOT4)
KERNEL(k_fma, I_FMA)
KERNEL(k_perm, I_PERM)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_max, I_MAX)
KERNEL(k_pkmax, I_PKMAX)
KERNEL(k_bfe, I_BFE)
KERNEL(k_align, I_ALIGN)

#define I_X_and(n) "v_and_b32 %" #n ", %8, %" #n "\n\t"
#define I_X_or(n) "v_or_b32 %" #n ", %8, %" #n "\n\t"
#define I_X_xor(n) "v_xor_b32 %" #n ", %8, %" #n "\n\t"
#define I_X_lshl(n) "v_lshlrev_b32 %" #n ", %10, %" #n "\n\t"
#define I_X_sub(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
#define I_X_mov(n) "v_mov_b32 %" #n ", %8\n\t"
#define I_X_cndmask(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n\t"
#define I_X_mullo(n) "v_mul_lo_u32 %" #n ", %8, %" #n "\n\t"
#define I_X_add3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_maxf(n) "v_max_f32 %" #n ", %8, %" #n "\n\t"
#define I_X_minf(n) "v_min_f32 %" #n ", %8, %" #n "\n\t"
#define I_X_med3f(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_mulf(n) "v_mul_f32 %" #n ", %8, %" #n "\n\t"
#define I_X_addf(n) "v_add_f32 %" #n ", %8, %" #n "\n\t"
#define I_X_cvtfi(n) "v_cvt_f32_i32 %" #n ", %" #n "\n\t"
#define I_X_cvtif(n) "v_cvt_i32_f32 %" #n ", %" #n "\n\t"
#define I_X_mul24(n) "v_mul_i32_i24 %" #n ", %8, %" #n "\n\t"
#define I_X_madu24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n\t"
#define I_X_bfi(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n\t"
#define I_X_mini(n) "v_min_i32 %" #n ", %8, %" #n "\n\t"
#define I_X_addsg(n) "v_add_u32 %" #n ", %10, %" #n "\n\t"
#define I_X_mulfsg(n) "v_mul_f32 %" #n ", %10, %" #n "\n\t"
#define I_X_addfsg(n) "v_add_f32 %" #n ", %10, %" #n "\n\t"
#define I_X_movsg(n) "v_mov_b32 %" #n ", %10\n\t"
#define I_X_fmacsg(n) "v_fmac_f32 %" #n ", %10, %8\n\t"
#define I_X_subrev(n) "v_subrev_u32 %" #n ", %8, %" #n "\n\t"
#define I_X_fmac(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
#define I_X_addi16(n) "v_add_i16 %" #n ", %" #n ", %8 clamp\n\t"
#define I_X_max3i(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_xad(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_lshlor(n) "v_lshl_or_b32 %" #n ", %" #n ", %10, %9\n\t"
#define I_X_addlshl(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, %10\n\t"
#define I_X_andor_v(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_dot2f(n) "v_dot2c_f32_f16 %" #n ", %8, %9\n\t"
#define I_X_sat_sub(n) "v_sub_i32 %" #n ", %" #n ", %8 clamp\n\t"
#define I_X_minu(n) "v_min_u32 %" #n ", %8, %" #n "\n\t"
#define I_X_pkminu(n) "v_pk_min_u16 %" #n ", %8, %" #n "\n\t"
#define I_X_mad_u16(n) "v_mad_u16 %" #n ", %8, %9, %" #n "\n\t"
#define I_X_msad(n) "v_msad_u8 %" #n ", %8, %9, %" #n "\n\t"
KERNEL(k_x_and, I_X_and)
KERNEL(k_x_or, I_X_or)
KERNEL(k_x_xor, I_X_xor)
KERNEL(k_x_lshl, I_X_lshl)
KERNEL(k_x_sub, I_X_sub)
KERNEL(k_x_mov, I_X_mov)
KERNEL(k_x_cndmask, I_X_cndmask)
KERNEL(k_x_mullo, I_X_mullo)
KERNEL(k_x_add3, I_X_add3)
KERNEL(k_x_maxf, I_X_maxf)
KERNEL(k_x_minf, I_X_minf)
KERNEL(k_x_med3f, I_X_med3f)
KERNEL(k_x_mulf, I_X_mulf)
KERNEL(k_x_addf, I_X_addf)
KERNEL(k_x_cvtfi, I_X_cvtfi)
KERNEL(k_x_cvtif, I_X_cvtif)
KERNEL(k_x_mul24, I_X_mul24)
KERNEL(k_x_madu24, I_X_madu24)
KERNEL(k_x_bfi, I_X_bfi)
KERNEL(k_x_mini, I_X_mini)
KERNEL(k_x_addsg, I_X_addsg)
KERNEL(k_x_mulfsg, I_X_mulfsg)
KERNEL(k_x_addfsg, I_X_addfsg)
KERNEL(k_x_movsg, I_X_movsg)
KERNEL(k_x_fmacsg, I_X_fmacsg)
KERNEL(k_x_subrev, I_X_subrev)
KERNEL(k_x_fmac, I_X_fmac)
KERNEL(k_x_addi16, I_X_addi16)
KERNEL(k_x_max3i, I_X_max3i)
KERNEL(k_x_xad, I_X_xad)
KERNEL(k_x_lshlor, I_X_lshlor)
KERNEL(k_x_addlshl, I_X_addlshl)
KERNEL(k_x_andor_v, I_X_andor_v)
KERNEL(k_x_dot2f, I_X_dot2f)
KERNEL(k_x_sat_sub, I_X_sat_sub)
KERNEL(k_x_minu, I_X_minu)
KERNEL(k_x_pkminu, I_X_pkminu)
KERNEL(k_x_mad_u16, I_X_mad_u16)
KERNEL(k_x_msad, I_X_msad)

typedef void (*kfn)(int *, unsigned long long *, int, int);

int main()
{
    struct { const char *name; kfn f; } ks[] = {
        {"v_dot2c_i32_i16 (VOP2, sgpr w)", k_dot2c}, {"v_dot2_i32_i16 (VOP3P, sgpr w)", k_dot2},
        {"v_dot2_i32_i16 (VOP3P, vgpr w)", k_dot2v}, {"v_add_u32", k_add}, {"v_ashrrev_i32", k_ashr},
        {"v_med3_i32", k_med3}, {"v_and_or_b32", k_andor}, {"v_pk_add_i16 clamp", k_pkadd},
        {"v_add_i32 clamp", k_addcl}, {"v_ashrrev_i32_sdwa WORD_1 preserve", k_sdwa}, {"v_mad_i32_i24", k_mad24},
        {"v_mad_i32_i16", k_madi16}, {"v_pk_mad_i16", k_pkmad}, {"v_dot4_i32_i8", k_dot4}, {"v_fma_f32", k_fma},
        {"v_perm_b32", k_perm}, {"v_lshl_add_u32", k_lshladd}, {"v_max_i32", k_max}, {"v_pk_max_i16", k_pkmax},
        {"v_bfe_i32", k_bfe}, {"v_alignbit_b32", k_align},
        {"v_and_b32 [and]", k_x_and}, {"v_or_b32 [or]", k_x_or}, {"v_xor_b32 [xor]", k_x_xor}, {"v_lshlrev_b32 [lshl]", k_x_lshl}, {"v_sub_u32 [sub]", k_x_sub}, {"v_mov_b32 [mov]", k_x_mov}, {"v_cndmask_b32 [cndmask]", k_x_cndmask}, {"v_mul_lo_u32 [mullo]", k_x_mullo}, {"v_add3_u32 [add3]", k_x_add3}, {"v_max_f32 [maxf]", k_x_maxf}, {"v_min_f32 [minf]", k_x_minf}, {"v_med3_f32 [med3f]", k_x_med3f}, {"v_mul_f32 [mulf]", k_x_mulf}, {"v_add_f32 [addf]", k_x_addf}, {"v_cvt_f32_i32 [cvtfi]", k_x_cvtfi}, {"v_cvt_i32_f32 [cvtif]", k_x_cvtif}, {"v_mul_i32_i24 [mul24]", k_x_mul24}, {"v_mad_u32_u24 [madu24]", k_x_madu24}, {"v_bfi_b32 [bfi]", k_x_bfi}, {"v_min_i32 [mini]", k_x_mini}, {"v_add_u32 [addsg]", k_x_addsg}, {"v_mul_f32 sgpr operand [mulfsg]", k_x_mulfsg}, {"v_add_f32 sgpr operand [addfsg]", k_x_addfsg}, {"v_mov_b32 from sgpr [movsg]", k_x_movsg}, {"v_fmac_f32 sgpr operand [fmacsg]", k_x_fmacsg}, {"v_subrev_u32 [subrev]", k_x_subrev}, {"v_fmac_f32 [fmac]", k_x_fmac}, {"v_add_i16 [addi16]", k_x_addi16}, {"v_max3_i32 [max3i]", k_x_max3i}, {"v_xad_u32 [xad]", k_x_xad}, {"v_lshl_or_b32 [lshlor]", k_x_lshlor}, {"v_add_lshl_u32 [addlshl]", k_x_addlshl}, {"v_and_or_b32 [andor_v]", k_x_andor_v}, {"v_dot2c_f32_f16 [dot2f]", k_x_dot2f}, {"v_sub_i32 [sat_sub]", k_x_sat_sub}, {"v_min_u32 [minu]", k_x_minu}, {"v_pk_min_u16 [pkminu]", k_x_pkminu}, {"v_mad_u16 [mad_u16]", k_x_mad_u16}, {"v_msad_u8 [msad]", k_x_msad}};
    int *out;
    unsigned long long *cyc;
    const int nblk = 256 * 4;  // 4 blocks per CU
    hipMalloc(&out, sizeof(int) * nblk * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * 16);
    std::vector<unsigned long long> h(nblk * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-40s %s\n", "instruction", "SIMD cycles per wave-instruction (slowest wave) and wall ns per instr per SIMD, at 1 / 2 / 4 / 8 waves per SIMD");
    for (auto &k : ks) {
        printf("%-40s", k.name);
        for (int wps : {1, 4}) {
            const int threads = 64 * wps;  // 4 blocks/CU x wps waves = wps waves per SIMD
            hipLaunchKernelGGL(k.f, dim3(nblk), dim3(threads), 0, 0, out, cyc, 3, 7);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.f, dim3(nblk), dim3(threads), 0, 0, out, cyc, 3, 7);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const int nw = nblk * wps;
            hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + nw);
            // SIMD cycles per instruction: the slowest wave's elapsed cycles cover wps waves' work
            const double per = (double)h[nw - 1] / (ITER * 64.0) / wps;
            // wall-clock view: instructions per SIMD / time, in ns per instruction
            const double ns = ms * 1e6 / (ITER * 64.0 * wps);
            printf("  %5.2f(%4.2fns)", per, ns);
        }
        printf("\n");
    }
    return 0;
}
