// ubench_step.hip -- what the int16 requantise step COSTS on gfx950 as an instruction MIX (tools/ubench_valu.hip prices one opcode
// at a time).  Each kernel repeats one candidate step sequence for 8 output channels of one (pixel, tap) - operands in registers,
// no memory access at all - with 1 to 8 wavefronts per SIMD on every CU; cycles by s_memtime, wall time by hipEvents.
// The number to compare with bench.py's `valu_roofline` is "cycles per step" = cycles per sequence / 8 channels: the roofline
// prices form D at 12 (2 x v_dot2 + half a v_perm + half a v_pk_add at 4 cycles each).
//   build: hipcc -O3 --offload-arch=gfx950 -o ubench_step ubench_step.hip ; run: ./ubench_step
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define ITER 2048

// 16 dots: channel m = 0..7, t_m = dot(x.x, w_m.x) + r ; t_m = dot(x.y, w_m.y) + t_m      (%0-%7 t, %8-%11 acc, w: sgpr or vgpr)
#define DOTS_S                                                                                                          \
    "v_dot2_i32_i16 %0, %12, %28, %30\n\tv_dot2_i32_i16 %1, %13, %28, %30\n\tv_dot2_i32_i16 %2, %14, %28, %30\n\tv_dot2_i32_i16 %3, %15, %28, %30\n\t" \
    "v_dot2_i32_i16 %4, %16, %28, %30\n\tv_dot2_i32_i16 %5, %17, %28, %30\n\tv_dot2_i32_i16 %6, %18, %28, %30\n\tv_dot2_i32_i16 %7, %19, %28, %30\n\t" \
    "v_dot2_i32_i16 %0, %20, %29, %0\n\tv_dot2_i32_i16 %1, %21, %29, %1\n\tv_dot2_i32_i16 %2, %22, %29, %2\n\tv_dot2_i32_i16 %3, %23, %29, %3\n\t"     \
    "v_dot2_i32_i16 %4, %24, %29, %4\n\tv_dot2_i32_i16 %5, %25, %29, %5\n\tv_dot2_i32_i16 %6, %26, %29, %6\n\tv_dot2_i32_i16 %7, %27, %29, %7\n\t"
#define PERMS "v_perm_b32 %0, %1, %0, %31\n\tv_perm_b32 %2, %3, %2, %31\n\tv_perm_b32 %4, %5, %4, %31\n\tv_perm_b32 %6, %7, %6, %31\n\t"
#define ADDS4 "v_pk_add_i16 %8, %8, %0 clamp\n\tv_pk_add_i16 %9, %9, %2 clamp\n\tv_pk_add_i16 %10, %10, %4 clamp\n\tv_pk_add_i16 %11, %11, %6 clamp\n\t"
#define ADDS8 ADDS4 "v_pk_add_i16 %8, %8, %1 clamp\n\tv_pk_add_i16 %9, %9, %3 clamp\n\tv_pk_add_i16 %10, %10, %5 clamp\n\tv_pk_add_i16 %11, %11, %7 clamp\n\t"
#define NOPS4 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"

#define KERNEL(NAME, WCON, SEQ)                                                                                         \
    __global__ void NAME(int *out, unsigned long long *cyc, int s0, int s1)                                            \
    {                                                                                                                   \
        int t0, t1, t2, t3, t4, t5, t6, t7;                                                                             \
        int a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;                                                    \
        int x0 = threadIdx.x * 0x10003 + 77, x1 = x0 ^ 0x5555, r = 32768, sel = 0x07060302;                             \
        int w[16];                                                                                                      \
        for (int i = 0; i < 16; ++i) w[i] = s0 * (i + 1) + s1;                                                          \
        unsigned long long c0, c1;                                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");                                      \
        for (int it = 0; it < ITER; ++it) {                                                                             \
            asm volatile(SEQ                                                                                            \
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "+v"(a0), "+v"(a1), \
                           "+v"(a2), "+v"(a3)                                                                           \
                         : WCON(w[0]), WCON(w[1]), WCON(w[2]), WCON(w[3]), WCON(w[4]), WCON(w[5]), WCON(w[6]), WCON(w[7]), WCON(w[8]),  \
                           WCON(w[9]), WCON(w[10]), WCON(w[11]), WCON(w[12]), WCON(w[13]), WCON(w[14]), WCON(w[15]), "v"(x0), "v"(x1),  \
                           "v"(r), "v"(sel));                                                                           \
        }                                                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + t0 + t7;                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;               \
    }

#define SG "s"
#define VG "v"
KERNEL(k_stepD_s, SG, DOTS_S PERMS ADDS4)          // the kernel's form D: weights in SGPRs
KERNEL(k_stepD_v, VG, DOTS_S PERMS ADDS4)          // the same with weights in VGPRs
KERNEL(k_dots_s, SG, DOTS_S)
KERNEL(k_dots_v, VG, DOTS_S)
KERNEL(k_dots_perm_s, SG, DOTS_S PERMS)
KERNEL(k_dots_add4_s, SG, DOTS_S ADDS4)
KERNEL(k_dots_add8_s, SG, DOTS_S ADDS8)            // one saturating add per channel on the high half, no v_perm (same count as form D)
KERNEL(k_tail_only, SG, PERMS ADDS4)
KERNEL(k_stepD_s_nop, SG, DOTS_S NOPS4 PERMS ADDS4)

typedef void (*kfn)(int *, unsigned long long *, int, int);

int main()
{
    struct { const char *name; kfn f; int instr; } ks[] = {
        {"form D: 16 dot2 (sgpr w) + 4 perm + 4 pk_add", k_stepD_s, 24}, {"form D with vgpr weights", k_stepD_v, 24},
        {"16 dot2 (sgpr w) only", k_dots_s, 16}, {"16 dot2 (vgpr w) only", k_dots_v, 16}, {"16 dot2 + 4 perm", k_dots_perm_s, 20},
        {"16 dot2 + 4 pk_add clamp", k_dots_add4_s, 20}, {"16 dot2 + 8 pk_add clamp (no perm)", k_dots_add8_s, 24},
        {"4 perm + 4 pk_add only", k_tail_only, 8}, {"form D + 4 s_nop between dots and perms", k_stepD_s_nop, 28}};
    int *out;
    unsigned long long *cyc;
    const int nblk = 256 * 8;  // up to 8 blocks of 256 threads per CU
    hipMalloc(&out, sizeof(int) * nblk * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * 4);
    std::vector<unsigned long long> h(nblk * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-50s %s\n", "sequence (8 channels of one pixel x tap)", "cycles per STEP (= per sequence / 8) by s_memtime [median wave] | by wall clock at 2.4 GHz, at 1 / 2 / 4 / 8 waves per SIMD");
    for (auto &k : ks) {
        printf("%-50s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            // 256-thread blocks = one wavefront per SIMD each; wps blocks per CU = wps wavefronts per SIMD, all resident at once
            const int threads = 256, nb = 256 * wps;
            hipLaunchKernelGGL(k.f, dim3(nb), dim3(threads), 0, 0, out, cyc, 3, 7);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.f, dim3(nb), dim3(threads), 0, 0, out, cyc, 3, 7);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const int nw = nb * 4;
            hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + nw);
            const double per = (double)h[nw / 2] / ITER / wps / 8.0;                   // SIMD cycles per step by the median wavefront's own s_memtime span
            const double wall = ms * 1e-3 * 2.4e9 / ((double)ITER * wps) / 8.0;        // the same from wall time, priced at 2.4 GHz
            printf("  %5.2f|%5.2f", per, wall);
        }
        printf("\n");
    }
    return 0;
}
