// ubench_mfma.hip -- what the matrix cores of THIS chip sustain on fp16 operands, as a ceiling for the
// fp16 conv kernels that is anchored on the hardware instead of the data-sheet clock:
//   * a bare MFMA loop (operands in registers, four accumulators per wave) on RANDOM and on ZERO data,
//     v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16, 1 / 2 / 4 waves per SIMD on every CU;
//   * the same loop with every operand re-read from LDS by ds_read_b128 (the conv kernels' inner loop
//     without staging, barriers or epilogue).
// Reported per case: TFLOP/s over the whole chip (wall clock, hipEvents around back-to-back launches that
// run >= 0.3 s in total), cycles per MFMA per SIMD (s_memtime) and the clock the chip held
// (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6).
//   build: hipcc -O3 --offload-arch=gfx950 -o ubench_mfma ubench_mfma.hip ; run: ./ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 4096;   // MFMA groups per wave and launch

template <int SHAPE, bool LDS>   // SHAPE 32: 32x32x16, 16: 16x16x32
__global__ __launch_bounds__(1024) void k_mfma(const half8_t *__restrict__ src, float *__restrict__ sink, unsigned long long *__restrict__ stamps)
{
    __shared__ half8_t tile[4 * 64 * 4];   // [wave % 4][operand 0..3][lane]: 16 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8_t a0 = src[(blockIdx.x * 64 + lane) * 4 + 0], a1 = src[(blockIdx.x * 64 + lane) * 4 + 1];
    half8_t b0 = src[(blockIdx.x * 64 + lane) * 4 + 2], b1 = src[(blockIdx.x * 64 + lane) * 4 + 3];
    half8_t *mine = tile + (wave & 3) * 256;
    mine[0 * 64 + lane] = a0; mine[1 * 64 + lane] = a1; mine[2 * 64 + lane] = b0; mine[3 * 64 + lane] = b1;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 32) {
        f16v c00 = {}, c01 = {}, c10 = {}, c11 = {};
#pragma unroll 4
        for (int it = 0; it < kIters; ++it) {
            if constexpr (LDS) {
                a0 = mine[0 * 64 + lane]; a1 = mine[1 * 64 + lane]; b0 = mine[2 * 64 + lane]; b1 = mine[3 * 64 + lane];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            }
            c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c11, 0, 0, 0);
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f4v c[2][2] = {};
        f4v d[2][2] = {};
#pragma unroll 4
        for (int it = 0; it < kIters; ++it) {
            if constexpr (LDS) {
                a0 = mine[0 * 64 + lane]; a1 = mine[1 * 64 + lane]; b0 = mine[2 * 64 + lane]; b1 = mine[3 * 64 + lane];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            }
            // eight 16x16x32 = the FLOPs of four 32x32x16
            c[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0][0], 0, 0, 0);
            c[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[0][1], 0, 0, 0);
            c[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[1][0], 0, 0, 0);
            c[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[1][1], 0, 0, 0);
            d[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a0, d[0][0], 0, 0, 0);
            d[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a1, d[0][1], 0, 0, 0);
            d[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a0, d[1][0], 0, 0, 0);
            d[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a1, d[1][1], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int r = 0; r < 4; ++r) s += c[i][j][r] + d[i][j][r];
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        stamps[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 0] = t1 - t0;
        stamps[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE, bool LDS>
static void run(const char *name, int waves_per_simd, bool zero, half8_t *src, float *sink, unsigned long long *stamps, int n_cu)
{
    const int threads = waves_per_simd * 4 * 64, blocks = n_cu;
    std::vector<_Float16> h((size_t)blocks * 64 * 4 * 8);
    srand(1);
    for (auto &x : h) x = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k_mfma<SHAPE, LDS>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps);
    CHECK(hipDeviceSynchronize());
    const int launches = 400;
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < launches; ++w) hipLaunchKernelGGL((k_mfma<SHAPE, LDS>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = blocks * waves_per_simd * 4;
    std::vector<unsigned long long> st((size_t)waves * 2);
    CHECK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int w = 0; w < waves; ++w) { cyc.push_back((double)st[w * 2]); clk.push_back((double)st[w * 2] / (double)st[w * 2 + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop_per_group = 4.0 * 32 * 32 * 16 * 2;   // both shapes: 4 x 32x32x16 = 8 x 16x16x32
    const double flops = (double)launches * waves * kIters * flop_per_group;
    const double mfma32_equiv = (double)kIters * 4 * waves_per_simd;   // 32x32x16-equivalents per SIMD and launch
    printf("%-44s %d wave(s)/SIMD %-6s  %8.1f TFLOP/s   %6.2f cycles per 32x32x16-equivalent per SIMD   clock %5.0f MHz\n", name, waves_per_simd,
           zero ? "zeros" : "random", flops / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2] / mfma32_equiv, clk[clk.size() / 2]);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("# %s, %d CUs; peak at 2.4 GHz = %.0f TFLOP/s dense fp16\n", p.name, n_cu, n_cu * 4 * 1024.0 * 2.4e9 / 1e12);
    half8_t *src; float *sink; unsigned long long *stamps;
    CHECK(hipMalloc(&src, (size_t)n_cu * 64 * 4 * 16));
    CHECK(hipMalloc(&sink, (size_t)n_cu * 1024 * 4));
    CHECK(hipMalloc(&stamps, (size_t)n_cu * 16 * 2 * 8));
    for (int zero = 0; zero < 2; ++zero)
        for (int w : {1, 2, 4}) {
            run<32, false>("v_mfma_f32_32x32x16_f16, operands in registers", w, zero, src, sink, stamps, n_cu);
            run<16, false>("v_mfma_f32_16x16x32_f16, operands in registers", w, zero, src, sink, stamps, n_cu);
        }
    for (int w : {1, 2, 4}) {
        run<32, true>("v_mfma_f32_32x32x16_f16, operands from LDS", w, false, src, sink, stamps, n_cu);
        run<16, true>("v_mfma_f32_16x16x32_f16, operands from LDS", w, false, src, sink, stamps, n_cu);
    }
    return 0;
}
