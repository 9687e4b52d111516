// Microbenchmark (GPU box): cycles per v_mfma_f32_16x16x32_f16 with NV VALU instructions between consecutive MFMAs, NC accumulator chains,
// one wavefront per SIMD (256 workgroups x 256 threads).  KIND 0: independent v_add_f32 fillers; KIND 1: fillers that read the accumulator
// of the OTHER chain (finished long ago: v_max_f32 of its registers).  Build twice: plain and with -mllvm -amdgpu-mfma-vgpr-form.
// usage: ./mfma_valu            (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));

template <int NC, int NV, int KIND>
__global__ __launch_bounds__(256) void k(const half8_t *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ cyc, int iters)
{
    const int tid = threadIdx.x;
    half8_t a = in[tid], b = in[tid + 256];
    acc_t acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = acc_t{0.f, 0.f, 0.f, 0.f};
    float f[4] = {1.f, 2.f, 3.f, 4.f};
    acc_t old = acc_t{1.f, 2.f, 3.f, 4.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    acc_t idle = acc_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2) idle = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, idle, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v & 3]) : "v"(f[(v + 1) & 3]));
                    else if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[v & 3]) : "v"(old[v & 3]));
                    else asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[v & 3]) : "v"(idle[v & 3]));    // KIND 2: the destination registers of an MFMA issued once per 16 x NC MFMAs
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (KIND == 1) old = acc[0];          // (a value an MFMA wrote, read by the fillers of the next iteration)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f[0] + f[1] + f[2] + f[3] + idle[1];
#pragma unroll
    for (int c = 0; c < NC; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NC, int NV, int KIND>
static double run(const half8_t *in, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NC, NV, KIND>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((k<NC, NV, KIND>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / 256 / (iters * 16.0 * NC);
}

int main()
{
    half8_t *in; float *out; unsigned long long *cyc;
    hipMalloc(&in, 512 * sizeof(half8_t)); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("cycles per v_mfma_f32_16x16x32_f16 (one wavefront per SIMD); rows: accumulator chains, columns: VALU fillers per MFMA 0 1 2 3\n");
#define ROW(NC, KIND) printf("chains %d, %s: %6.2f %6.2f %6.2f %6.2f\n", NC, KIND == 2 ? "fillers read an idle MFMA destination" : KIND ? "fillers read an MFMA result" : "independent fillers      ", \
    run<NC, 0, KIND>(in, out, cyc), run<NC, 1, KIND>(in, out, cyc), run<NC, 2, KIND>(in, out, cyc), run<NC, 3, KIND>(in, out, cyc));
    ROW(1, 0) ROW(2, 0) ROW(4, 0) ROW(2, 1) ROW(4, 1) ROW(2, 2) ROW(4, 2)
    return 0;
}
