// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 on gfx950 over launches of increasing length (a few ms to ~0.2 s): does the
// chip hold the 157.3 TFLOP/s of its 2.4 GHz peak under a long fp32 MFMA load, or does the clock settle lower?  (The GEMM of
// amos_conv1x1.hip and MIOpen's fp32 convolutions both level off near 130 TFLOP/s.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f - (float)threadIdx.x * 1e-3f;
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    float *out;
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 2;  // 512: two waves per SIMD; 256: one (the same wave issues every MFMA of its SIMD)
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {2000, 20000, 100000, 400000}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 32. * 32 * 2 * 2 * 4.0 * iters * blocks * 4;
        printf("iters %7d: %8.3f ms, %.1f TFLOP/s (%.3f of 157.3)\n", iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
    return 0;
}
