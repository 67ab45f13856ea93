// Micro-benchmark: issue rate of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 on gfx950 (one wave per SIMD,
// 4 independent accumulators), in cycles per instruction per SIMD from s_memtime and in TOP/s from wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int kShape>
__global__ __launch_bounds__(256) void k(int *out, long long *cyc, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v4i d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (kShape == 32) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; i++) s += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    int *out; long long *cyc;
    const int blocks = 256 * 2, iters = 20000;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    for (int shape : {32, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double ops = (shape == 32 ? 32. * 32 * 32 : 16. * 16 * 64) * 2 * 4.0 * iters * blocks * 4;
        printf("shape %d: %.3f ms, %.1f TOP/s; memtime ticks per MFMA (2 waves/SIMD resident: block pairs) %.2f\n", shape, ms, ops / ms / 1e9, (double)h / (4.0 * iters));
    }
    return 0;
}
