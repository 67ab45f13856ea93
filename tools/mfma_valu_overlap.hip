// Micro-benchmark (gfx950): how many independent vector instructions hide behind one v_mfma_i32_32x32x32_i8 in ONE
// wave's instruction stream?  Two alternating accumulators, N x v_min3_u32 between consecutive MFMAs; B operand in
// AGPRs or VGPRs; accumulators in VGPRs.  One wave per SIMD (256 blocks x 256 threads).  Prints cycles per MFMA at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define VALU1 "v_min3_u32 %[x0], %[x0], %[y], %[z]\n"
#define VALU2 VALU1 "v_min3_u32 %[x1], %[x1], %[y], %[z]\n"
#define VALU4 VALU2 "v_med3_u32 %[x2], %[x2], %[y], %[z]\n" "v_lshl_add_u32 %[x3], %[x3], 16, %[z]\n"
template <int N, bool kAgprB>
__global__ __launch_bounds__(256) void k(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0}, c1 = {0};
    unsigned x0 = threadIdx.x, x1 = 7, x2 = 9, x3 = 11, y = 1000 + threadIdx.x, z = 55;
    if (kAgprB) asm volatile("" : "+a"(b));
    for (int i = 0; i < iters; i++) {
#define STEP(ACC)                                                                                                   \
        if (kAgprB) asm volatile("v_mfma_i32_32x32x32_i8 %[c], %[a], %[b], %[c]\n" : [c] "+v"(ACC) : [a] "v"(a), [b] "a"(b)); \
        else asm volatile("v_mfma_i32_32x32x32_i8 %[c], %[a], %[b], %[c]\n" : [c] "+v"(ACC) : [a] "v"(a), [b] "v"(b));        \
        if (N >= 2) asm volatile(VALU2 : [x0] "+v"(x0), [x1] "+v"(x1) : [y] "v"(y), [z] "v"(z));                   \
        if (N >= 4) asm volatile(VALU2 : [x0] "+v"(x2), [x1] "+v"(x3) : [y] "v"(y), [z] "v"(z));                   \
        if (N >= 6) asm volatile(VALU2 : [x0] "+v"(x0), [x1] "+v"(x1) : [y] "v"(y), [z] "v"(z));                   \
        if (N >= 8) asm volatile(VALU2 : [x0] "+v"(x2), [x1] "+v"(x3) : [y] "v"(y), [z] "v"(z));                   \
        if (N >= 10) asm volatile(VALU2 : [x0] "+v"(x0), [x1] "+v"(x1) : [y] "v"(y), [z] "v"(z));                  \
        if (N >= 12) asm volatile(VALU2 : [x0] "+v"(x2), [x1] "+v"(x3) : [y] "v"(y), [z] "v"(z));
        STEP(c0) STEP(c1) STEP(c0) STEP(c1)
    }
    int s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int N, bool kAgprB>
static void run(int *out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<N, kAgprB>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("N=%2d VALU per MFMA, B in %s, %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", N, kAgprB ? "AGPR" : "VGPR", waves_per_simd,
           ms * 1e-3 * 2.4e9 / (4.0 * iters * waves_per_simd));
}
int main()
{
    int *out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0, false>(out, 1); run<2, false>(out, 1); run<4, false>(out, 1); run<6, false>(out, 1); run<8, false>(out, 1); run<10, false>(out, 1); run<12, false>(out, 1);
    run<6, true>(out, 1); run<8, true>(out, 1);
    run<0, false>(out, 2); run<6, false>(out, 2); run<8, false>(out, 2); run<12, false>(out, 2);
    return 0;
}
