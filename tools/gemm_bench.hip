// tools/gemm_bench.hip -- the 1 x 1 convolution GEMM (amos-slam_amd/csrc/amos_conv1x1.hip, compiled into this program) on one layer shape, launched
// directly: with 0 or more bytes of extra dynamic LDS (enough extra LDS leaves ONE work-group per CU instead of two: what does a group do alone?).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o gemm_bench tools/gemm_bench.hip
//   ./gemm_bench M K N [residual 0/1] [extra LDS bytes]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace amos { void set_error(const char *fmt, ...); }
#include "../amos-slam_amd/csrc/amos_conv1x1.hip"
namespace amos { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 78400, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 1024;
    const int withRes = argc > 4 ? atoi(argv[4]) : 1, extra = argc > 5 ? atoi(argv[5]) : 0;
    float *x, *w, *bias, *res, *y;
    hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&bias, N * 4); hipMalloc(&res, (size_t)M * N * 4); hipMalloc(&y, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * std::max(K, N));
    unsigned s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.f - 1.f; }
    hipMemcpy(x, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(res, h.data(), (size_t)M * N * 4, hipMemcpyHostToDevice);
    ConvGemmArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = withRes ? res : nullptr; a.y = y;
    a.M = M; a.N = N; a.K = K; a.outW = M; a.outHW = M; a.inW = M; a.inH = 1; a.stride = 1; a.relu = 1; a.kh = 1; a.kw = 1; a.pad = 0;
    a.mTiles = (M + 127) / 128; a.nTiles = N / 128; a.splits = 1; a.stagesPerSplit = 0; a.partial = nullptr; a.counters = nullptr;
    const dim3 grid((unsigned)(((a.mTiles + 7) / 8) * 8 * a.nTiles)), block(256);
    if (extra) hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_gemm<2, 2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, extra);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_conv_gemm<2, 2, 2, false>), grid, block, extra, 0, a);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("M %d K %d N %d res %d extra LDS %d: %.1f us per launch, %.1f TFLOP/s, %.2f TB/s\n", M, K, N, withRes, extra, ms * 100.f,
               2.0 * M * K * N / (ms * 1e-4) / 1e12, ((double)M * K + (double)M * N * (1 + withRes)) * 4 / (ms * 1e-4) / 1e12);
    }
    return 0;
}
