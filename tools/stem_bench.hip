// tools/stem_bench.hip -- timing experiments on the stem kernel (amos-slam_amd/csrc/amos_stem.hip, compiled into this program with
// AMOS_STEM_EXP switches: wrong results, timing only).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DAMOS_STEM_EXP=<bits> -o stem_bench_<bits> tools/stem_bench.hip
//   ./stem_bench_<bits> [frames]      bits: 1 no MFMA phase, 2 no hand-over / pooling, 4 no patch load
#define AMOS_STEM_BENCH 1
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace amos { void set_error(const char *fmt, ...); }
#include "../amos-slam_amd/csrc/amos_stem.hip"
namespace amos { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = 550, W = 550;
    float *x, *w, *wp, *bias, *y;
    const size_t nx = (size_t)B * 3 * H * W, ny = (size_t)B * 138 * 138 * 64;
    hipMalloc(&x, nx * 4); hipMalloc(&w, 64 * 147 * 4); hipMalloc(&wp, amos_mask_stem_weight_floats() * 4); hipMalloc(&bias, 256); hipMalloc(&y, ny * 4);
    std::vector<float> h(nx);
    unsigned s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.f - 1.f; }
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 64 * 147 * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), 256, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    amos_mask_stem_weights_device(st, w, 147, 49, 7, 1, wp);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = B >= 32 ? 10 : 50;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, st);
        for (int i = 0; i < n; i++)
            if (amos_mask_stem_device(st, x, (long long)3 * H * W, (long long)H * W, W, 1, wp, bias, y, B, H, W) != 0) return 1;
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("EXP=%d frames=%d: %.1f us per launch\n", AMOS_STEM_EXP, B, ms * 1e3f / n);
    }
    return 0;
}
