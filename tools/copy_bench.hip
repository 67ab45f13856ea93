// Micro-benchmark (GPU box): what does a frame -> padded-plane copy of the import kernel's shape
// reach with different per-lane widths, rows per thread and store flavours?  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int W = 640, H = 480, DSTRIDE = 768, DROWS = 518, PADL = 32, EDGE = 19;
constexpr size_t FRAME_SRC = (size_t)W * H, FRAME_DST = 1204224;  // like the real pyramid frame pitch

template <int BYTES, int ROWS, bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst)
{
    constexpr int PIECES = W / BYTES;
    const int frame = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int chunk = t / PIECES, piece = t - chunk * PIECES;
    const int row0 = chunk * ROWS;
    if (row0 >= H) return;
    const uint8_t *s = src + frame * FRAME_SRC + (size_t)row0 * W + piece * BYTES;
    uint8_t *d = dst + frame * FRAME_DST + (size_t)(row0 + EDGE) * DSTRIDE + PADL + piece * BYTES;
    if constexpr (BYTES == 16) {
        u4v v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) v[r] = *reinterpret_cast<const u4v *>(s + r * W);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if constexpr (NT) __builtin_nontemporal_store(v[r], reinterpret_cast<u4v *>(d + r * DSTRIDE));
            else *reinterpret_cast<u4v *>(d + r * DSTRIDE) = v[r];
        }
    } else {
        uint32_t v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) v[r] = *reinterpret_cast<const uint32_t *>(s + r * W);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if constexpr (NT) __builtin_nontemporal_store(v[r], reinterpret_cast<uint32_t *>(d + r * DSTRIDE));
            else *reinterpret_cast<uint32_t *>(d + r * DSTRIDE) = v[r];
        }
    }
}

// persistent: a fixed number of work-groups walks all (frame, row) pairs, one 16-byte piece per lane, 4 rows in flight
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_persist(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int nFrames)
{
    constexpr int PIECES = W / 16;               // 40 lanes per row
    constexpr int ROWS_PER_WG = 256 / PIECES;    // 6 rows per work-group pass (240 lanes active)
    const int lane = threadIdx.x;
    const int rsub = lane / PIECES, piece = lane - rsub * PIECES;
    if (rsub >= ROWS_PER_WG) return;
    const int totalRows = nFrames * H;
    for (int base = blockIdx.x * ROWS_PER_WG * 4; base < totalRows; base += gridDim.x * ROWS_PER_WG * 4) {
        u4v v[4];
        int fr[4], ro[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int gr = min(base + k * ROWS_PER_WG + rsub, totalRows - 1);
            fr[k] = gr / H; ro[k] = gr - fr[k] * H;
            v[k] = *reinterpret_cast<const u4v *>(src + fr[k] * FRAME_SRC + (size_t)ro[k] * W + piece * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u4v *p = reinterpret_cast<u4v *>(dst + fr[k] * FRAME_DST + (size_t)(ro[k] + EDGE) * DSTRIDE + PADL + piece * 16);
            if constexpr (NT) __builtin_nontemporal_store(v[k], p); else *p = v[k];
        }
    }
}

template <typename F>
float timeit(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main()
{
    const int N = 256;
    uint8_t *src, *dst;
    CK(hipMalloc(&src, N * FRAME_SRC));
    CK(hipMalloc(&dst, N * FRAME_DST));
    CK(hipMemset(src, 7, N * FRAME_SRC));
    CK(hipMemset(dst, 0, N * FRAME_DST));
    const double bytes = 2.0 * N * FRAME_SRC;
#define RUN(B, R, NT) { const int thr = (W / B) * ((H + R - 1) / R); dim3 g((thr + 255) / 256, 1, N); \
    float us = timeit([&] { hipLaunchKernelGGL((k_copy<B, R, NT>), g, dim3(256), 0, 0, src, dst); }); \
    printf("bytes/lane %2d rows %d nt %d: %7.1f us  %.2f TB/s\n", B, R, NT, us, bytes / us / 1e6); }
    RUN(4, 1, false) RUN(4, 4, false) RUN(4, 8, false) RUN(16, 1, false) RUN(16, 2, false) RUN(16, 4, false) RUN(16, 8, false)
    RUN(16, 1, true) RUN(16, 4, true) RUN(4, 4, true)
    for (int wg : {1024, 2048, 4096, 8192}) {
        float us = timeit([&] { hipLaunchKernelGGL((k_copy_persist<false>), dim3(wg), dim3(256), 0, 0, src, dst, N); });
        printf("persistent %5d WGs:        %7.1f us  %.2f TB/s\n", wg, us, bytes / us / 1e6);
        us = timeit([&] { hipLaunchKernelGGL((k_copy_persist<true>), dim3(wg), dim3(256), 0, 0, src, dst, N); });
        printf("persistent %5d WGs nt:     %7.1f us  %.2f TB/s\n", wg, us, bytes / us / 1e6);
    }
    float us = timeit([&] { (void)hipMemcpyAsync(dst, src, N * FRAME_SRC, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D:               %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    return 0;
}
