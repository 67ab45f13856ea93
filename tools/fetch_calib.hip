// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths
// this project's kernels use (MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of a 16 B/lane stream;
// other widths must be calibrated on a known byte count).  Streams a 1 GiB buffer (beyond the 256 MiB
// Infinity Cache) with 4 B/lane and 16 B/lane loads and writes 256 MiB with 4 B/lane stores.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read4(const unsigned *p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read16(const uint4 *p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void write4(unsigned *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}

int main()
{
    const size_t bytes = 1ull << 30;
    unsigned *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    read4<<<2048, 256>>>(buf, bytes / 4, sink);
    hipDeviceSynchronize();
    read16<<<2048, 256>>>((const uint4 *)buf, bytes / 16, sink);
    hipDeviceSynchronize();
    write4<<<2048, 256>>>(buf, (256ull << 20) / 4);
    hipDeviceSynchronize();
    printf("read4/read16: %zu bytes each; write4: %zu bytes\n", bytes, (size_t)(256ull << 20));
    return 0;
}
