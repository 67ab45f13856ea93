// amos_slic.hip -- cluster::SLIC of the reference (src/cluster.cc:300-343 with initilizeCenters :212-244,
// fituneCenter :246-298, clustering :88-158, updateCenter :160-210) on the device, from the Lab image on:
// the cv::cvtColor(BGR2Lab) in front of it (cluster.cc:310) is OpenCV's table-driven 8-bit conversion and stays
// with the caller; the k-means that follows (cluster.cc:345-464) reads
// uninitialised memory and libc rand() and stays there as well.
//
// The sequential "for every centre, for every pixel of its 2 len x 2 len window: if (dis < disMask) take it" of
// clustering() keeps, per pixel, the centre of minimal distance, the EARLIEST centre on ties; that is two
// atomic-min passes (distance bits, then centre index among the equals).  All arithmetic is double as in the
// reference, written without fused multiply-adds: sqrt(pow(disc,2) + m*pow(diss,2)) = sqrt(disc*disc + m*(diss*diss)).
#include "amos_common.h"

#include <vector>

namespace amos {

struct SlicGeom {
    int w, h, nx, ny, len, m;
};

__device__ __forceinline__ int refl101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// 0.5 * Sobel(dy=1) + 0.5 * Sobel(dx=1) of channel c at (y, x), CV_64F, BORDER_REFLECT_101 (cluster.cc:314-316)
__device__ __forceinline__ double sobel_half_sum(const uint8_t *lab, int w, int h, int y, int x, int c)
{
    const int ym = refl101(y - 1, h), yp = refl101(y + 1, h), xm = refl101(x - 1, w), xp = refl101(x + 1, w);
    auto P = [&](int yy, int xx) { return (int)lab[((size_t)yy * w + xx) * 3 + c]; };
    const int gy = (P(yp, xm) + 2 * P(yp, x) + P(yp, xp)) - (P(ym, xm) + 2 * P(ym, x) + P(ym, xp));   // dx = 0, dy = 1: "sobelImagex"
    const int gx = (P(ym, xp) + 2 * P(y, xp) + P(yp, xp)) - (P(ym, xm) + 2 * P(y, xm) + P(yp, xm));   // dx = 1, dy = 0: "sobelImagey"
    return __dadd_rn(__dmul_rn((double)gy, 0.5), __dmul_rn((double)gx, 0.5));
}

// initilizeCenters + fituneCenter: one thread per centre.  grid = (ceil(nx * ny / 256), frames)
__global__ __launch_bounds__(256) void k_slic_init(const uint8_t *__restrict__ lab, const uint16_t *__restrict__ depth, SlicGeom g,
                                                  amos_slic_center *__restrict__ centers)
{
    const int ck = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (ck >= g.nx * g.ny) return;
    lab += (size_t)frame * g.w * g.h * 3;
    depth += (size_t)frame * g.w * g.h;
    const int iy = ck / g.nx, ix = ck - iy * g.nx;
    amos_slic_center c;
    c.y = iy * g.len + g.len / 2;
    c.x = ix * g.len + g.len / 2;
    const uint8_t *p = lab + ((size_t)c.y * g.w + c.x) * 3;
    c.L = p[0]; c.A = p[1]; c.B = p[2];
    c.label = ck + 1;
    c.D = depth[(size_t)c.y * g.w + c.x];
    c.id = 0;
    if (!(c.x - 1 < 0 || c.x + 1 >= g.w || c.y - 1 < 0 || c.y + 1 >= g.h)) {  // cluster.cc:259-263
        double minGradient = 9999999;
        int tempx = 0, tempy = 0;
        for (int m = -1; m < 2; m++)
            for (int n = -1; n < 2; n++) {
                const double s0 = sobel_half_sum(lab, g.w, g.h, c.y + m, c.x + n, 0), s1 = sobel_half_sum(lab, g.w, g.h, c.y + m, c.x + n, 1),
                             s2 = sobel_half_sum(lab, g.w, g.h, c.y + m, c.x + n, 2);
                const double gradient = __dadd_rn(__dadd_rn(__dmul_rn(s0, s0), __dmul_rn(s1, s1)), __dmul_rn(s2, s2));
                if (gradient < minGradient) { minGradient = gradient; tempy = m; tempx = n; }
            }
        c.x += tempx;
        c.y += tempy;
        const uint8_t *q = lab + ((size_t)c.y * g.w + c.x) * 3;
        c.L = q[0]; c.A = q[1]; c.B = q[2];  // D keeps the depth of the grid position (cluster.cc:289-294)
    }
    centers[(size_t)frame * g.nx * g.ny + ck] = c;
}

__device__ __forceinline__ double slic_dist(const SlicGeom &g, const amos_slic_center &c, const uint8_t *px, int i, int j)
{
    const int dL = (int)px[0] - c.L, dA = (int)px[1] - c.A, dB = (int)px[2] - c.B;
    const double disc = __dsqrt_rn((double)(dL * dL + dA * dA + dB * dB));
    const double diss = __dsqrt_rn((double)((j - c.x) * (j - c.x) + (i - c.y) * (i - c.y)));
    return __dsqrt_rn(__dadd_rn(__dmul_rn(disc, disc), __dmul_rn((double)g.m, __dmul_rn(diss, diss))));
}

// clustering(), pass kPass = 0: disMask = min over covering centres (as ordered bits of a non-negative double);
// pass 1: ckMap = smallest centre index among those reaching that minimum.  One wave per centre, the lanes walk its
// 2 len x 2 len window.  grid = (ceil(ncent / 4), frames), block = 256.
template <int kPass>
__global__ __launch_bounds__(256) void k_slic_assign(const uint8_t *__restrict__ lab, SlicGeom g, const amos_slic_center *__restrict__ centers,
                                                    unsigned long long *__restrict__ disMask, int *__restrict__ ckMap)
{
    const int lane = threadIdx.x & 63, ck = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
    const int ncent = g.nx * g.ny;
    if (ck >= ncent) return;
    const size_t fo = (size_t)frame * g.w * g.h;
    lab += fo * 3;
    const amos_slic_center c = centers[(size_t)frame * ncent + ck];
    const int side = 2 * g.len;
    for (int e = lane; e < side * side; e += 64) {
        const int i = c.y - g.len + e / side, j = c.x - g.len + e % side;
        if (i < 0 || i >= g.h || j < 0 || j >= g.w) continue;
        const size_t p = (size_t)i * g.w + j;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(slic_dist(g, c, lab + p * 3, i, j));
        if (kPass == 0) atomicMin(&disMask[fo + p], bits);
        else if (bits == disMask[fo + p]) atomicMin(&ckMap[fo + p], ck);
    }
}

// labelMask(i, j) = label of the winning centre where any window covered the pixel this iteration; pixels no window
// reached keep their previous value (the reference allocates labelMask once, cluster.cc:323).  One thread per pixel.
__global__ __launch_bounds__(256) void k_slic_label(SlicGeom g, const amos_slic_center *__restrict__ centers, const int *__restrict__ ckMap,
                                                   double *__restrict__ labelMask)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int frame = blockIdx.y;
    if (p >= (size_t)g.w * g.h) return;
    const size_t fo = (size_t)frame * g.w * g.h;
    const int ck = ckMap[fo + p];
    if (ck != 0x7fffffff) labelMask[fo + p] = (double)centers[(size_t)frame * g.nx * g.ny + ck].label;
}

// updateCenter(): one wave per centre; sums of integers in double are exact in any order.
__global__ __launch_bounds__(256) void k_slic_update(const uint8_t *__restrict__ lab, const uint16_t *__restrict__ depth, SlicGeom g,
                                                    const double *__restrict__ labelMask, amos_slic_center *__restrict__ centers)
{
    const int lane = threadIdx.x & 63, ck = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
    const int ncent = g.nx * g.ny;
    if (ck >= ncent) return;
    const size_t fo = (size_t)frame * g.w * g.h;
    amos_slic_center c = centers[(size_t)frame * ncent + ck];
    const int side = 2 * g.len;
    long long sx = 0, sy = 0, sL = 0, sA = 0, sB = 0, sD = 0, sN = 0;
    for (int e = lane; e < side * side; e += 64) {
        const int i = c.y - g.len + e / side, j = c.x - g.len + e % side;
        if (i < 0 || i >= g.h || j < 0 || j >= g.w) continue;
        const size_t p = (size_t)i * g.w + j;
        if (labelMask[fo + p] == (double)c.label) {
            const uint8_t *px = lab + (fo + p) * 3;
            sL += px[0]; sA += px[1]; sB += px[2];
            sx += j; sy += i; sN += 1;
            sD += depth[fo + p];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sL += __shfl_xor(sL, off, 64); sA += __shfl_xor(sA, off, 64);
        sB += __shfl_xor(sB, off, 64); sD += __shfl_xor(sD, off, 64); sN += __shfl_xor(sN, off, 64);
    }
    if (lane == 0) {
        const double num = sN == 0 ? 0.000000001 : (double)sN;  // cluster.cc:199
        c.x = (int)__ddiv_rn((double)sx, num);
        c.y = (int)__ddiv_rn((double)sy, num);
        c.L = (int)__ddiv_rn((double)sL, num);
        c.A = (int)__ddiv_rn((double)sA, num);
        c.B = (int)__ddiv_rn((double)sB, num);
        c.D = (int)__ddiv_rn((double)sD, num);
        centers[(size_t)frame * ncent + ck] = c;
    }
}

template <typename T>
__global__ void k_fill(T *p, size_t n, T v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}


// ---- cluster::randCent + cluster::kmeans (src/cluster.cc:353-460): the k-means over the SLIC centres that assigns every
// superpixel its cluster id.  One work-group per frame; the centres stay in registers (kKmPer per thread), the k
// centroids in LDS; a pass = nearest centroid of every centre (distEclud :374-387 in doubles with explicit
// round-to-nearest operations: |dD| / 20000 + sqrt(dx^2 + dy^2) / 800, strict < keeps the first centroid), then the
// integer means of x, y, D per cluster (wave-level sums, one LDS atomic per wave and cluster), until no assignment
// changes.  The reference's undefined behaviours are defined as in the oracle (seeded glibc TYPE_0 generator, index
// rowLen wraps to 0, bounded redraw, zero-initialised accumulator): DESIGN.md section 7.
constexpr int kKmThreads = 1024, kKmPer = 16, kKmMaxK = 64;

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kKmThreads) void k_kmeans(amos_slic_center *__restrict__ centers, int n, size_t frameStride, int k, uint32_t seed,
                                                      int maxIter, int *__restrict__ passesOut)
{
    __shared__ int cx[kKmMaxK], cy[kKmMaxK], cd[kKmMaxK], sx[kKmMaxK], sy[kKmMaxK], sd[kKmMaxK], cnt[kKmMaxK], changed[2];
    amos_slic_center *C = centers + (size_t)blockIdx.x * frameStride;
    const int tid = threadIdx.x;
    if (tid == 0) {  // randCent, sequential like the reference
        uint32_t state = seed;
        auto draw = [&]() { state = state * 1103515245u + 12345u; int idx = (int)((state & 0x7fffffffu) % (uint32_t)n) + 1; return idx >= n ? 0 : idx; };
        for (int i = 0; i < k; i++) {
            int idx = draw();
            for (int tries = 0; C[idx].D <= 0 && tries < 4 * n; tries++) idx = draw();
            cx[i] = C[idx].x; cy[i] = C[idx].y; cd[i] = C[idx].D;
        }
        changed[0] = changed[1] = 0;
    }
    int px[kKmPer], py[kKmPer], pd[kKmPer], as[kKmPer];
#pragma unroll
    for (int j = 0; j < kKmPer; j++) {
        const int i = tid + j * kKmThreads;
        px[j] = i < n ? C[i].x : 0; py[j] = i < n ? C[i].y : 0; pd[j] = i < n ? C[i].D : 0;
        as[j] = -1;
    }
    __syncthreads();
    int passes = 0;
    for (;;) {
        if (passes >= maxIter) { passes = -1; break; }
        passes++;
        if (tid < k) { sx[tid] = sy[tid] = sd[tid] = cnt[tid] = 0; }
        int mine = 0;
#pragma unroll
        for (int j = 0; j < kKmPer; j++) {
            if (tid + j * kKmThreads < n) {
                int minIndex = -1;
                double minDist = 2147483647.0;
                for (int c = 0; c < k; c++) {
                    const double sumD = __ddiv_rn((double)abs(pd[j] - cd[c]), 20000.0);
                    const int dx = cx[c] - px[j], dy = cy[c] - py[j];
                    const double sumE = __ddiv_rn(__dsqrt_rn((double)(dx * dx + dy * dy)), 800.0);  // sqrt(640^2 + 480^2) = 800
                    const double dist = __dadd_rn(sumE, sumD);
                    if (dist < minDist) { minDist = dist; minIndex = c; }
                }
                if (as[j] != minIndex) { mine = 1; as[j] = minIndex; }
            }
        }
        if (mine) changed[passes & 1] = 1;  // any writer: the flag of this pass
        __syncthreads();
        for (int c = 0; c < k; c++) {  // integer sums per cluster
            int a = 0, b = 0, d = 0, m = 0;
#pragma unroll
            for (int j = 0; j < kKmPer; j++)
                if (as[j] == c) { a += px[j]; b += py[j]; d += pd[j]; m++; }
            a = wave_sum(a); b = wave_sum(b); d = wave_sum(d); m = wave_sum(m);
            if ((tid & 63) == 0 && m) { atomicAdd(&sx[c], a); atomicAdd(&sy[c], b); atomicAdd(&sd[c], d); atomicAdd(&cnt[c], m); }
        }
        __syncthreads();
        const int any = changed[passes & 1];
        if (tid < k) {
            const int m = cnt[tid];
            cx[tid] = m ? sx[tid] / m : 0; cy[tid] = m ? sy[tid] / m : 0; cd[tid] = m ? sd[tid] / m : 0;
        }
        if (tid == 0) changed[(passes + 1) & 1] = 0;
        __syncthreads();
        if (!any) break;
    }
#pragma unroll
    for (int j = 0; j < kKmPer; j++) {
        const int i = tid + j * kKmThreads;
        if (i < n) {
            const int lab = C[i].label;  // 1 .. n as the SLIC entry points write them; anything else is left alone
            if (lab >= 1 && lab <= n) C[lab - 1].id = as[j];
        }
    }
    if (tid == 0 && passesOut) passesOut[blockIdx.x] = passes;
}


// ---- cv::cvtColor(image, imageLAB, COLOR_BGR2Lab) on 8-bit pixels (src/cluster.cc:310), OpenCV 4.5's fixed-point RGB2Lab_b:
// sRGB gamma table (256 entries, 3 extra bits), 12-bit XYZ / white-point coefficients, cube-root table (3072 entries, 15
// fractional bits), L = (296 fY - 1336934) >> 15 etc. with rounding, saturated.  Tables are built on the host in double
// (OpenCV builds them with its softfloat pow / cbrt: PARITY UNPINNED at the level of single table entries); the primaries and
// grays come out at OpenCV's documented values (blue 82/207/20, green 224/42/211, red 136/208/195, white 255/128/128).
// grid = ceil(n / 256), block 256: one pixel per thread.
struct LabTables {
    const unsigned short *gamma, *cbrt;
    int c[9];  // by (R, G, B) for X, Y, Z
};

__global__ __launch_bounds__(256) void k_bgr2lab(const uint8_t *__restrict__ src, size_t n, int blueIdx, const LabTables t, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int R = t.gamma[src[3 * i + (blueIdx ^ 2)]], G = t.gamma[src[3 * i + 1]], B = t.gamma[src[3 * i + blueIdx]];
    const int fX = t.cbrt[(R * t.c[0] + G * t.c[1] + B * t.c[2] + 2048) >> 12];
    const int fY = t.cbrt[(R * t.c[3] + G * t.c[4] + B * t.c[5] + 2048) >> 12];
    const int fZ = t.cbrt[(R * t.c[6] + G * t.c[7] + B * t.c[8] + 2048) >> 12];
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    auto sat = [](int v) { return (uint8_t)min(max(v, 0), 255); };
    dst[3 * i] = sat((Lscale * fY + Lshift + (1 << 14)) >> 15);
    dst[3 * i + 1] = sat((500 * (fX - fY) + 128 * (1 << 15) + (1 << 14)) >> 15);
    dst[3 * i + 2] = sat((200 * (fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15);
}

}  // namespace amos

using namespace amos;

struct amos_slic {
    int device = 0, maxW = 0, maxH = 0, maxB = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    unsigned long long *dDis = nullptr;
    int *dCk = nullptr;
    // host-call staging
    uint8_t *dLab = nullptr;
    uint16_t *dDepth = nullptr;
    double *dLabels = nullptr;
    amos_slic_center *dCenters = nullptr;
    int *dPasses = nullptr;
    unsigned short *dLabTabs = nullptr;  // gamma [256] then cube root [3072]
    int labCoeffs[9] = {0};
};

extern "C" {

int amos_slic_center_count(int width, int height, int len, int *nx, int *ny)
{
    if (width < 1 || height < 1 || len < 1) { set_error("amos_slic_center_count: invalid argument"); return AMOS_ERR_INVALID; }
    int cx = 0, cy = 0;
    for (int i = 0; i < height; i += len) if (i + len / 2 < height) cy++;   // cluster.cc:224-228
    for (int j = 0; j < width; j += len) if (j + len / 2 < width) cx++;
    if (nx) *nx = cx;
    if (ny) *ny = cy;
    return cx * cy;
}

int amos_slic_create(int device, void *stream, int max_width, int max_height, int max_batch, amos_slic **out)
{
    if (!out || max_width < 3 || max_height < 3 || max_batch < 1) { set_error("amos_slic_create: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_slic *s = new amos_slic();
    s->device = device; s->maxW = max_width; s->maxH = max_height; s->maxB = max_batch;
    if (stream) s->stream = (hipStream_t)stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete s; return AMOS_ERR_DEVICE; }
        s->ownStream = true;
    }
    const size_t px = (size_t)max_width * max_height, B = (size_t)max_batch;
    hipError_t e = hipMalloc((void **)&s->dDis, sizeof(unsigned long long) * px * B);
    if (e == hipSuccess) e = hipMalloc((void **)&s->dCk, sizeof(int) * px * B);
    if (e == hipSuccess) e = hipMalloc((void **)&s->dLab, px * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&s->dDepth, px * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&s->dLabels, px * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->dCenters, sizeof(amos_slic_center) * px);  // len >= 1: at most one centre per pixel
    if (e != hipSuccess) { set_error("amos_slic_create: %s", hipGetErrorString(e)); amos_slic_destroy(s); return AMOS_ERR_DEVICE; }
    *out = s;
    return AMOS_OK;
}

void amos_slic_destroy(amos_slic *s)
{
    if (!s) return;
    if (s->dLabTabs) (void)hipFree(s->dLabTabs);
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    void *ptrs[] = {s->dDis, s->dCk, s->dLab, s->dDepth, s->dLabels, s->dCenters};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (s->ownStream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

void *amos_slic_stream(amos_slic *s) { return s ? (void *)s->stream : nullptr; }

int amos_slic_batch_device(amos_slic *s, const uint8_t *d_lab, const uint16_t *d_depth, int width, int height, int n_frames, int len, int m,
                           int iterations, double *d_labels, amos_slic_center *d_centers)
{
    if (!s || !d_lab || !d_depth || !d_labels || !d_centers || n_frames < 1 || len < 1 || iterations < 0 || width < 3 || height < 3) {
        set_error("amos_slic_batch_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (width > s->maxW || height > s->maxH || (size_t)width * height > (size_t)s->maxW * s->maxH || n_frames > s->maxB) {
        set_error("amos_slic_batch_device: %d frames of %dx%d exceed the handle's %d x %dx%d", n_frames, width, height, s->maxB, s->maxW, s->maxH);
        return AMOS_ERR_CAPACITY;
    }
    AMOS_HIP_CHECK(hipSetDevice(s->device));
    SlicGeom g;
    g.w = width; g.h = height; g.len = len; g.m = m;
    const int ncent = amos_slic_center_count(width, height, len, &g.nx, &g.ny);
    if (ncent < 1) { set_error("amos_slic_batch_device: no centre fits"); return AMOS_ERR_INVALID; }
    const size_t px = (size_t)width * height, tot = px * n_frames;
    AMOS_HIP_CHECK(hipMemsetAsync(d_labels, 0, sizeof(double) * tot, s->stream));  // labelMask = Mat::zeros, cluster.cc:323
    hipLaunchKernelGGL(k_slic_init, dim3((ncent + 255) / 256, n_frames), dim3(256), 0, s->stream, d_lab, d_depth, g, d_centers);
    const unsigned long long maxdis = (unsigned long long)0x412E847E00000000ULL;  // bits of 999999.0 (MAXDIS, cluster.cc:302)
    for (int it = 0; it < iterations; it++) {
        hipLaunchKernelGGL(k_fill<unsigned long long>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, s->dDis, tot, maxdis);
        hipLaunchKernelGGL(k_fill<int>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, s->dCk, tot, 0x7fffffff);
        hipLaunchKernelGGL(k_slic_assign<0>, dim3((ncent + 3) / 4, n_frames), dim3(256), 0, s->stream, d_lab, g, d_centers, s->dDis, s->dCk);
        hipLaunchKernelGGL(k_slic_assign<1>, dim3((ncent + 3) / 4, n_frames), dim3(256), 0, s->stream, d_lab, g, d_centers, s->dDis, s->dCk);
        hipLaunchKernelGGL(k_slic_label, dim3((unsigned)((px + 255) / 256), n_frames), dim3(256), 0, s->stream, g, d_centers, s->dCk, d_labels);
        hipLaunchKernelGGL(k_slic_update, dim3((ncent + 3) / 4, n_frames), dim3(256), 0, s->stream, d_lab, d_depth, g, d_labels, d_centers);
    }
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_slic_run(amos_slic *s, const uint8_t *lab, const uint16_t *depth, int width, int height, int len, int m, int iterations, double *labels,
              amos_slic_center *centers, int *n_centers)
{
    if (!s || !lab || !depth || !labels || !centers) { set_error("amos_slic_run: invalid argument"); return AMOS_ERR_INVALID; }
    if (width > s->maxW || height > s->maxH) { set_error("amos_slic_run: frame %dx%d exceeds the handle's %dx%d", width, height, s->maxW, s->maxH); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipSetDevice(s->device));
    const size_t px = (size_t)width * height;
    const int ncent = amos_slic_center_count(width, height, len, nullptr, nullptr);
    if (ncent < 1) { set_error("amos_slic_run: invalid size"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipMemcpyAsync(s->dLab, lab, px * 3, hipMemcpyHostToDevice, s->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(s->dDepth, depth, px * 2, hipMemcpyHostToDevice, s->stream));
    const int rc = amos_slic_batch_device(s, s->dLab, s->dDepth, width, height, 1, len, m, iterations, s->dLabels, s->dCenters);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipMemcpyAsync(labels, s->dLabels, px * 8, hipMemcpyDeviceToHost, s->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(centers, s->dCenters, sizeof(amos_slic_center) * ncent, hipMemcpyDeviceToHost, s->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (n_centers) *n_centers = ncent;
    return AMOS_OK;
}


int amos_cluster_kmeans_batch_device(amos_slic *s, amos_slic_center *d_centers, int n_centers, int n_frames, int k, uint32_t seed, int max_iter,
                                     int32_t *d_passes)
{
    if (!s || !d_centers || n_centers < 1 || n_frames < 1 || k < 1 || k > kKmMaxK || max_iter < 1 || n_centers > kKmThreads * kKmPer) {
        set_error("amos_cluster_kmeans_batch_device: invalid argument (1 <= k <= %d, centres <= %d)", kKmMaxK, kKmThreads * kKmPer);
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(s->device));
    hipLaunchKernelGGL(k_kmeans, dim3(n_frames), dim3(kKmThreads), 0, s->stream, d_centers, n_centers, (size_t)n_centers, k, seed, max_iter, d_passes);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_cluster_kmeans(amos_slic *s, amos_slic_center *centers, int n_centers, int k, uint32_t seed, int max_iter, int *passes)
{
    if (!s || !centers || n_centers < 1) { set_error("amos_cluster_kmeans: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(s->device));
    amos_slic_center *d = nullptr;
    AMOS_HIP_CHECK(hipMalloc((void **)&d, sizeof(amos_slic_center) * n_centers + sizeof(int)));
    int *dp = reinterpret_cast<int *>(d + n_centers);
    int rc = AMOS_OK, p = 0;
    if (hipMemcpyAsync(d, centers, sizeof(amos_slic_center) * n_centers, hipMemcpyHostToDevice, s->stream) != hipSuccess) rc = AMOS_ERR_DEVICE;
    if (rc == AMOS_OK) rc = amos_cluster_kmeans_batch_device(s, d, n_centers, 1, k, seed, max_iter, dp);
    if (rc == AMOS_OK && (hipMemcpyAsync(centers, d, sizeof(amos_slic_center) * n_centers, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                          hipMemcpyAsync(&p, dp, sizeof(int), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                          hipStreamSynchronize(s->stream) != hipSuccess)) {
        set_error("amos_cluster_kmeans: copy failed");
        rc = AMOS_ERR_DEVICE;
    }
    (void)hipFree(d);
    if (passes) *passes = p;
    return rc;
}


int amos_cluster_bgr2lab_batch_device(amos_slic *s, const uint8_t *d_bgr, size_t n_pixels, int rgb_order, uint8_t *d_lab)
{
    if (!s || !d_bgr || !d_lab) { set_error("amos_cluster_bgr2lab_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_pixels == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(s->device));
    if (!s->dLabTabs) {  // built once per handle, in double like the checker
        std::vector<unsigned short> tabs(256 + 3072);
        for (int i = 0; i < 256; i++) {
            const float x = (float)i / 255.f;
            const double g = x <= 0.04045f ? (double)x / 12.92 : std::pow(((double)x + 0.055) / 1.055, 2.4);
            tabs[i] = (unsigned short)std::lrint(255.0 * 8.0 * g);
        }
        for (int i = 0; i < 3072; i++) {
            const float x = (float)i / (255.f * 8.f);
            const double f = x < 0.008856f ? (double)x * 7.787 + 0.13793103448275862 : std::cbrt((double)x);
            tabs[256 + i] = (unsigned short)std::lrint(32768.0 * f);
        }
        static const double xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
        static const double white[3] = {0.950456, 1., 1.088754};
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) s->labCoeffs[3 * i + k] = (int)std::lrint(4096.0 * xyz[3 * i + k] / white[i]);
        AMOS_HIP_CHECK(hipMalloc((void **)&s->dLabTabs, sizeof(unsigned short) * tabs.size()));
        AMOS_HIP_CHECK(hipMemcpy(s->dLabTabs, tabs.data(), sizeof(unsigned short) * tabs.size(), hipMemcpyHostToDevice));
    }
    LabTables t;
    t.gamma = s->dLabTabs;
    t.cbrt = s->dLabTabs + 256;
    for (int i = 0; i < 9; i++) t.c[i] = s->labCoeffs[i];
    hipLaunchKernelGGL(k_bgr2lab, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s->stream, d_bgr, n_pixels, rgb_order ? 2 : 0, t, d_lab);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
