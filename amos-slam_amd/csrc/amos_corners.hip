// amos_corners.hip -- the corner source of Tracking::GetSceneFlowObj (src/Tracking.cc:894-895):
//     cv::goodFeaturesToTrack(imlast, prepoint, 1000, 0.01, 8, cv::Mat(), 3, true, 0.04);
//     cv::cornerSubPix(imlast, prepoint, cv::Size(10, 10), cv::Size(-1, -1), cv::TermCriteria(ITER | EPS, 20, 0.03));
// on the device, resident: the corners feed amos_lk_track_device without leaving HBM.  Restated from OpenCV 4.5's published
// implementation (featureselect.cpp, corner.cpp, cornersubpix.cpp, samplers.cpp) rule for rule like the project's CPU checker (whose
// header lists them); PARITY UNPINNED like every OpenCV-derived stage.  What the tests pin is GPU == CPU checker,
// bit for bit: responses, corner lists in order, refined positions.
//
//   k_harris            Sobel 3 x 3 (float, scaled smoothing taps) -> products -> 3 x 3 box sums (double, one rounding) -> response, and the
//                       frame's maximum (ordered-integer atomic); one 32 x 8 tile per work-group, gradients of the tile + 1 in LDS, the
//                       REFLECT_101 borders of BOTH filters applied where they belong (the box filter reflects the gradient PLANES)
//   k_corner_candidates threshold, 3 x 3 local maximum, append (value bits << 32 | pixel index) with a wave-aggregated atomic
//   k_corner_rank       order by (value, pixel index) descending: rank = number of larger keys (n <= 65 536: n^2 / 2^32 comparisons are
//                       microseconds here), scatter
//   k_corner_select     the greedy minimum-distance selection, which is sequential in the reference, as a fixed point: a candidate is
//                       kept iff no kept candidate of higher rank lies within minDistance; every round settles the candidates whose
//                       higher-ranked neighbours are all settled (one work-group, candidates bucketed in minDistance cells)
//   k_corner_subpix     one wave per corner: bilinear (2 win + 3)^2 patch in LDS, gradient products of the (2 win + 1)^2 window computed by all
//                       lanes, the five sums taken in the reference's order (row-major, double) by five lanes
#include <float.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

namespace amos {

constexpr int kCornerCandCap = 65536;
constexpr int kSubpixMaxWin = 15;

__device__ __forceinline__ int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// order-preserving map float -> unsigned (all floats), for the atomic maximum
__device__ __forceinline__ unsigned float_key(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __host__ __forceinline__ float key_float(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(b);
#else
    __builtin_memcpy(&f, &b, 4);
#endif
    return f;
}

constexpr int kHarrisTW = 32, kHarrisTH = 8;

__global__ __launch_bounds__(256) void k_harris(const uint8_t *__restrict__ img, size_t stride, int w, int h, float k1, float k2, float kf,
                                                float *__restrict__ eig, unsigned *__restrict__ maxKey)
{
    __shared__ float gx[kHarrisTH + 2][kHarrisTW + 2], gy[kHarrisTH + 2][kHarrisTW + 2];
    __shared__ unsigned blockMax;
    const int x0 = blockIdx.x * kHarrisTW, y0 = blockIdx.y * kHarrisTH, t = threadIdx.x;
    if (t == 0) blockMax = 0u;
    // gradients at the (reflected) positions of the tile and its one-pixel ring
    for (int idx = t; idx < (kHarrisTH + 2) * (kHarrisTW + 2); idx += 256) {
        const int ly = idx / (kHarrisTW + 2), lx = idx - ly * (kHarrisTW + 2);
        const int qy = refl101(min(y0 - 1 + ly, h), h), qx = refl101(min(x0 - 1 + lx, w), w);  // (positions past the image's far edge + 1 are never used)
        const uint8_t *r0 = img + (size_t)refl101(qy - 1, h) * stride, *r1 = img + (size_t)qy * stride, *r2 = img + (size_t)refl101(qy + 1, h) * stride;
        const int xm = refl101(qx - 1, w), xp = refl101(qx + 1, w);
        const float a0 = (float)((int)r0[xp] - (int)r0[xm]), a1 = (float)((int)r1[xp] - (int)r1[xm]), a2 = (float)((int)r2[xp] - (int)r2[xm]);
        gx[ly][lx] = __fadd_rn(__fmul_rn(__fadd_rn(a0, a2), k1), __fmul_rn(a1, k2));
        const float b0 = __fadd_rn(__fmul_rn((float)r0[qx], k2), __fmul_rn((float)((int)r0[xm] + (int)r0[xp]), k1));
        const float b2 = __fadd_rn(__fmul_rn((float)r2[qx], k2), __fmul_rn((float)((int)r2[xm] + (int)r2[xp]), k1));
        gy[ly][lx] = __fsub_rn(b2, b0);
    }
    __syncthreads();
    const int lx = t & 31, ly = t >> 5, x = x0 + lx, y = y0 + ly;
    if (x < w && y < h) {
        double sa = 0, sb = 0, sc = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double ra = 0, rb = 0, rc = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float dx = gx[ly + j][lx + i], dy = gy[ly + j][lx + i];
                ra = __dadd_rn(ra, (double)__fmul_rn(dx, dx));
                rb = __dadd_rn(rb, (double)__fmul_rn(dx, dy));
                rc = __dadd_rn(rc, (double)__fmul_rn(dy, dy));
            }
            sa = __dadd_rn(sa, ra);
            sb = __dadd_rn(sb, rb);
            sc = __dadd_rn(sc, rc);
        }
        const float a = (float)sa, b = (float)sb, c = (float)sc;
        const float tr = __fadd_rn(a, c);
        const float r = __fsub_rn(__fsub_rn(__fmul_rn(a, c), __fmul_rn(b, b)), __fmul_rn(__fmul_rn(kf, tr), tr));
        eig[(size_t)y * w + x] = r;
        atomicMax(&blockMax, float_key(r));
    }
    __syncthreads();
    if (t == 0) atomicMax(maxKey, blockMax);
}

struct CornerWork {
    unsigned *maxKey;                 // [1]
    int *count;                       // [4]: candidates, kept, unknown left, spare
    unsigned long long *keys, *sorted;  // [kCornerCandCap] each
    int *cellStart, *cellFill, *cellItems;  // cells + 1, cells, kCornerCandCap
    uint8_t *state;                   // [kCornerCandCap]
};

__global__ __launch_bounds__(256) void k_corner_candidates(const float *__restrict__ eig, int w, int h, double quality, CornerWork wk)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const float thr = (float)((double)key_float(*wk.maxKey) * quality);
    bool isCand = false;
    float v = 0.f;
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        v = eig[(size_t)y * w + x];
        v = v > thr ? v : 0.f;
        if (v != 0.f) {
            float m = v;
#pragma unroll
            for (int j = -1; j <= 1; j++)
#pragma unroll
                for (int i = -1; i <= 1; i++) {
                    float tv = eig[(size_t)(y + j) * w + x + i];
                    tv = tv > thr ? tv : 0.f;
                    m = tv > m ? tv : m;
                }
            isCand = v == m;
        }
    }
    const unsigned long long bal = __ballot(isCand);
    if (bal) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(&wk.count[0], __popcll(bal));
        base = __shfl(base, 0, 64);
        if (isCand) {
            const int o = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (o < kCornerCandCap) wk.keys[o] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(y * w + x);  // v > 0: the bits order like the value
        }
    }
}

// sorted[rank] = key, rank = number of larger keys (keys are distinct: the pixel index is part of them)
__global__ __launch_bounds__(256) void k_corner_rank(CornerWork wk)
{
    __shared__ unsigned long long tile[256];
    const int n = min(wk.count[0], kCornerCandCap), i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const unsigned long long me = i < n ? wk.keys[i] : 0ull;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        tile[threadIdx.x] = j0 + threadIdx.x < n ? wk.keys[j0 + threadIdx.x] : 0ull;
        __syncthreads();
        const int m = min(256, n - j0);
        for (int j = 0; j < m; j++) rank += tile[j] > me;
        __syncthreads();
    }
    if (i < n) wk.sorted[rank] = me;
}

// one work-group: cell lists, then the fixed point of "kept iff no kept candidate of higher rank within minDistance", then the first
// max_corners kept candidates in rank order
__global__ __launch_bounds__(1024) void k_corner_select(CornerWork wk, int w, int h, int cell, int gw, int gh, double md2, int maxCorners, int cap,
                                                       float *__restrict__ xyOut, int *__restrict__ countOut)
{
    __shared__ int sh[1024];
    __shared__ int shTotal, shUnknown;
    const int n = min(wk.count[0], kCornerCandCap), t = threadIdx.x, cells = gw * gh;
    for (int c = t; c <= cells; c += 1024) wk.cellStart[c] = 0;
    for (int c = t; c < cells; c += 1024) wk.cellFill[c] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int idx = (int)(wk.sorted[i] & 0xffffffffull), y = idx / w, x = idx - y * w;
        atomicAdd(&wk.cellStart[(y / cell) * gw + x / cell + 1], 1);
        wk.state[i] = 0;
    }
    __syncthreads();
    // inclusive scan of cellStart[1 .. cells] in chunks of 1024
    if (t == 0) shTotal = 0;
    __syncthreads();
    for (int c0 = 1; c0 <= cells; c0 += 1024) {
        const int c = c0 + t;
        int v = c <= cells ? wk.cellStart[c] : 0;
        sh[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += add;
            __syncthreads();
        }
        const int base = shTotal;
        if (c <= cells) wk.cellStart[c] = base + sh[t];
        __syncthreads();
        if (t == 1023) shTotal = base + sh[1023];
        __syncthreads();
    }
    for (int i = t; i < n; i += 1024) {
        const int idx = (int)(wk.sorted[i] & 0xffffffffull), y = idx / w, x = idx - y * w, c = (y / cell) * gw + x / cell;
        wk.cellItems[wk.cellStart[c] + atomicAdd(&wk.cellFill[c], 1)] = i;
    }
    __threadfence_block();
    __syncthreads();
    for (int round = 0; round < n + 1; round++) {
        if (t == 0) shUnknown = 0;
        __syncthreads();
        for (int i = t; i < n; i += 1024) {
            if (wk.state[i] != 0) continue;
            const int idx = (int)(wk.sorted[i] & 0xffffffffull), y = idx / w, x = idx - y * w;
            const int xc = x / cell, yc = y / cell;
            bool anyKept = false, anyUnknown = false;
            for (int yy = max(yc - 1, 0); yy <= min(yc + 1, gh - 1); yy++)
                for (int xx = max(xc - 1, 0); xx <= min(xc + 1, gw - 1); xx++) {
                    const int cc = yy * gw + xx;
                    for (int e = wk.cellStart[cc]; e < wk.cellStart[cc + 1]; e++) {
                        const int j = wk.cellItems[e];
                        if (j >= i) continue;  // lower rank (or itself): never looked at by the reference when i's turn comes
                        const int jdx = (int)(wk.sorted[j] & 0xffffffffull), jy = jdx / w, jx = jdx - jy * w;
                        const float dx = (float)x - (float)jx, dy = (float)y - (float)jy;
                        if ((double)(dx * dx + dy * dy) < md2) {
                            const int s = wk.state[j];
                            anyKept |= s == 1;
                            anyUnknown |= s == 0;
                        }
                    }
                }
            if (anyKept) wk.state[i] = 2;
            else if (!anyUnknown) wk.state[i] = 1;
            else atomicAdd(&shUnknown, 1);
        }
        __threadfence_block();
        __syncthreads();
        const int left = shUnknown;
        __syncthreads();
        if (left == 0) break;
    }
    // the kept candidates in rank order, the first maxCorners of them
    if (t == 0) shTotal = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        const int keep = i < n && wk.state[i] == 1;
        sh[t] = keep;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += add;
            __syncthreads();
        }
        const int base = shTotal, o = base + sh[t] - keep;
        const int limit = maxCorners > 0 ? min(maxCorners, cap) : cap;
        if (keep && o < limit) {
            const int idx = (int)(wk.sorted[i] & 0xffffffffull), y = idx / w, x = idx - y * w;
            xyOut[2 * o] = (float)x;
            xyOut[2 * o + 1] = (float)y;
        }
        __syncthreads();
        if (t == 1023) shTotal = base + sh[1023];
        __syncthreads();
        if (shTotal >= limit) break;
    }
    if (t == 0) {
        const int limit = maxCorners > 0 ? min(maxCorners, cap) : cap;
        *countOut = min(shTotal, limit);
        wk.count[1] = min(shTotal, limit);
    }
}

constexpr int kSubpixWavesPerGroup = 4;

// getRectSubPix element (i, j) of the (pw x pw) patch whose top-left sample is (ipx, ipy): samplers.cpp's two formulas with replicated borders
__device__ __forceinline__ float subpix_sample(const uint8_t *img, size_t stride, int sw, int sh, int pw, int ipx, int ipy, int rx, int rw, int i, int j,
                                               float a11, float a12, float a21, float a22, float b1, float b2)
{
    const int y0 = min(max(ipy + i, 0), sh - 1), y1 = min(max(ipy + i + 1, 0), sh - 1);
    const uint8_t *s = img + (size_t)y0 * stride, *s2 = img + (size_t)y1 * stride;
    if (j < rx) return __fadd_rn(__fmul_rn((float)s[0], b1), __fmul_rn((float)s2[0], b2));
    if (j < rw) {
        const int x = ipx + j;
        return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn((float)s[x], a11), __fmul_rn((float)s[x + 1], a12)), __fmul_rn((float)s2[x], a21)), __fmul_rn((float)s2[x + 1], a22));
    }
    return __fadd_rn(__fmul_rn((float)s[sw - 1], b1), __fmul_rn((float)s2[sw - 1], b2));
}

__global__ __launch_bounds__(64 * kSubpixWavesPerGroup) void k_corner_subpix(const uint8_t *__restrict__ img, size_t stride, int w, int h, float *__restrict__ xy,
                                                                             const int *__restrict__ countPtr, int nGiven, int win, int maxIters, double eps,
                                                                             const float *__restrict__ mask)
{
    extern __shared__ __align__(16) unsigned char subpix_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = countPtr ? *countPtr : nGiven, p = blockIdx.x * kSubpixWavesPerGroup + wave;
    if (p >= n) return;  // wave-uniform; no work-group barrier below
    const int ww = 2 * win + 1, pw = ww + 2, nWin = ww * ww, nPatch = pw * pw;
    const size_t waveBytes = (((size_t)nPatch * 4 + 15) & ~(size_t)15) + (size_t)5 * nWin * 8 + 64;
    float *patch = reinterpret_cast<float *>(subpix_smem + wave * waveBytes);
    double *terms = reinterpret_cast<double *>(subpix_smem + wave * waveBytes + (((size_t)nPatch * 4 + 15) & ~(size_t)15));
    double *sums = terms + (size_t)5 * nWin;
    const float ctx = xy[2 * p], cty = xy[2 * p + 1];
    float cix = ctx, ciy = cty;
    int iter = 0;
    double err = 0;
    do {
        // ---- the bilinear patch around (cix, ciy)
        const float ox = __fsub_rn(cix, (float)(pw - 1) * 0.5f), oy = __fsub_rn(ciy, (float)(pw - 1) * 0.5f);
        const int ipx = (int)floorf(ox), ipy = (int)floorf(oy);
        const float a = __fsub_rn(ox, (float)ipx), b = __fsub_rn(oy, (float)ipy);
        const float a11 = __fmul_rn(__fsub_rn(1.f, a), __fsub_rn(1.f, b)), a12 = __fmul_rn(a, __fsub_rn(1.f, b)), a21 = __fmul_rn(__fsub_rn(1.f, a), b),
                    a22 = __fmul_rn(a, b), b1 = __fsub_rn(1.f, b), b2 = b;
        int rx = 0, rw = pw;
        if (ipx < 0) rx = min(-ipx, pw);
        if (!(ipx < w - pw)) rw = max(w - ipx - 1, 0);
        for (int e = lane; e < nPatch; e += 64) {
            const int i = e / pw, j = e - i * pw;
            patch[e] = subpix_sample(img, stride, w, h, pw, ipx, ipy, rx, rw, i, j, a11, a12, a21, a22, b1, b2);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the window's products, all lanes
        for (int k = lane; k < nWin; k += 64) {
            const int i = k / ww, j = k - i * ww;
            const float *sp = patch + (i + 1) * pw + (j + 1);
            const double m = (double)mask[k];
            const double tgx = (double)__fsub_rn(sp[1], sp[-1]), tgy = (double)__fsub_rn(sp[pw], sp[-pw]);
            const double gxx = __dmul_rn(__dmul_rn(tgx, tgx), m), gxy = __dmul_rn(__dmul_rn(tgx, tgy), m), gyy = __dmul_rn(__dmul_rn(tgy, tgy), m);
            const double px = (double)(j - win), py = (double)(i - win);
            terms[k] = gxx;
            terms[nWin + k] = gxy;
            terms[2 * nWin + k] = gyy;
            terms[3 * nWin + k] = __dadd_rn(__dmul_rn(gxx, px), __dmul_rn(gxy, py));
            terms[4 * nWin + k] = __dadd_rn(__dmul_rn(gxy, px), __dmul_rn(gyy, py));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- five sums in the reference's order, one lane each
        if (lane < 5) {
            const double *tq = terms + (size_t)lane * nWin;
            double acc = 0;
            for (int k = 0; k < nWin; k++) acc = __dadd_rn(acc, tq[k]);
            sums[lane] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double A = sums[0], B = sums[1], C = sums[2], bb1 = sums[3], bb2 = sums[4];
        __builtin_amdgcn_wave_barrier();
        const double det = __dsub_rn(__dmul_rn(A, C), __dmul_rn(B, B));
        if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
        const double scale = __ddiv_rn(1.0, det);
        const float c2x = (float)__dsub_rn(__dadd_rn((double)cix, __dmul_rn(__dmul_rn(C, scale), bb1)), __dmul_rn(__dmul_rn(B, scale), bb2));
        const float c2y = (float)__dadd_rn(__dsub_rn((double)ciy, __dmul_rn(__dmul_rn(B, scale), bb1)), __dmul_rn(__dmul_rn(A, scale), bb2));
        const float ex = __fsub_rn(c2x, cix), ey = __fsub_rn(c2y, ciy);
        err = (double)__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
        cix = c2x;
        ciy = c2y;
        if (cix < 0 || cix >= (float)w || ciy < 0 || ciy >= (float)h) break;
    } while (++iter < maxIters && err > eps);
    if (fabsf(__fsub_rn(cix, ctx)) > (float)win || fabsf(__fsub_rn(ciy, cty)) > (float)win) { cix = ctx; ciy = cty; }
    if (lane == 0) {
        xy[2 * p] = cix;
        xy[2 * p + 1] = ciy;
    }
}

}  // namespace amos

using namespace amos;

struct amos_corners {
    int device = 0, w = 0, h = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    float *dEig = nullptr, *dMask = nullptr;
    int maskWin = 0;
    void *dWork = nullptr;
    CornerWork wk{};
};

extern "C" {

int amos_corners_create(int device, void *stream, int width, int height, amos_corners **out)
{
    if (!out || width < 8 || height < 8 || (long long)width * height > 0x7fffffffLL / 4) {
        set_error("amos_corners_create: invalid argument");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_corners *c = new amos_corners();
    c->device = device; c->w = width; c->h = height;
    if (stream) c->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return AMOS_ERR_DEVICE; }
        c->ownStream = true;
    }
    const size_t cells = (size_t)width * height + 2;  // cells of >= 1 pixel
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t oMax = carve(4), oCount = carve(16), oKeys = carve(8 * (size_t)kCornerCandCap), oSorted = carve(8 * (size_t)kCornerCandCap),
                 oStart = carve(4 * (cells + 1)), oFill = carve(4 * cells), oItems = carve(4 * (size_t)kCornerCandCap), oState = carve(kCornerCandCap);
    hipError_t e = hipMalloc(&c->dWork, off);
    if (e == hipSuccess) e = hipMalloc((void **)&c->dEig, sizeof(float) * (size_t)width * height);
    if (e == hipSuccess) e = hipMalloc((void **)&c->dMask, sizeof(float) * (2 * kSubpixMaxWin + 1) * (2 * kSubpixMaxWin + 1));
    if (e != hipSuccess) { set_error("amos_corners_create: %s", hipGetErrorString(e)); amos_corners_destroy(c); return AMOS_ERR_DEVICE; }
    unsigned char *b = (unsigned char *)c->dWork;
    c->wk.maxKey = (unsigned *)(b + oMax); c->wk.count = (int *)(b + oCount);
    c->wk.keys = (unsigned long long *)(b + oKeys); c->wk.sorted = (unsigned long long *)(b + oSorted);
    c->wk.cellStart = (int *)(b + oStart); c->wk.cellFill = (int *)(b + oFill); c->wk.cellItems = (int *)(b + oItems); c->wk.state = b + oState;
    *out = c;
    return AMOS_OK;
}

void amos_corners_destroy(amos_corners *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : {(void *)c->dWork, (void *)c->dEig, (void *)c->dMask}) if (p) (void)hipFree(p);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void *amos_corners_stream(amos_corners *c) { return c ? (void *)c->stream : nullptr; }

int amos_corners_good_features_device(amos_corners *c, const uint8_t *d_gray, size_t stride, int width, int height, int max_corners, double quality_level,
                                      double min_distance, double harris_k, float *d_xy, int xy_capacity, int *d_count, float *d_response)
{
    if (!c || !d_gray || !d_xy || !d_count || width < 8 || height < 8 || width > c->w || height > c->h || stride < (size_t)width || xy_capacity < 1 ||
        !(quality_level > 0) || !(min_distance >= 1) || min_distance > 1024) {
        set_error("amos_corners_good_features_device: invalid argument (frame within the handle's size, quality > 0, 1 <= min_distance <= 1024)");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(c->device));
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)(1.0 * scale), k2 = (float)(2.0 * scale);
    AMOS_HIP_CHECK(hipMemsetAsync(c->wk.maxKey, 0, 4, c->stream));
    AMOS_HIP_CHECK(hipMemsetAsync(c->wk.count, 0, 16, c->stream));
    hipLaunchKernelGGL(k_harris, dim3((width + kHarrisTW - 1) / kHarrisTW, (height + kHarrisTH - 1) / kHarrisTH), dim3(256), 0, c->stream, d_gray, stride, width, height,
                       k1, k2, (float)harris_k, c->dEig, c->wk.maxKey);
    if (d_response) AMOS_HIP_CHECK(hipMemcpyAsync(d_response, c->dEig, sizeof(float) * (size_t)width * height, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_corner_candidates, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, c->stream, c->dEig, width, height, quality_level, c->wk);
    hipLaunchKernelGGL(k_corner_rank, dim3(kCornerCandCap / 256), dim3(256), 0, c->stream, c->wk);
    const int cell = (int)lrint(min_distance), gw = (width + cell - 1) / cell, gh = (height + cell - 1) / cell;
    hipLaunchKernelGGL(k_corner_select, dim3(1), dim3(1024), 0, c->stream, c->wk, width, height, cell, gw, gh, min_distance * min_distance, max_corners, xy_capacity,
                       d_xy, d_count);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

// number of local-maximum candidates of the last amos_corners_good_features_device call (synchronises the handle's stream);
// more than 65 536 means the result was truncated: AMOS_ERR_CAPACITY
int amos_corners_candidate_count(amos_corners *c, int *count)
{
    if (!c || !count) { set_error("amos_corners_candidate_count: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(c->device));
    AMOS_HIP_CHECK(hipMemcpyAsync(count, c->wk.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (*count > kCornerCandCap) { set_error("amos_corners: %d corner candidates, capacity %d", *count, kCornerCandCap); return AMOS_ERR_CAPACITY; }
    return AMOS_OK;
}

int amos_corners_subpix_device(amos_corners *c, const uint8_t *d_gray, size_t stride, int width, int height, float *d_xy, const int *d_count, int n, int win,
                               int max_count, double epsilon)
{
    if (!c || !d_gray || !d_xy || win < 1 || win > kSubpixMaxWin || width < 2 * win + 5 || height < 2 * win + 5 || stride < (size_t)width || n < 0) {
        set_error("amos_corners_subpix_device: invalid argument (1 <= win <= %d, frame larger than the window)", kSubpixMaxWin);
        return AMOS_ERR_INVALID;
    }
    if (n == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(c->device));
    const int ww = 2 * win + 1, pw = ww + 2;
    if (c->maskWin != win) {  // the window weights, computed on the host like the reference computes them (float exp)
        float m[(2 * kSubpixMaxWin + 1) * (2 * kSubpixMaxWin + 1)];
        for (int i = 0; i < ww; i++) {
            const float y = (float)(i - win) / win;
            const float vy = expf(-y * y);
            for (int j = 0; j < ww; j++) {
                const float x = (float)(j - win) / win;
                m[i * ww + j] = (float)(vy * expf(-x * x));
            }
        }
        AMOS_HIP_CHECK(hipMemcpyAsync(c->dMask, m, sizeof(float) * ww * ww, hipMemcpyHostToDevice, c->stream));
        AMOS_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->maskWin = win;
    }
    double eps = epsilon > 0 ? epsilon : 0;
    eps *= eps;
    const int iters = max_count < 1 ? 1 : (max_count > 100 ? 100 : max_count);
    const size_t waveBytes = (((size_t)pw * pw * 4 + 15) & ~(size_t)15) + (size_t)5 * ww * ww * 8 + 64;
    const size_t lds = waveBytes * kSubpixWavesPerGroup;
    static DeviceOnce ldsAttr;  // per device (amos_common.h)
    AMOS_HIP_CHECK(set_max_dynamic_lds(ldsAttr, reinterpret_cast<const void *>(k_corner_subpix), 160 * 1024));
    hipLaunchKernelGGL(k_corner_subpix, dim3((n + kSubpixWavesPerGroup - 1) / kSubpixWavesPerGroup), dim3(64 * kSubpixWavesPerGroup), lds, c->stream, d_gray, stride,
                       width, height, d_xy, d_count, n, win, iters, eps, c->dMask);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
