// amos_mask_post.hip -- the memory-bound ends of the mask network's post-processing (a15: layers/functions/detection.py Detect,
// yolact_interface.py postprocess / prep_display as amos-slam_amd/mask/detect.py and post.py restate them) as single passes.
//
//   k_class_scores  conf [B][P][1 + C] (softmax output) -> scores [B][C][P]: the background column dropped, the classes transposed to
//                   the layout top-k wants, and every prior whose best class score is <= the threshold set to -1 (detect_batch's
//                   static-shape form of `conf_scores.max(0) > conf_thresh`).  PyTorch: a strided max (0.28 ms at 32 frames), a fill,
//                   a where and a transposing copy (0.35 ms) over 200 MB each; here one read and one write.  Pure selection: exact.
//   k_person_mask   cropped sigmoid masks [B][n][ph][pw] + per-detection flags -> uint8 [B][H][W]: bilinear upsample to the frame
//                   (align_corners = False, PyTorch's source index and weights), > 0.5, count of the flagged detections, x 255 modulo
//                   256 (the reference's `(m * 255).byte()`).  PyTorch: upsample to a [B][15][H][W] float tensor (590 MB at 32 frames),
//                   compare, and, widen, sum; here the 36 MB of masks are read and the 10 MB result written, unflagged detections skipped.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

namespace amos {

// One work-group transposes a tile of 64 priors x all classes through LDS: coalesced reads along the class axis, coalesced writes
// along the prior axis.  grid = (ceil(P / 64), B), block = 256.
constexpr int kScoreTile = 64;
__global__ __launch_bounds__(256) void k_class_scores(const float *__restrict__ conf, float *__restrict__ scores, int P, int C1, float thresh)
{
    extern __shared__ float tile[];  // [kScoreTile][C1 + 1]
    __shared__ float best[kScoreTile];
    const int b = blockIdx.y, p0 = blockIdx.x * kScoreTile, np = min(kScoreTile, P - p0), pitch = C1 + 1;
    const float *src = conf + ((size_t)b * P + p0) * C1;
    for (int e = threadIdx.x; e < np * C1; e += 256) {
        const int r = e / C1, c = e - r * C1;
        tile[r * pitch + c] = src[e];
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const float *row = tile + threadIdx.x * pitch;
        float m = row[1];
        bool nan = m != m;
        for (int c = 2; c < C1; c++) {
            const float v = row[c];
            nan = nan || v != v;
            m = v > m ? v : m;
        }
        best[threadIdx.x] = nan ? __builtin_nanf("") : m;  // torch.max propagates a NaN; NaN > thresh is false
    }
    __syncthreads();
    const int C = C1 - 1;
    for (int e = threadIdx.x; e < C * kScoreTile; e += 256) {
        const int c = e / kScoreTile, r = e - c * kScoreTile;
        if (r < np) scores[((size_t)b * C + c) * P + p0 + r] = best[r] > thresh ? tile[r * pitch + c + 1] : -1.f;
    }
}

// grid = (ceil(W / 64), ceil(H / 16), B), block = (64, 4).  The 64 x 16 output pixels of a work-group read a source window of at most
// kPmCols x kPmRows mask pixels (the masks are upsampled: 138 -> 640 is 0.216 source pixels per output pixel); the window of every flagged
// detection is staged in LDS by a few coalesced loads, and the four taps of a pixel come from there -- per-thread global gathers (4 per
// detection and pixel) made this kernel vector-memory-issue bound: 0.39 ms per 64 frames for 20 MB of output.  Windows larger than the
// staging area (downsampling, or upsampling by less than ~3.7 x) take the gathers as before.  Same arithmetic either way.
constexpr int kPmCols = 20, kPmRows = 6, kPmDet = 16, kPmTileH = 16;  // a work-group: 64 x 16 output pixels, four rows per thread
__global__ __launch_bounds__(256) void k_person_mask(const float *__restrict__ masks, const uint8_t *__restrict__ flags, uint8_t *__restrict__ out, int n,
                                                    int ph, int pw, int H, int W, float scaleH, float scaleW)
{
    __shared__ float win[kPmDet][kPmRows][kPmCols];
    __shared__ int sList[kPmDet], sCount;
    const int ox = blockIdx.x * 64 + threadIdx.x, b = blockIdx.z, t = threadIdx.y * 64 + threadIdx.x;
    auto src = [](float scale, int o) {
        const float f = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)o, 0.5f)), 0.5f);
        return f < 0.f ? 0.f : f;
    };
    // the work-group's source window (its first and last output pixel inside the image bound it: the map is monotone)
    const int ox0 = blockIdx.x * 64, ox1 = min(ox0 + 63, W - 1), oy0 = blockIdx.y * kPmTileH, oy1 = min(oy0 + kPmTileH - 1, H - 1);
    const int wx0 = (int)src(scaleW, ox0), wx1 = min((int)src(scaleW, ox1) + 1, pw - 1), wy0 = (int)src(scaleH, oy0), wy1 = min((int)src(scaleH, oy1) + 1, ph - 1);
    const int wc = wx1 - wx0 + 1, wr = wy1 - wy0 + 1;
    const bool staged = wc <= kPmCols && wr <= kPmRows && n <= kPmDet;  // uniform over the work-group
    if (staged) {
        if (t < 64) {  // the flagged detections, in order (wave 0: a ballot and its prefix counts)
            const bool f = t < n && flags[b * n + t] != 0;
            const unsigned long long m = __ballot(f);
            if (f) sList[__popcll(m & ((1ull << t) - 1ull))] = t;
            if (t == 0) sCount = (int)__popcll(m);
        }
        __syncthreads();
        const int cnt = sCount, per = wr * wc;
        for (int e = t; e < cnt * per; e += 256) {
            const int d = e / per, rem = e - d * per, r = rem / wc, c = rem - r * wc;
            win[d][r][c] = masks[((size_t)b * n + sList[d]) * ph * pw + (size_t)(wy0 + r) * pw + wx0 + c];
        }
        __syncthreads();
    }
    if (ox >= W) return;
    const float fx = src(scaleW, ox);
    const int x0 = (int)fx, x1 = x0 + (x0 < pw - 1 ? 1 : 0);
    const float lx = __fsub_rn(fx, (float)x0), hx = __fsub_rn(1.f, lx);
    for (int ry = threadIdx.y; ry < kPmTileH; ry += 4) {
        const int oy = oy0 + ry;
        if (oy >= H) break;
        const float fy = src(scaleH, oy);
        const int y0 = (int)fy, y1 = y0 + (y0 < ph - 1 ? 1 : 0);
        const float ly = __fsub_rn(fy, (float)y0), hy = __fsub_rn(1.f, ly);
        unsigned count = 0;
        if (staged) {
            const int cnt = sCount, ry0 = y0 - wy0, ry1 = y1 - wy0, cx0 = x0 - wx0, cx1 = x1 - wx0;
            for (int d = 0; d < cnt; d++) {
                const float p = win[d][ry0][cx0], r = win[d][ry0][cx1], s2 = win[d][ry1][cx0], u = win[d][ry1][cx1];
                const float v = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, p), __fmul_rn(lx, r))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, s2), __fmul_rn(lx, u))));
                count += v > 0.5f ? 1u : 0u;
            }
        } else {
            for (int i = 0; i < n; i++) {
                if (!flags[b * n + i]) continue;  // uniform over the work-group
                const float *m = masks + ((size_t)b * n + i) * ph * pw;
                const float p = m[y0 * pw + x0], r = m[y0 * pw + x1], s2 = m[y1 * pw + x0], u = m[y1 * pw + x1];
                const float v = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, p), __fmul_rn(lx, r))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, s2), __fmul_rn(lx, u))));
                count += v > 0.5f ? 1u : 0u;
            }
        }
        out[((size_t)b * H + oy) * W + ox] = (uint8_t)((count * 255u) & 0xffu);
    }
}

// The prediction head's outputs for one pyramid level (yolact.py PredictionModule.forward + Yolact.forward's cat / softmax, as mask/net.py
// restates them): raw = the merged 3 x 3 convolution's output WITHOUT bias, channels-last [B][cells][cpad] with channels
// [A x 4 box | A x (1 + C) class | A x D coefficient | padding]; this level's priors (cell-major, anchor-minor) start at prior p_off of
// the concatenated outputs loc [B][P][4] = raw + bias, conf [B][P][1 + C] = softmax over the classes of (raw + bias), coef [B][P][D] =
// tanh(raw + bias).  PyTorch: a bias pass, three strided reshape copies per level, three concatenations, a softmax and a tanh pass;
// here each level's tensor is read once and the three outputs are written once, as contiguous runs (the priors of consecutive cells are
// consecutive).  A work-group stages kHeadCells cells in LDS.  Softmax as PyTorch's: max, exp(x - max), sum, divide -- the sum is taken
// sequentially here and by a butterfly there, so conf agrees to float32 rounding (1 - 2 ulp), loc and coef bit for bit.
// grid = (ceil(cells / kHeadCells), B), block = 256.
constexpr int kHeadCells = 16;
__global__ __launch_bounds__(256) void k_head_outputs(const float *__restrict__ raw, const float *__restrict__ bias, float *__restrict__ loc, float *__restrict__ conf,
                                                     float *__restrict__ coef, int cells, int cpad, int A, int C1, int D, int P, int pOff,
                                                     float *__restrict__ scores, float thresh)
{
    extern __shared__ float sh[];  // [kHeadCells][cpad]
    const int b = blockIdx.y, c0 = blockIdx.x * kHeadCells, nc = min(kHeadCells, cells - c0);
    const int nLoc = A * 4, nConf = A * C1, nCoef = A * D;
    const float *src = raw + ((size_t)b * cells + c0) * cpad;
    for (int e = threadIdx.x * 4; e < nc * cpad; e += 256 * 4) {  // cpad % 4 == 0
        float4 v = *reinterpret_cast<const float4 *>(src + e);
        const float4 bb = *reinterpret_cast<const float4 *>(bias + e % cpad);
        v.x = __fadd_rn(v.x, bb.x); v.y = __fadd_rn(v.y, bb.y); v.z = __fadd_rn(v.z, bb.z); v.w = __fadd_rn(v.w, bb.w);
        *reinterpret_cast<float4 *>(sh + e) = v;
    }
    __syncthreads();
    // softmax over each prior's 1 + C class values, in place.  The maximum and the sum are taken by ONE thread per prior (sequential
    // order: the same bits whatever the work-group shape); the exponentials and the divisions by all threads.  The element loops walk
    // (cell, offset) pairs incrementally: an integer division per element cost more than the arithmetic it indexed.
    __shared__ float sMax[256], sSum[256];
#define AMOS_HEAD_FOREACH(n, body)                                                             \
    {                                                                                          \
        const int stepCell = 256 / (n), stepR = 256 % (n);                                     \
        int cell = (int)threadIdx.x / (n), r = (int)threadIdx.x - cell * (n);                  \
        for (int e = threadIdx.x; e < nc * (n); e += 256) {                                    \
            body;                                                                              \
            cell += stepCell;                                                                  \
            r += stepR;                                                                        \
            if (r >= (n)) { r -= (n); cell++; }                                                \
        }                                                                                      \
    }
    if ((int)threadIdx.x < nc * A) {
        const int cell = threadIdx.x / A, a = threadIdx.x - cell * A;
        const float *v = sh + cell * cpad + nLoc + a * C1;
        float m = v[0];
        for (int c = 1; c < C1; c++) m = fmaxf(m, v[c]);
        sMax[threadIdx.x] = m;
    }
    __syncthreads();
    const float invC1 = 1.f / (float)C1;
    AMOS_HEAD_FOREACH(nConf, {
        float *v = sh + cell * cpad + nLoc + r;
        int a = (int)((float)r * invC1);  // r / C1 for r < A * C1 <= a few hundred: the float quotient is off by at most one
        a -= a * C1 > r ? 1 : 0;
        a += (a + 1) * C1 <= r ? 1 : 0;
        *v = expf(__fsub_rn(*v, sMax[cell * A + a]));
    })
    AMOS_HEAD_FOREACH(nCoef, {
        float *v = sh + cell * cpad + nLoc + nConf + r;
        *v = tanhf(*v);
    })
    __syncthreads();
    if ((int)threadIdx.x < nc * A) {
        const int cell = threadIdx.x / A, a = threadIdx.x - cell * A;
        const float *v = sh + cell * cpad + nLoc + a * C1;
        float sum = 0.f;
        for (int c = 0; c < C1; c++) sum = __fadd_rn(sum, v[c]);
        sSum[threadIdx.x] = sum;
    }
    __syncthreads();
    const size_t prior0 = (size_t)b * P + pOff + (size_t)c0 * A;  // first prior of this work-group
    AMOS_HEAD_FOREACH(nLoc, loc[prior0 * 4 + e] = sh[cell * cpad + r])
    AMOS_HEAD_FOREACH(nCoef, coef[prior0 * D + e] = sh[cell * cpad + nLoc + nConf + r])
    if (!scores) {
        AMOS_HEAD_FOREACH(nConf, {
            int a = (int)((float)r * invC1);
            a -= a * C1 > r ? 1 : 0;
            a += (a + 1) * C1 <= r ? 1 : 0;
            conf[prior0 * C1 + e] = sh[cell * cpad + nLoc + r] / sSum[cell * A + a];
        })
        return;
    }
    // With `scores`: the class scores of Detect as k_class_scores makes them from the softmax output -- [B][C][P], the background column
    // dropped, -1 for every prior whose best class is not above the threshold -- written here, from the values this group holds, instead of
    // by a pass that reads the softmax tensor back (and `conf` itself only if the caller still wants it).  The same quotients, the same
    // comparisons: the same bits.
    AMOS_HEAD_FOREACH(nConf, {
        int a = (int)((float)r * invC1);
        a -= a * C1 > r ? 1 : 0;
        a += (a + 1) * C1 <= r ? 1 : 0;
        float *v = sh + cell * cpad + nLoc + r;
        *v = *v / sSum[cell * A + a];
    })
    __syncthreads();
    if ((int)threadIdx.x < nc * A) {
        const int cell = threadIdx.x / A, a = threadIdx.x - cell * A;
        const float *v = sh + cell * cpad + nLoc + a * C1;
        float m = v[1];
        bool nan = m != m;
        for (int c = 2; c < C1; c++) {
            const float q = v[c];
            nan = nan || q != q;
            m = q > m ? q : m;
        }
        sMax[threadIdx.x] = nan ? __builtin_nanf("") : m;  // torch.max propagates a NaN; NaN > thresh is false
    }
    __syncthreads();
    if (conf) AMOS_HEAD_FOREACH(nConf, conf[prior0 * C1 + e] = sh[cell * cpad + nLoc + r])
    {
        const int C = C1 - 1, np = nc * A;  // priors of this work-group: consecutive in every class row
        float *dst = scores + (size_t)b * C * P + pOff + (size_t)c0 * A;
        int c = (int)threadIdx.x / np, pl = (int)threadIdx.x - c * np;
        const int stepC = 256 / np, stepP = 256 % np;
        for (int e = threadIdx.x; e < C * np; e += 256) {
            const int cell = pl / A, a = pl - cell * A;
            dst[(size_t)c * P + pl] = sMax[pl] > thresh ? sh[cell * cpad + nLoc + a * C1 + c + 1] : -1.f;
            c += stepC;
            pl += stepP;
            if (pl >= np) { pl -= np; c++; }
        }
    }
#undef AMOS_HEAD_FOREACH
}


// Row-wise top-k, sorted descending (the `scores.topk(200)` per class of Fast NMS, layers/functions/detection.py:103-111 as
// detect_batch restates it): one work-group per row maps the values to order-preserving integer keys, finds the k-th largest key
// by a three-pass radix select (11 + 11 + 10 bits, LDS histograms), gathers everything above it plus the first ones equal to it
// (lowest index first), and sorts the k of them by (value descending, index ascending) with a bitonic network: five scans of the
// row whatever its values are (the first from HBM, the others from L2).  Values equal torch.topk's; among EQUAL values torch's
// order is unspecified, this one is by index.  PyTorch's multi-block top-k takes 0.75 ms for 2 560 rows of 19 248 at 32 frames.
// grid = rows, block = 256; k <= 256.
constexpr int kTopkBins = 2048;
__device__ __forceinline__ unsigned topk_key(float f)  // monotone: a < b  <=>  key(a) < key(b)  (-0 < +0; NaNs sort above +inf as in torch)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float topk_value(unsigned key) { return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key); }

// Sparse rows (round 5).  The class-score rows this kernel exists for are almost entirely ONE value -- `fill` = -1, the priors under the
// confidence threshold -- with a few hundred live scores above it.  With `fill` given (not NaN) the row is scanned ONCE: everything above
// `fill` is compacted into an LDS list of (key, ~index) and sorted there (bitonic, padded to a power of two); if the list holds fewer
// than k, the rest of the result is `fill` at its first indices, found by a scan from the row's start that stops as soon as it has
// them.  Exactly the generic result (value descending, index ascending).  The generic five-scan path below still serves: no `fill`, more
// than kTopkList live values, or values BELOW `fill` in a row whose list is short of k.
constexpr int kTopkList = 1024;   // (8 KB of LDS: a larger list costs the generic path, which lives on occupancy, more than it saves)
__global__ __launch_bounds__(256) void k_topk_rows(const float *__restrict__ x, float *__restrict__ values, long long *__restrict__ indices, int n, int k, float fill)
{
    __shared__ unsigned hist[kTopkBins];
    __shared__ unsigned part[256];
    __shared__ unsigned long long sel[256];
    __shared__ unsigned sPrefix, sMask, sNeed, sCountAbove, sWaveEq[4];
    __shared__ unsigned long long list[kTopkList];
    __shared__ unsigned sLive, sBelow;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *row = x + (size_t)blockIdx.x * n;  // generic path: read five times; after the first pass it comes from L2
    if (fill == fill) {
        const unsigned fk = topk_key(fill);
        if (t == 0) { sLive = 0; sBelow = 0; }
        __syncthreads();
        const bool vec = (n & 3) == 0 && ((uintptr_t)row & 15) == 0;
        const int nVec = vec ? n : 0;
        // (one LDS atomic per wave and element slot: the lanes' positions come from a ballot)
#define AMOS_TOPK_TAKE(have, v, i)                                                                                        \
        {                                                                                                                 \
            const unsigned key = topk_key(v);                                                                             \
            const bool up = (have) && key > fk;                                                                           \
            const unsigned long long m = __ballot(up);                                                                    \
            if (m) {                                                                                                      \
                unsigned base = 0;                                                                                        \
                if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&sLive, (unsigned)__popcll(m));                     \
                base = __shfl(base, (int)__builtin_ctzll(m), 64);                                                         \
                const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));                             \
                if (up && slot < (unsigned)kTopkList) list[slot] = ((unsigned long long)key << 32) | (unsigned)(0xffffffffu - (unsigned)(i)); \
            }                                                                                                             \
            if ((have) && key < fk) sBelow = 1u;                                                                          \
        }
        for (int i0 = 0; i0 < nVec; i0 += 1024) {
            if (*reinterpret_cast<volatile unsigned *>(&sLive) > (unsigned)kTopkList) break;  // a dense row: no point in finishing the scan (a hint: the barrier below decides)
            const int i = i0 + 4 * t;
            const bool have = i < n;
            const float4 v = have ? *reinterpret_cast<const float4 *>(row + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            AMOS_TOPK_TAKE(have, v.x, i)
            AMOS_TOPK_TAKE(have, v.y, i + 1)
            AMOS_TOPK_TAKE(have, v.z, i + 2)
            AMOS_TOPK_TAKE(have, v.w, i + 3)
        }
        for (int i0 = nVec; i0 < n; i0 += 256) {
            const int i = i0 + t;
            const bool have = i < n;
            const float v = have ? row[i] : 0.f;
            AMOS_TOPK_TAKE(have, v, i)
        }
#undef AMOS_TOPK_TAKE
        __syncthreads();
        const unsigned live = sLive;
        if (live <= (unsigned)kTopkList && (live >= (unsigned)k || !sBelow)) {  // uniform over the work-group
            unsigned n2 = 256;
            while (n2 < live) n2 <<= 1;
            for (unsigned i = live + t; i < n2; i += 256) list[i] = 0ull;  // padding sorts last
            __syncthreads();
            for (unsigned size = 2; size <= n2; size <<= 1)
                for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                    for (unsigned i = t; i < n2 / 2; i += 256) {  // pair (lo, lo + stride) of the bitonic network, descending overall
                        const unsigned lo = ((i / stride) * stride * 2) + (i % stride), hi = lo + stride;
                        const unsigned long long a = list[lo], b = list[hi];
                        const bool descending = (lo & size) == 0;
                        if (descending ? a < b : a > b) { list[lo] = b; list[hi] = a; }
                    }
                    __syncthreads();
                }
            const unsigned take = live < (unsigned)k ? live : (unsigned)k;
            if ((unsigned)t < take) {
                const unsigned long long v = list[t];
                values[(size_t)blockIdx.x * k + t] = topk_value((unsigned)(v >> 32));
                indices[(size_t)blockIdx.x * k + t] = (long long)(0xffffffffu - (unsigned)(v & 0xffffffffu));
            }
            // the rest: `fill` at its first indices (every element that is not live equals `fill` here)
            unsigned have = take;  // uniform
            for (int i0 = 0; i0 < n && have < (unsigned)k; i0 += 256) {
                const int i = i0 + t;
                const bool eq = i < n && topk_key(row[i]) == fk;
                const unsigned long long m = __ballot(eq);
                if (lane == 0) sWaveEq[wave] = (unsigned)__popcll(m);
                __syncthreads();
                unsigned base = have, total = 0;
                for (int w = 0; w < 4; w++) {
                    const unsigned c = sWaveEq[w];
                    if (w < wave) base += c;
                    total += c;
                }
                const unsigned rank = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (eq && rank < (unsigned)k) {
                    values[(size_t)blockIdx.x * k + rank] = fill;
                    indices[(size_t)blockIdx.x * k + rank] = (long long)i;
                }
                have += total;
                __syncthreads();
            }
            return;
        }
        __syncthreads();
    }
    if (t == 0) { sPrefix = 0; sMask = 0; sNeed = (unsigned)k; sCountAbove = 0; }
    __syncthreads();
    // ---- radix select of the k-th largest key
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; pass++) {
        const int shift = shifts[pass], bins = 1 << widths[pass];
        for (int b = t; b < kTopkBins; b += 256) hist[b] = 0;
        __syncthreads();
        const unsigned prefix = sPrefix, mask = sMask, need = sNeed;
        // a row is mostly ONE value (-1 for priors under the threshold): when every participating lane of the wave has the same
        // bin, one lane adds the count instead of 64 atomics on one address
#define AMOS_TOPK_COUNT(in, bin)                                                                                          \
        {                                                                                                                 \
            const unsigned long long m = __ballot(in);                                                                    \
            if (m) {                                                                                                      \
                const unsigned b0 = __builtin_amdgcn_readfirstlane((in) ? (bin) : __shfl((bin), __builtin_ctzll(m), 64)); \
                const bool same = __ballot((in) && (bin) != b0) == 0ull;                                                  \
                if (same) {                                                                                               \
                    if (lane == (int)__builtin_ctzll(m)) atomicAdd(&hist[b0], (unsigned)__popcll(m));                     \
                } else if (in) {                                                                                          \
                    atomicAdd(&hist[(bin)], 1u);                                                                          \
                }                                                                                                         \
            }                                                                                                             \
        }
        // the histogram does not care about the order of the elements: four per thread and trip through one 16-byte load when the row
        // allows it (four independent loads in flight instead of one: the scan is a chain of load -> ballot -> LDS atomic latencies)
        const bool vec = (n & 3) == 0 && ((uintptr_t)row & 15) == 0;
        const int nVec = vec ? n : 0;
        for (int i0 = 0; i0 < nVec; i0 += 1024) {
            const int i = i0 + 4 * t;
            const bool have = i < n;
            const float4 v = have ? *reinterpret_cast<const float4 *>(row + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned k0 = topk_key(v.x), k1 = topk_key(v.y), k2 = topk_key(v.z), k3 = topk_key(v.w);
            const bool in0 = have && (k0 & mask) == prefix, in1 = have && (k1 & mask) == prefix, in2 = have && (k2 & mask) == prefix, in3 = have && (k3 & mask) == prefix;
            const unsigned b0_ = (k0 >> shift) & (bins - 1), b1_ = (k1 >> shift) & (bins - 1), b2_ = (k2 >> shift) & (bins - 1), b3_ = (k3 >> shift) & (bins - 1);
            AMOS_TOPK_COUNT(in0, b0_)
            AMOS_TOPK_COUNT(in1, b1_)
            AMOS_TOPK_COUNT(in2, b2_)
            AMOS_TOPK_COUNT(in3, b3_)
        }
        for (int i0 = nVec; i0 < n; i0 += 256) {
            const int i = i0 + t;
            const unsigned key = i < n ? topk_key(row[i]) : 0u;
            const bool in = i < n && (key & mask) == prefix;
            const unsigned bin = (key >> shift) & (bins - 1);
            AMOS_TOPK_COUNT(in, bin)
        }
#undef AMOS_TOPK_COUNT
        __syncthreads();
        {  // partial sums of 8 bins per thread
            unsigned p = 0;
            for (int b = 0; b < 8; b++) p += hist[8 * t + b];
            part[t] = p;
        }
        __syncthreads();
        if (t < 64) {  // wave 0: the bin where the count from the top reaches `need`
            const unsigned q = part[4 * lane] + part[4 * lane + 1] + part[4 * lane + 2] + part[4 * lane + 3];  // bins 32 lane .. 32 lane + 31
            unsigned suffix = q;  // inclusive suffix sum over lanes >= lane
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_down(suffix, d, 64);
                if (lane + d < 64) suffix += o;
            }
            const unsigned above = suffix - q;  // elements in bins of higher lanes
            if (above < need && suffix >= need) {  // exactly one lane
                unsigned acc = above;
                int bin = 32 * lane + 31;
                for (; bin > 32 * lane; bin--) {
                    const unsigned c = hist[bin];
                    if (acc + c >= need) break;
                    acc += c;
                }
                sPrefix = prefix | ((unsigned)bin << shift);
                sMask = mask | ((unsigned)(bins - 1) << shift);
                sNeed = need - acc;  // how many of this bin's elements are still wanted
            }
        }
        __syncthreads();
    }
    const unsigned T = sPrefix, needEq = sNeed;  // the k-th largest key; how many elements equal to it belong to the top k
    const unsigned nAbove = (unsigned)k - needEq;
    // ---- gather: every key > T (any order: they are sorted below), and the first needEq keys == T in index order
    unsigned eqSeen = 0;  // uniform over the work-group
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + t;
        const unsigned key = i < n ? topk_key(row[i]) : 0u;
        const bool gt = i < n && key > T, eq = i < n && key == T;
        if (gt) {
            const unsigned slot = atomicAdd(&sCountAbove, 1u);
            sel[slot] = ((unsigned long long)key << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
        if (eqSeen < needEq) {  // uniform
            const unsigned long long m = __ballot(eq);
            if (lane == 0) sWaveEq[wave] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned base = eqSeen, total = 0;
            for (int w = 0; w < 4; w++) {
                const unsigned c = sWaveEq[w];
                if (w < wave) base += c;
                total += c;
            }
            const unsigned rank = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (eq && rank < needEq) sel[nAbove + rank] = ((unsigned long long)key << 32) | (unsigned)(0xffffffffu - (unsigned)i);
            eqSeen += total;
            __syncthreads();
        }
    }
    if (t >= k) sel[t] = 0ull;  // padding sorts last
    __syncthreads();
    // ---- bitonic sort of 256 composite keys, descending
    for (int size = 2; size <= 256; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = t ^ stride;
            const unsigned long long a = sel[t], b = sel[partner];
            __syncthreads();
            const bool descending = (t & size) == 0;  // direction of this thread's block at this stage
            const bool lower = (t & stride) == 0;
            const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
            sel[t] = (descending == lower) ? hi : lo;
            __syncthreads();
        }
    if (t < k) {
        const unsigned long long v = sel[t];
        values[(size_t)blockIdx.x * k + t] = topk_value((unsigned)(v >> 32));
        indices[(size_t)blockIdx.x * k + t] = (long long)(0xffffffffu - (unsigned)(v & 0xffffffffu));
    }
}

// The same selection with the ROW IN LDS (round 5): rows of up to kTopkLdsMax values -- the 19 248 class scores per row of this network are
// 77 KB -- are loaded once, as keys, by eight waves (eight 16-byte loads in flight per thread); the sparse attempt, the three histogram
// passes, the gather and the fill completion then read LDS, four keys per thread and trip, instead of making four more trips through L2 with
// a memory latency per step.  The gather has no barrier per 256 elements any more: values above the k-th key take their slots by an LDS
// atomic (any order: sorted below), the first `needEq` values equal to it are ranked by ONE prefix scan over per-thread counts of CONTIGUOUS
// chunks.  Same results as k_topk_rows.  The work-group owns its CU (89 KB of LDS), so this form serves the launches that leave CUs idle
// anyway: one frame = 80 rows, dense 76 -> 39 us, sparse (300 live scores) 28 -> 22 us; at 64 frames (5 120 rows) the five-scan kernel's
// eight rows per CU in flight win 404 against 884 us and keep the launch (launch_topk; tools/r5_topk_probe.py).
constexpr int kTopkLdsThreads = 512, kTopkLdsMax = 32768;
__global__ __launch_bounds__(kTopkLdsThreads) void k_topk_rows_lds(const float *__restrict__ x, float *__restrict__ values, long long *__restrict__ indices, int n, int k,
                                                                  float fill)
{
    extern __shared__ __align__(16) unsigned keys[];  // n
    __shared__ unsigned hist[kTopkBins];
    __shared__ unsigned part[kTopkLdsThreads];
    __shared__ unsigned long long sel[256];
    __shared__ unsigned long long list[kTopkList];
    __shared__ unsigned sPrefix, sMask, sNeed, sCountAbove, sLive, sBelow, sWave[kTopkLdsThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *row = x + (size_t)blockIdx.x * n;
    const bool sparse = fill == fill;
    const unsigned fk = topk_key(fill);
    if (t == 0) { sLive = 0; sBelow = 0; sPrefix = 0; sMask = 0; sNeed = (unsigned)k; sCountAbove = 0; }
    __syncthreads();
    // ---- the row -> LDS as keys; with `fill`, everything above it is also compacted into `list` on the way (one LDS atomic per wave and slot)
#define AMOS_TOPK_TAKE(have, key, i)                                                                                      \
    if (collect) {                                                                                                        \
        const bool up = (have) && (key) > fk;                                                                             \
        const unsigned long long m = __ballot(up);                                                                        \
        if (m) {                                                                                                          \
            unsigned base = 0;                                                                                            \
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&sLive, (unsigned)__popcll(m));                         \
            base = __shfl(base, (int)__builtin_ctzll(m), 64);                                                             \
            const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));                                 \
            if (up && slot < (unsigned)kTopkList) list[slot] = ((unsigned long long)(key) << 32) | (unsigned)(0xffffffffu - (unsigned)(i)); \
        }                                                                                                                 \
        if ((have) && (key) < fk) sBelow = 1u;                                                                            \
    }
    const bool vec = (n & 3) == 0 && ((uintptr_t)row & 15) == 0;
    const int nVec = vec ? n : 0;
    constexpr int kLoads = 8;  // 16-byte loads in flight per thread: the row arrives in two or three memory latencies, not in ten
    for (int base = 0; base < nVec; base += 4 * kTopkLdsThreads * kLoads) {
        float4 v[kLoads];
#pragma unroll
        for (int j = 0; j < kLoads; j++) {
            const int i = base + 4 * (j * kTopkLdsThreads + t);
            v[j] = i < n ? *reinterpret_cast<const float4 *>(row + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < kLoads; j++) {
            const int i = base + 4 * (j * kTopkLdsThreads + t);
            const bool have = i < n;
            const unsigned k0 = topk_key(v[j].x), k1 = topk_key(v[j].y), k2 = topk_key(v[j].z), k3 = topk_key(v[j].w);
            if (have) *reinterpret_cast<uint4 *>(&keys[i]) = make_uint4(k0, k1, k2, k3);
            // (a dense row overflows the list early: its waves stop collecting -- the same word for every lane of a wave, so uniform)
            const bool collect = sparse && *reinterpret_cast<volatile unsigned *>(&sLive) <= (unsigned)kTopkList;
            AMOS_TOPK_TAKE(have, k0, i)
            AMOS_TOPK_TAKE(have, k1, i + 1)
            AMOS_TOPK_TAKE(have, k2, i + 2)
            AMOS_TOPK_TAKE(have, k3, i + 3)
        }
    }
    for (int i0 = nVec; i0 < n; i0 += kTopkLdsThreads) {
        const int i = i0 + t;
        const bool have = i < n;
        const unsigned key = have ? topk_key(row[i]) : 0u;
        if (have) keys[i] = key;
        const bool collect = sparse && *reinterpret_cast<volatile unsigned *>(&sLive) <= (unsigned)kTopkList;
        AMOS_TOPK_TAKE(have, key, i)
    }
#undef AMOS_TOPK_TAKE
    __syncthreads();
    const unsigned live = sLive;
    if (sparse && live <= (unsigned)kTopkList && (live >= (unsigned)k || !sBelow)) {  // uniform: a sparse row -- sort the list, complete with `fill`
        unsigned n2 = 256;
        while (n2 < live) n2 <<= 1;
        for (unsigned i = live + t; i < n2; i += kTopkLdsThreads) list[i] = 0ull;  // padding sorts last
        __syncthreads();
        for (unsigned size = 2; size <= n2; size <<= 1)
            for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                for (unsigned i = t; i < n2 / 2; i += kTopkLdsThreads) {
                    const unsigned lo = ((i / stride) * stride * 2) + (i % stride), hi = lo + stride;
                    const unsigned long long a = list[lo], b = list[hi];
                    const bool descending = (lo & size) == 0;
                    if (descending ? a < b : a > b) { list[lo] = b; list[hi] = a; }
                }
                __syncthreads();
            }
        const unsigned take = live < (unsigned)k ? live : (unsigned)k;
        if ((unsigned)t < take) {
            const unsigned long long v = list[t];
            values[(size_t)blockIdx.x * k + t] = topk_value((unsigned)(v >> 32));
            indices[(size_t)blockIdx.x * k + t] = (long long)(0xffffffffu - (unsigned)(v & 0xffffffffu));
        }
        unsigned have = take;  // uniform.  The rest: `fill` at its first indices (every element that is not live equals `fill` here)
        for (int i0 = 0; i0 < n && have < (unsigned)k; i0 += kTopkLdsThreads) {
            const int i = i0 + t;
            const bool eq = i < n && keys[i] == fk;
            const unsigned long long m = __ballot(eq);
            if (lane == 0) sWave[wave] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned base = have, total = 0;
            for (int w = 0; w < kTopkLdsThreads / 64; w++) {
                const unsigned c = sWave[w];
                if (w < wave) base += c;
                total += c;
            }
            const unsigned rank = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (eq && rank < (unsigned)k) {
                values[(size_t)blockIdx.x * k + rank] = fill;
                indices[(size_t)blockIdx.x * k + rank] = (long long)i;
            }
            have += total;
            __syncthreads();
        }
        return;
    }
    // ---- radix select of the k-th largest key, from LDS
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; pass++) {
        const int shift = shifts[pass], bins = 1 << widths[pass];
        for (int b = t; b < kTopkBins; b += kTopkLdsThreads) hist[b] = 0;
        __syncthreads();
        const unsigned prefix = sPrefix, mask = sMask, need = sNeed;
#define AMOS_TOPK_COUNT(in, bin)                                                                                          \
        {   /* (a row that is mostly one value: one lane adds the wave's count instead of 64 atomics on one address) */  \
            const unsigned long long m = __ballot(in);                                                                    \
            if (m) {                                                                                                      \
                const unsigned b0 = __builtin_amdgcn_readlane((bin), (int)__builtin_ctzll(m)); /* the first counting lane's bin */ \
                const bool same = __ballot((in) && (bin) != b0) == 0ull;                                                  \
                if (same) {                                                                                               \
                    if (lane == (int)__builtin_ctzll(m)) atomicAdd(&hist[b0], (unsigned)__popcll(m));                     \
                } else if (in) {                                                                                          \
                    atomicAdd(&hist[(bin)], 1u);                                                                          \
                }                                                                                                         \
            }                                                                                                             \
        }
        for (int i0 = 0; i0 < n; i0 += 4 * kTopkLdsThreads) {  // four keys per thread and trip: one 16-byte LDS read, four independent chains
            const int i = i0 + 4 * t;
            const uint4 kk = i < n ? *reinterpret_cast<const uint4 *>(&keys[i]) : make_uint4(0u, 0u, 0u, 0u);  // (the array is padded to four)
            const bool in0 = i < n && (kk.x & mask) == prefix, in1 = i + 1 < n && (kk.y & mask) == prefix;
            const bool in2 = i + 2 < n && (kk.z & mask) == prefix, in3 = i + 3 < n && (kk.w & mask) == prefix;
            const unsigned b0_ = (kk.x >> shift) & (bins - 1), b1_ = (kk.y >> shift) & (bins - 1), b2_ = (kk.z >> shift) & (bins - 1), b3_ = (kk.w >> shift) & (bins - 1);
            AMOS_TOPK_COUNT(in0, b0_)
            AMOS_TOPK_COUNT(in1, b1_)
            AMOS_TOPK_COUNT(in2, b2_)
            AMOS_TOPK_COUNT(in3, b3_)
        }
#undef AMOS_TOPK_COUNT
        __syncthreads();
        {  // partial sums of 4 bins per thread
            unsigned p = 0;
            for (int b = 0; b < kTopkBins / kTopkLdsThreads; b++) p += hist[(kTopkBins / kTopkLdsThreads) * t + b];
            part[t] = p;
        }
        __syncthreads();
        if (t < 64) {  // wave 0: the bin where the count from the top reaches `need`; lane = bins 32 lane .. 32 lane + 31 = 8 partial sums
            unsigned q = 0;
            for (int j = 0; j < kTopkLdsThreads / 64; j++) q += part[(kTopkLdsThreads / 64) * lane + j];
            unsigned suffix = q;  // inclusive suffix sum over lanes >= lane
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_down(suffix, d, 64);
                if (lane + d < 64) suffix += o;
            }
            const unsigned above = suffix - q;
            if (above < need && suffix >= need) {  // exactly one lane
                unsigned acc = above;
                int bin = 32 * lane + 31;
                for (; bin > 32 * lane; bin--) {
                    const unsigned c = hist[bin];
                    if (acc + c >= need) break;
                    acc += c;
                }
                sPrefix = prefix | ((unsigned)bin << shift);
                sMask = mask | ((unsigned)(bins - 1) << shift);
                sNeed = need - acc;
            }
        }
        __syncthreads();
    }
    const unsigned T = sPrefix, needEq = sNeed, nAbove = (unsigned)k - needEq;
    // ---- gather: every key > T (any order), then the first needEq keys == T in index order
    for (int i0 = 0; i0 < n; i0 += 4 * kTopkLdsThreads) {
        const int i = i0 + 4 * t;
        if (i < n) {
            const uint4 kk = *reinterpret_cast<const uint4 *>(&keys[i]);
            const unsigned key[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (i + e < n && key[e] > T) sel[atomicAdd(&sCountAbove, 1u)] = ((unsigned long long)key[e] << 32) | (unsigned)(0xffffffffu - (unsigned)(i + e));
        }
    }
    {
        const int chunk = (n + kTopkLdsThreads - 1) / kTopkLdsThreads, lo = min(t * chunk, n), hi = min(lo + chunk, n);
        unsigned cnt = 0;
        for (int i = lo; i < hi; i++) cnt += keys[i] == T ? 1u : 0u;
        unsigned incl = cnt;  // inclusive scan over the wave, then the waves' totals
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        unsigned rank = incl - cnt;
        for (int w = 0; w < wave; w++) rank += sWave[w];
        for (int i = lo; i < hi && rank < needEq; i++)
            if (keys[i] == T) {
                sel[nAbove + rank] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)i);
                rank++;
            }
    }
    if (t >= k && t < 256) sel[t] = 0ull;  // padding sorts last
    __syncthreads();
    // ---- bitonic sort of 256 composite keys, descending (the first 256 threads; everybody keeps the barriers)
    for (int size = 2; size <= 256; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            unsigned long long a = 0, b = 0;
            if (t < 256) { a = sel[t]; b = sel[t ^ stride]; }
            __syncthreads();
            if (t < 256) {
                const bool descending = (t & size) == 0, lower = (t & stride) == 0;
                const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
                sel[t] = (descending == lower) ? hi : lo;
            }
            __syncthreads();
        }
    if (t < k) {
        const unsigned long long v = sel[t];
        values[(size_t)blockIdx.x * k + t] = topk_value((unsigned)(v >> 32));
        indices[(size_t)blockIdx.x * k + t] = (long long)(0xffffffffu - (unsigned)(v & 0xffffffffu));
    }
}


// ---- the rest of Detect + postprocess + prep_display for the static-shape batch path (mask/detect.py detect_batch, mask/post.py
// person_mask_batch), fused: PyTorch spends ~75 small launches per pass on these steps (box decoding, two gathers, where / topk / gathers of
// the best 100 and then the best 15, `_sanitize` and the crop mask op by op, an einsum and a sigmoid) -- at one frame per pass that is a
// fifth of the whole mask pass (0.6 of 3.0 ms, tools/r5_one_frame_trace.sh).  Here: four kernels.

// SSD box decoding with variances (0.1, 0.2), layers/box_utils.py decode as detect_batch evaluates it, operation by operation in float32:
//   centre = prior_xy + (loc_xy * 0.1) * prior_wh;  size = prior_wh * exp(loc_wh * 0.2);  x1y1 = centre - size / 2;  x2y2 = x1y1 + size.
// grid = ceil(B * P / 256), block = 256.
__global__ __launch_bounds__(256) void k_decode_boxes(const float4 *__restrict__ loc, const float4 *__restrict__ priors, float4 *__restrict__ boxes, int P, int total)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const float4 l = loc[e], p = priors[e % P];
    const float cx = __fadd_rn(p.x, __fmul_rn(__fmul_rn(l.x, 0.1f), p.z)), cy = __fadd_rn(p.y, __fmul_rn(__fmul_rn(l.y, 0.1f), p.w));
    const float sw = __fmul_rn(p.z, expf(__fmul_rn(l.z, 0.2f))), sh = __fmul_rn(p.w, expf(__fmul_rn(l.w, 0.2f)));
    const float x1 = __fsub_rn(cx, __fmul_rn(sw, 0.5f)), y1 = __fsub_rn(cy, __fmul_rn(sh, 0.5f));
    boxes[e] = make_float4(x1, y1, __fadd_rn(x1, sw), __fadd_rn(y1, sh));
}

// Fast NMS on one class list (k score-sorted candidates, given as prior indices): k_nms_column_max's suppression term on boxes gathered
// here, then detect_batch's `alive = (term <= thresh) & (score > 0)`; out = alive ? score : -1.  grid = B * C lists, block = 256 (k <= 256).
__global__ __launch_bounds__(256) void k_nms_alive(const float4 *__restrict__ boxes, const long long *__restrict__ idx, const float *__restrict__ scores,
                                                  float *__restrict__ out, int k, int P, int C, float thresh)
{
    __shared__ float4 sb[256];
    __shared__ float sarea[256];
    const int j = threadIdx.x, list = blockIdx.x, b = list / C;
    float4 me = {0.f, 0.f, 0.f, 0.f};
    float myArea = 0.f;
    if (j < k) {
        me = boxes[(size_t)b * P + (size_t)idx[(size_t)list * k + j]];
        myArea = __fmul_rn(__fsub_rn(me.z, me.x), __fsub_rn(me.w, me.y));
        sb[j] = me;
        sarea[j] = myArea;
    }
    __syncthreads();
    if (j >= k) return;
    float best = 0.f;
    bool nan = false;
    for (int i = 0; i < j; i++) {
        const float4 o = sb[i];
        float w = __fsub_rn(fminf(o.z, me.z), fmaxf(o.x, me.x)), h = __fsub_rn(fminf(o.w, me.w), fmaxf(o.y, me.y));
        w = w < 0.f ? 0.f : w;
        h = h < 0.f ? 0.f : h;
        const float inter = __fmul_rn(w, h);
        const float uni = __fsub_rn(__fadd_rn(sarea[i], myArea), inter);
        const float iou = (float)((double)inter / (double)uni);
        nan = nan || iou != iou;
        best = iou > best ? iou : best;
    }
    const float sc = scores[(size_t)list * k + j];
    out[(size_t)list * k + j] = (!nan && best <= thresh && sc > 0.f) ? sc : -1.f;  // (a NaN term compares false in torch too)
}

// The detections the reference displays: the nDisplay best of a frame's C * k surviving scores (detect_batch's best 100 followed by
// person_mask_batch's best 15 of those: the same set; order: score descending, equal scores lowest flat index first, as k_topk_rows
// orders them) that exceed the score threshold.  For each: its class (flat index / k), its box as the crop rectangle of box_utils.crop
// on the prototype grid (`_sanitize`: scale, order, pad by 1, clamp), its 32 mask coefficients, and the flag "valid and a person".
// found[b] = some score exceeds the threshold.  grid = B, block = 256.
__global__ __launch_bounds__(256) void k_select_display(const float *__restrict__ alive, const long long *__restrict__ idx, const float4 *__restrict__ boxes,
                                                       const float *__restrict__ coef, float *__restrict__ selCoef, float4 *__restrict__ selRect,
                                                       uint8_t *__restrict__ flags, uint8_t *__restrict__ found, int n, int k, int P, int D, int nDisplay,
                                                       float scoreThresh, int personClass, int pw, int ph)
{
    // Only scores above the threshold can be displayed: they are first compacted into a list of (score key, ~flat index) in LDS (a frame
    // has a few hundred at most: Fast NMS has already thinned 80 x 200 candidates), and the nDisplay rounds of "largest key below the last
    // one" run over that list.  A frame with more than kSelectList of them (never seen; possible in principle) selects over the whole row.
    constexpr int kSelectList = 4096;
    __shared__ unsigned long long list[kSelectList];
    __shared__ unsigned long long sWave[4], sSel[32];
    __shared__ unsigned sCount;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float *row = alive + (size_t)b * n;
    if (t == 0) sCount = 0;
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        const float v = row[i];
        if (v > scoreThresh) {
            const unsigned slot = atomicAdd(&sCount, 1u);
            if (slot < (unsigned)kSelectList) list[slot] = ((unsigned long long)topk_key(v) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
    }
    __syncthreads();
    const unsigned count = sCount;
    const bool listed = count <= (unsigned)kSelectList;
    unsigned long long prev = ~0ull;
    for (int r = 0; r < nDisplay; r++) {
        unsigned long long best = 0ull;
        if (listed) {
            for (unsigned i = t; i < count; i += 256) {
                const unsigned long long key = list[i];
                best = (key < prev && key > best) ? key : best;
            }
        } else {
            for (int i = t; i < n; i += 256) {
                const float v = row[i];
                const unsigned long long key = v > scoreThresh ? (((unsigned long long)topk_key(v) << 32) | (unsigned)(0xffffffffu - (unsigned)i)) : 0ull;
                best = (key < prev && key > best) ? key : best;
            }
        }
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) sWave[wave] = best;
        __syncthreads();
        unsigned long long m = sWave[0];
        for (int w = 1; w < 4; w++) m = sWave[w] > m ? sWave[w] : m;
        if (t == 0) sSel[r] = m;
        prev = m ? m : 1ull;  // (nothing left: every later round finds nothing either)
        __syncthreads();
    }
    if (t < nDisplay) {
        const unsigned long long key = sSel[t];
        const float score = topk_value((unsigned)(key >> 32));
        const int flat = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
        const bool valid = key != 0ull && score > scoreThresh;
        const int cls = valid ? flat / k : 0;
        const long long prior = valid ? idx[(size_t)b * n + flat] : 0;
        const float4 bx = boxes[(size_t)b * P + (size_t)prior];
        // _sanitize(x1, x2, pw, padding = 1): a = x1 * size, b = x2 * size; clamp(min(a, b) - 1, min = 0), clamp(max(a, b) + 1, max = size)
        const float ax = __fmul_rn(bx.x, (float)pw), bxx = __fmul_rn(bx.z, (float)pw), ay = __fmul_rn(bx.y, (float)ph), by = __fmul_rn(bx.w, (float)ph);
        float4 rc;
        rc.x = fmaxf(__fsub_rn(fminf(ax, bxx), 1.f), 0.f);
        rc.y = fminf(__fadd_rn(fmaxf(ax, bxx), 1.f), (float)pw);
        rc.z = fmaxf(__fsub_rn(fminf(ay, by), 1.f), 0.f);
        rc.w = fminf(__fadd_rn(fmaxf(ay, by), 1.f), (float)ph);
        selRect[(size_t)b * nDisplay + t] = rc;  // (x1, x2, y1, y2)
        flags[(size_t)b * nDisplay + t] = (valid && cls == personClass) ? 1 : 0;
        if (t == 0) found[b] = valid ? 1 : 0;
        sSel[t] = (unsigned long long)prior;  // for the coefficient copy below
    }
    __syncthreads();
    for (int e = t; e < nDisplay * D; e += 256) {
        const int r = e / D, c = e - r * D;
        selCoef[((size_t)b * nDisplay + r) * D + c] = coef[((size_t)b * P + (size_t)sSel[r]) * D + c];
    }
}

// masks[b][n][y][x] = inside(rect n) ? sigmoid(proto[b][y][x][:] . coef[b][n][:]) : 0 for the flagged detections (k_person_mask reads no
// other): postprocess's `sigmoid(proto @ coef^T)` + crop.  Sum in channel order with fused multiply-adds (the BLAS kernel PyTorch calls
// for the einsum fuses too, in an order of its own: the two agree to float32 rounding of a 32-term sum); sigmoid as PyTorch's
// 1 / (1 + exp(-x)) in float32.  grid = (ceil(ph * pw / 256), nDisplay, B), block = 256.
__global__ __launch_bounds__(256) void k_assemble_masks(const float *__restrict__ proto, const float *__restrict__ selCoef, const float4 *__restrict__ selRect,
                                                       const uint8_t *__restrict__ flags, float *__restrict__ masks, int ph, int pw, int D, int nDisplay)
{
    const int n = blockIdx.y, b = blockIdx.z, pix = blockIdx.x * 256 + threadIdx.x;
    if (!flags[(size_t)b * nDisplay + n] || pix >= ph * pw) return;  // (the flag is uniform over the work-group)
    const int y = pix / pw, x = pix - y * pw;
    const float4 rc = selRect[(size_t)b * nDisplay + n];
    float v = 0.f;
    if ((float)x >= rc.x && (float)x < rc.y && (float)y >= rc.z && (float)y < rc.w) {
        const float4 *pp = reinterpret_cast<const float4 *>(proto + ((size_t)b * ph * pw + pix) * D);
        const float4 *cc = reinterpret_cast<const float4 *>(selCoef + ((size_t)b * nDisplay + n) * D);
        float acc = 0.f;
        for (int q = 0; q < D / 4; q++) {
            const float4 a = pp[q], c = cc[q];
            acc = __fmaf_rn(a.x, c.x, acc); acc = __fmaf_rn(a.y, c.y, acc); acc = __fmaf_rn(a.z, c.z, acc); acc = __fmaf_rn(a.w, c.w, acc);
        }
        v = 1.f / (1.f + expf(-acc));
    }
    masks[((size_t)b * nDisplay + n) * ph * pw + pix] = v;
}

}  // namespace amos

using namespace amos;

extern "C" {

int amos_mask_class_scores_device(void *stream, const float *d_conf, float *d_scores, int batch, int n_priors, int n_classes_with_background, float threshold)
{
    if (!d_conf || !d_scores || batch < 0 || n_priors < 1 || n_classes_with_background < 2 || n_classes_with_background > 200 || batch > 65535) {  // 64 x (classes + 1) floats of LDS
        set_error("amos_mask_class_scores_device: invalid argument (2 <= classes incl. background <= 200)");
        return AMOS_ERR_INVALID;
    }
    if (batch == 0) return AMOS_OK;
    const size_t lds = (size_t)kScoreTile * (n_classes_with_background + 1) * sizeof(float);
    hipLaunchKernelGGL(k_class_scores, dim3((n_priors + kScoreTile - 1) / kScoreTile, batch), dim3(256), lds, (hipStream_t)stream, d_conf, d_scores, n_priors,
                       n_classes_with_background, threshold);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_person_mask_device(void *stream, const float *d_masks, const uint8_t *d_flags, uint8_t *d_out, int batch, int n_det, int mask_h, int mask_w,
                                 int out_h, int out_w)
{
    if (!d_masks || !d_flags || !d_out || batch < 0 || n_det < 0 || mask_h < 1 || mask_w < 1 || out_h < 1 || out_w < 1 || batch > 65535 || out_h > 16 * 65535) {
        set_error("amos_mask_person_mask_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (batch == 0) return AMOS_OK;
    // PyTorch's area_pixel_compute_scale for a given output size: input / output in float32
    const float sh = (float)mask_h / (float)out_h, sw = (float)mask_w / (float)out_w;
    hipLaunchKernelGGL(k_person_mask, dim3((out_w + 63) / 64, (out_h + kPmTileH - 1) / kPmTileH, batch), dim3(64, 4), 0, (hipStream_t)stream, d_masks, d_flags, d_out, n_det, mask_h,
                       mask_w, out_h, out_w, sh, sw);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_head_outputs_device(void *stream, const float *d_raw, const float *d_bias, float *d_loc, float *d_conf, float *d_coef, int batch, int cells,
                                  int channels_padded, int anchors, int n_classes_with_background, int mask_dim, int n_priors_total, int prior_offset)
{
    if (!d_conf) { set_error("amos_mask_head_outputs_device: invalid argument"); return AMOS_ERR_INVALID; }
    return amos_mask_head_outputs_scores_device(stream, d_raw, d_bias, d_loc, d_conf, d_coef, nullptr, 0.f, batch, cells, channels_padded, anchors,
                                                n_classes_with_background, mask_dim, n_priors_total, prior_offset);
}

int amos_mask_head_outputs_scores_device(void *stream, const float *d_raw, const float *d_bias, float *d_loc, float *d_conf, float *d_coef, float *d_scores,
                                         float threshold, int batch, int cells, int channels_padded, int anchors, int n_classes_with_background, int mask_dim,
                                         int n_priors_total, int prior_offset)
{
    const long long used = (long long)anchors * (4 + n_classes_with_background + mask_dim);
    if (!d_raw || !d_bias || !d_loc || (!d_conf && !d_scores) || (d_scores && n_classes_with_background < 2) || !d_coef || batch < 0 || cells < 1 || anchors < 1 || n_classes_with_background < 1 || mask_dim < 1 ||
        channels_padded % 4 != 0 || used > channels_padded || channels_padded > 1000 || prior_offset < 0 ||  // 16 x channels floats of LDS
       
        (long long)prior_offset + (long long)cells * anchors > n_priors_total || batch > 65535 || kHeadCells * anchors > 256 ||
        ((uintptr_t)d_raw | (uintptr_t)d_bias) % 16 != 0) {
        set_error("amos_mask_head_outputs_scores_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (batch == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_head_outputs, dim3((cells + kHeadCells - 1) / kHeadCells, batch), dim3(256), (size_t)kHeadCells * channels_padded * sizeof(float),
                       (hipStream_t)stream, d_raw, d_bias, d_loc, d_conf, d_coef, cells, channels_padded, anchors, n_classes_with_background, mask_dim,
                       n_priors_total, prior_offset, d_scores, threshold);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

// rows that fit LDS as keys take k_topk_rows_lds (AMOS_TOPK_LDS=0 in the environment: always the five-scan kernel -- A/B runs)
static DeviceOnce g_topk_lds_once;
static int launch_topk(hipStream_t stream, const float *d_x, float *d_values, long long *d_indices, int rows, int n, int k, float fill)
{
    static const bool ldsAllowed = [] { const char *e = getenv("AMOS_TOPK_LDS"); return !(e && e[0] == '0'); }();
    // one row per CU at most: the LDS kernel's work-group owns a CU (89 KB of LDS), so it serves the launches that leave CUs idle anyway -- a
    // frame or three (80 rows each); the five-scan kernel keeps eight rows per CU in flight and wins every larger launch
    // (tools/r5_topk_probe.py: 80 dense rows 76 -> 43 us, 5 120 rows 404 against 884)
    if (ldsAllowed && n <= kTopkLdsMax && rows <= 256) {
        const int ldsBytes = ((n + 3) & ~3) * (int)sizeof(unsigned);
        AMOS_HIP_CHECK(set_max_dynamic_lds(g_topk_lds_once, reinterpret_cast<const void *>(k_topk_rows_lds), kTopkLdsMax * (int)sizeof(unsigned), stream));
        hipLaunchKernelGGL(k_topk_rows_lds, dim3(rows), dim3(kTopkLdsThreads), ldsBytes, stream, d_x, d_values, d_indices, n, k, fill);
    } else {
        hipLaunchKernelGGL(k_topk_rows, dim3(rows), dim3(256), 0, stream, d_x, d_values, d_indices, n, k, fill);
    }
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_topk_rows_device(void *stream, const float *d_x, float *d_values, long long *d_indices, int rows, int n, int k)
{
    if (!d_x || !d_values || !d_indices || rows < 0 || n < 1 || k < 1 || k > 256 || k > n) {
        set_error("amos_mask_topk_rows_device: invalid argument (1 <= k <= min(256, n))");
        return AMOS_ERR_INVALID;
    }
    if (rows == 0) return AMOS_OK;
    return launch_topk((hipStream_t)stream, d_x, d_values, d_indices, rows, n, k, __builtin_nanf(""));
}

int amos_mask_topk_rows_sparse_device(void *stream, const float *d_x, float *d_values, long long *d_indices, int rows, int n, int k, float fill)
{
    if (!d_x || !d_values || !d_indices || rows < 0 || n < 1 || k < 1 || k > 256 || k > n) {
        set_error("amos_mask_topk_rows_sparse_device: invalid argument (1 <= k <= min(256, n))");
        return AMOS_ERR_INVALID;
    }
    if (rows == 0) return AMOS_OK;
    return launch_topk((hipStream_t)stream, d_x, d_values, d_indices, rows, n, k, fill);
}


// ---- the whole post-processing chain as one call (seven launches on `stream`)
namespace {
struct PostLayout {
    size_t boxes, cls, topv, topi, alive, selCoef, selRect, flags, masks, total;
};
inline size_t post_align(size_t v) { return (v + 255) & ~(size_t)255; }
PostLayout post_layout(int B, int P, int C1, int D, int ph, int pw)
{
    const size_t C = (size_t)C1 - 1, k = AMOS_MASK_NMS_TOP_K, nd = AMOS_MASK_TOP_K_DISPLAY;
    PostLayout l;
    size_t o = 0;
    l.boxes = o; o += post_align((size_t)B * P * 16);
    l.cls = o; o += post_align((size_t)B * C * P * 4);
    l.topv = o; o += post_align((size_t)B * C * k * 4);
    l.topi = o; o += post_align((size_t)B * C * k * 8);
    l.alive = o; o += post_align((size_t)B * C * k * 4);
    l.selCoef = o; o += post_align((size_t)B * nd * D * 4);
    l.selRect = o; o += post_align((size_t)B * nd * 16);
    l.flags = o; o += post_align((size_t)B * nd);
    l.masks = o; o += post_align((size_t)B * nd * ph * pw * 4);
    l.total = o;
    return l;
}
}  // namespace

size_t amos_mask_post_workspace_bytes(int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w)
{
    if (batch < 1 || n_priors < AMOS_MASK_NMS_TOP_K || n_classes_with_background < 2 || mask_dim < 4 || proto_h < 1 || proto_w < 1) return 0;
    return post_layout(batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w).total;
}

static int person_masks(void *stream, const float *d_loc, const float *d_conf, const float *d_scores, const float *d_coef, const float *d_priors, const float *d_proto,
                        int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w, int out_h, int out_w, void *d_workspace,
                        size_t workspace_bytes, uint8_t *d_masks, uint8_t *d_found);

int amos_mask_person_masks_device(void *stream, const float *d_loc, const float *d_conf, const float *d_coef, const float *d_priors, const float *d_proto,
                                  int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w, int out_h, int out_w,
                                  void *d_workspace, size_t workspace_bytes, uint8_t *d_masks, uint8_t *d_found)
{
    if (!d_conf) { set_error("amos_mask_person_masks_device: invalid argument"); return AMOS_ERR_INVALID; }
    return person_masks(stream, d_loc, d_conf, nullptr, d_coef, d_priors, d_proto, batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w, out_h, out_w,
                        d_workspace, workspace_bytes, d_masks, d_found);
}

int amos_mask_person_masks_scores_device(void *stream, const float *d_loc, const float *d_scores, const float *d_coef, const float *d_priors, const float *d_proto,
                                         int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w, int out_h, int out_w,
                                         void *d_workspace, size_t workspace_bytes, uint8_t *d_masks, uint8_t *d_found)
{
    if (!d_scores) { set_error("amos_mask_person_masks_scores_device: invalid argument"); return AMOS_ERR_INVALID; }
    return person_masks(stream, d_loc, nullptr, d_scores, d_coef, d_priors, d_proto, batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w, out_h, out_w,
                        d_workspace, workspace_bytes, d_masks, d_found);
}

static int person_masks(void *stream, const float *d_loc, const float *d_conf, const float *d_scores, const float *d_coef, const float *d_priors, const float *d_proto,
                        int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w, int out_h, int out_w, void *d_workspace,
                        size_t workspace_bytes, uint8_t *d_masks, uint8_t *d_found)
{
    const int B = batch, P = n_priors, C1 = n_classes_with_background, C = C1 - 1, D = mask_dim, k = AMOS_MASK_NMS_TOP_K, nd = AMOS_MASK_TOP_K_DISPLAY;
    if (!d_loc || (!d_conf && !d_scores) || !d_coef || !d_priors || !d_proto || !d_workspace || !d_masks || !d_found || B < 1 || B > 65535 || P < k || C1 < 2 || C1 > 200 ||
        D < 4 || D % 4 != 0 || proto_h < 1 || proto_w < 1 || out_h < 1 || out_w < 1 || (size_t)C * k * 4 > 64 * 1024 ||
        ((uintptr_t)d_loc | (uintptr_t)d_priors | (uintptr_t)d_proto | (uintptr_t)d_coef | (uintptr_t)d_workspace) % 16 != 0) {
        set_error("amos_mask_person_masks_device: invalid argument (16-byte aligned tensors, mask_dim %% 4 == 0, at most %d class lists of %d)", 16384 / k, k);
        return AMOS_ERR_INVALID;
    }
    const PostLayout l = post_layout(B, P, C1, D, proto_h, proto_w);
    if (workspace_bytes < l.total) { set_error("amos_mask_person_masks_device: workspace of %zu bytes, %zu needed", workspace_bytes, l.total); return AMOS_ERR_CAPACITY; }
    uint8_t *ws = (uint8_t *)d_workspace;
    float4 *boxes = (float4 *)(ws + l.boxes);
    const float *cls = d_scores ? d_scores : (const float *)(ws + l.cls);
    float *topv = (float *)(ws + l.topv), *alive = (float *)(ws + l.alive), *selCoef = (float *)(ws + l.selCoef);
    long long *topi = (long long *)(ws + l.topi);
    float4 *selRect = (float4 *)(ws + l.selRect);
    uint8_t *flags = ws + l.flags;
    float *masks = (float *)(ws + l.masks);
    hipStream_t st = (hipStream_t)stream;
    const int total = B * P;
    hipLaunchKernelGGL(k_decode_boxes, dim3((total + 255) / 256), dim3(256), 0, st, (const float4 *)d_loc, (const float4 *)d_priors, boxes, P, total);
    int rc = d_scores ? AMOS_OK : amos_mask_class_scores_device(stream, d_conf, (float *)(ws + l.cls), B, P, C1, AMOS_MASK_CONF_THRESH);
    if (rc != AMOS_OK) return rc;
    rc = amos_mask_topk_rows_sparse_device(stream, cls, topv, topi, B * C, P, k, -1.f);  // (k_class_scores writes -1 for every prior under the threshold)
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_nms_alive, dim3(B * C), dim3(256), 0, st, boxes, topi, topv, alive, k, P, C, AMOS_MASK_NMS_THRESH);
    hipLaunchKernelGGL(k_select_display, dim3(B), dim3(256), 0, st, alive, topi, boxes, d_coef, selCoef, selRect, flags, d_found, C * k, k,
                       P, D, nd, AMOS_MASK_SCORE_THRESHOLD, AMOS_MASK_PERSON_CLASS, proto_w, proto_h);
    hipLaunchKernelGGL(k_assemble_masks, dim3((proto_h * proto_w + 255) / 256, nd, B), dim3(256), 0, st, d_proto, selCoef, selRect, flags, masks, proto_h, proto_w, D, nd);
    AMOS_HIP_CHECK(hipGetLastError());
    return amos_mask_person_mask_device(stream, masks, flags, d_masks, B, nd, proto_h, proto_w, out_h, out_w);
}

}  // extern "C"
