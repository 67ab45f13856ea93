// amos_flow.hip -- the point-parallel arithmetic of Tracking::GetSceneFlowObj (src/Tracking.cc:850-1186) that is the
// reference's OWN code: everything between its OpenCV calls.
//   k_flow_check      :902-925   per tracked point: the 5-px border test on both positions, the 3 x 3 sum of absolute
//                                gray differences (> 2520 rejects), result = the state the match lists are built from
//   k_epipolar        :928-946, 1141-1152   distance of the tracked position from the epipolar line F * p (doubles)
//   k_scene_flow_3d   :955-990, 1153-1183   back-projection of a match with the two depth maps, last / current camera to
//                                world (cv::Mat expressions: one gemm = double accumulation, one rounding), 3-D flow norm
// goodFeaturesToTrack, cornerSubPix, calcOpticalFlowPyrLK, findFundamentalMat and solvePnPRansac are OpenCV (the two
// RANSACs draw from OpenCV's RNG): they stay with the caller -- DESIGN.md section 7.
#include "amos_common.h"

#include <algorithm>

namespace amos {

struct FlowPoint {
    float x, y;
};

__global__ __launch_bounds__(256) void k_flow_check(const uint8_t *__restrict__ last, const uint8_t *__restrict__ cur, size_t lastStride, size_t curStride,
                                                   int cols, int rows, const FlowPoint *__restrict__ pre, const FlowPoint *__restrict__ next,
                                                   const uint8_t *__restrict__ stateIn, int n, uint8_t *__restrict__ stateOut)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int lim = 5;  // limit_edge_corner
    const int x1 = (int)pre[i].x, y1 = (int)pre[i].y, x2 = (int)next[i].x, y2 = (int)next[i].y;
    uint8_t st = stateIn[i];
    if (x1 < lim || x1 >= cols - lim || x2 < lim || x2 >= cols - lim || y1 < lim || y1 >= rows - lim || y2 < lim || y2 >= rows - lim) {
        stateOut[i] = 0;
        return;
    }
    int sum = 0;  // the reference sums exact small integers into a double
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++)
            sum += abs((int)last[(size_t)(y1 + dy) * lastStride + x1 + dx] - (int)cur[(size_t)(y2 + dy) * curStride + x2 + dx]);
    if (sum > 2520) st = 0;
    stateOut[i] = st;
}

__global__ __launch_bounds__(256) void k_epipolar(const double *__restrict__ F, const FlowPoint *__restrict__ pre, const FlowPoint *__restrict__ next,
                                                 const uint8_t *__restrict__ state, int n, double *__restrict__ dd)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (state && state[i] == 0) { dd[i] = -1.0; return; }
    const double px = pre[i].x, py = pre[i].y, qx = next[i].x, qy = next[i].y;
    // A = F00 * x + F01 * y + F02 evaluated left to right, no fused multiply-add
    const double A = __dadd_rn(__dadd_rn(__dmul_rn(F[0], px), __dmul_rn(F[1], py)), F[2]);
    const double B = __dadd_rn(__dadd_rn(__dmul_rn(F[3], px), __dmul_rn(F[4], py)), F[5]);
    const double C = __dadd_rn(__dadd_rn(__dmul_rn(F[6], px), __dmul_rn(F[7], py)), F[8]);
    const double num = fabs(__dadd_rn(__dadd_rn(__dmul_rn(A, qx), __dmul_rn(B, qy)), C));
    dd[i] = __ddiv_rn(num, __dsqrt_rn(__dadd_rn(__dmul_rn(A, A), __dmul_rn(B, B))));
}

// ---- hypothesis scoring for the RANSACs of GetSceneFlowObj (Tracking.cc:927, 945: cv::findFundamentalMat(..., FM_RANSAC, 0.1, 0.99);
// :1006: cv::solvePnPRansac(..., 500, 0.4, 0.98, inliers, SOLVEPNP_P3P)).  The minimal solvers, the sampling and the iteration-count
// update stay with the caller; what the device takes is the part that touches every point: the error of every
// correspondence under every hypothesis, the inlier test and the inlier counts -- one work-group per hypothesis, counts by a reduction
// (no atomics: deterministic, nothing to zero).
//
// Fundamental matrix: OpenCV 4.5's FMEstimatorCallback::computeError (modules/calib3d/src/fundam.cpp) restated: in doubles, left to right,
//   a = F0 x1 + F1 y1 + F2, b = F3 x1 + F4 y1 + F5, c = F6 x1 + F7 y1 + F8;  s2 = 1 / (a a + b b);  d2 = x2 a + y2 b + c;
//   a = F0 x2 + F3 y2 + F6, b = F1 x2 + F4 y2 + F7, c = F2 x2 + F5 y2 + F8;  s1 = 1 / (a a + b b);  d1 = x1 a + y1 b + c;
//   err = (float) max(d1 d1 s1, d2 d2 s2);  inlier <=> err <= (float)(threshold * threshold)   (RANSACPointSetRegistrator::findInliers).
// OpenCV-derived, hence parity unpinned like the other OpenCV stages; the GPU tests hold it to the numpy restatement bit for bit.
__global__ __launch_bounds__(256) void k_fundamental_score(const double *__restrict__ Fs, const FlowPoint *__restrict__ p1, const FlowPoint *__restrict__ p2, int n,
                                                          float thresh2, float *__restrict__ err, int *__restrict__ inliers, uint8_t *__restrict__ mask)
{
    __shared__ int sCount[4];
    const int h = blockIdx.x, t = threadIdx.x;
    const double *F = Fs + (size_t)h * 9;
    const double F0 = F[0], F1 = F[1], F2 = F[2], F3 = F[3], F4 = F[4], F5 = F[5], F6 = F[6], F7 = F[7], F8 = F[8];
    int count = 0;
    for (int i = t; i < n; i += 256) {
        const double x1 = p1[i].x, y1 = p1[i].y, x2 = p2[i].x, y2 = p2[i].y;
        double a = __dadd_rn(__dadd_rn(__dmul_rn(F0, x1), __dmul_rn(F1, y1)), F2);
        double b = __dadd_rn(__dadd_rn(__dmul_rn(F3, x1), __dmul_rn(F4, y1)), F5);
        double c = __dadd_rn(__dadd_rn(__dmul_rn(F6, x1), __dmul_rn(F7, y1)), F8);
        const double s2 = __ddiv_rn(1.0, __dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)));
        const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(x2, a), __dmul_rn(y2, b)), c);
        a = __dadd_rn(__dadd_rn(__dmul_rn(F0, x2), __dmul_rn(F3, y2)), F6);
        b = __dadd_rn(__dadd_rn(__dmul_rn(F1, x2), __dmul_rn(F4, y2)), F7);
        c = __dadd_rn(__dadd_rn(__dmul_rn(F2, x2), __dmul_rn(F5, y2)), F8);
        const double s1 = __ddiv_rn(1.0, __dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)));
        const double d1 = __dadd_rn(__dadd_rn(__dmul_rn(x1, a), __dmul_rn(y1, b)), c);
        const double e1 = __dmul_rn(__dmul_rn(d1, d1), s1), e2 = __dmul_rn(__dmul_rn(d2, d2), s2);
        const float e = (float)(e1 > e2 ? e1 : (e2 > e1 ? e2 : (e1 != e1 ? e2 : e1)));  // std::max(a, b) = (a < b) ? b : a
        const bool in = e <= thresh2;
        if (err) err[(size_t)h * n + i] = e;
        if (mask) mask[(size_t)h * n + i] = in ? 1 : 0;
        count += in ? 1 : 0;
    }
    for (int d = 32; d > 0; d >>= 1) count += __shfl_xor(count, d, 64);
    if ((t & 63) == 0) sCount[t >> 6] = count;
    __syncthreads();
    if (t == 0) inliers[h] = sCount[0] + sCount[1] + sCount[2] + sCount[3];
}

// Pose hypotheses (R | t): OpenCV's PnPRansacCallback::computeError restated for zero distortion: the object point goes through
// cv::projectPoints' arithmetic in doubles -- X = R00 x + R01 y + R02 z + t0 (left to right) ..., x' = X / Z, y' = Y / Z (as X * (1 / Z)),
// u = fx x' + cx, v = fy y' + cy, stored as float -- and err = (float) ||image point - (u, v)||^2 in float (Matx21f norm L2SQR);
// inlier <=> err <= (float)(reprojectionError * reprojectionError).
__global__ __launch_bounds__(256) void k_pnp_score(const double *__restrict__ Rts, const float *__restrict__ obj, const FlowPoint *__restrict__ img, int n, double fx,
                                                  double fy, double cx, double cy, float thresh2, float *__restrict__ err, int *__restrict__ inliers,
                                                  uint8_t *__restrict__ mask)
{
    __shared__ int sCount[4];
    const int h = blockIdx.x, t = threadIdx.x;
    const double *M = Rts + (size_t)h * 12;
    double R[9], T[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = M[k];
#pragma unroll
    for (int k = 0; k < 3; k++) T[k] = M[9 + k];
    int count = 0;
    for (int i = t; i < n; i += 256) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        const double xc = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[0], X), __dmul_rn(R[1], Y)), __dmul_rn(R[2], Z)), T[0]);
        const double yc = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[3], X), __dmul_rn(R[4], Y)), __dmul_rn(R[5], Z)), T[1]);
        double zc = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[6], X), __dmul_rn(R[7], Y)), __dmul_rn(R[8], Z)), T[2]);
        zc = zc != 0.0 ? __ddiv_rn(1.0, zc) : 1.0;  // cvProjectPoints2: z = z ? 1. / z : 1
        const double xn = __dmul_rn(xc, zc), yn = __dmul_rn(yc, zc);
        const float u = (float)__dadd_rn(__dmul_rn(xn, fx), cx), v = (float)__dadd_rn(__dmul_rn(yn, fy), cy);
        const float dx = __fsub_rn(img[i].x, u), dy = __fsub_rn(img[i].y, v);
        const float e = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        const bool in = e <= thresh2;
        if (err) err[(size_t)h * n + i] = e;
        if (mask) mask[(size_t)h * n + i] = in ? 1 : 0;
        count += in ? 1 : 0;
    }
    for (int d = 32; d > 0; d >>= 1) count += __shfl_xor(count, d, 64);
    if ((t & 63) == 0) sCount[t >> 6] = count;
    __syncthreads();
    if (t == 0) inliers[h] = sCount[0] + sCount[1] + sCount[2] + sCount[3];
}

struct SceneFlowArgs {
    float cx, cy, invfx, invfy;
    float Rwl[9], twl[3];  // last camera -> world (Rlw^T, -Rlw^T tlw as floats, Tracking.cc:970-973)
    float Rwc[9], Ow[3];   // current camera -> world (Frame::mRwc, mOw)
};

__device__ __forceinline__ float gemm_row(const float *R, int r, float x, float y, float z, float t)
{
    return (float)((double)R[3 * r] * x + (double)R[3 * r + 1] * y + (double)R[3 * r + 2] * z + (double)t);
}

// out per match: pre_3d (3), cur_3d (3), sf_norm, valid (z1 > 0 && z2 > 0)
__global__ __launch_bounds__(256) void k_scene_flow_3d(const float *__restrict__ depthLast, const float *__restrict__ depthCur, size_t lastStride,
                                                      size_t curStride, const FlowPoint *__restrict__ matchPre, const FlowPoint *__restrict__ matchCur,
                                                      int n, const SceneFlowArgs a, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float z1 = depthLast[(size_t)(int)matchPre[i].y * lastStride + (int)matchPre[i].x];
    const float z2 = depthCur[(size_t)(int)matchCur[i].y * curStride + (int)matchCur[i].x];
    float *o = out + (size_t)i * 8;
    if (!(z1 > 0 && z2 > 0)) {
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = 0.f;
        return;
    }
    // :960-961
    const float xl = __fmul_rn(__fmul_rn(__fsub_rn(matchPre[i].x, a.cx), z1), a.invfx);
    const float yl = __fmul_rn(__fmul_rn(__fsub_rn(matchPre[i].y, a.cy), z1), a.invfy);
    const float p0 = gemm_row(a.Rwl, 0, xl, yl, z1, a.twl[0]), p1 = gemm_row(a.Rwl, 1, xl, yl, z1, a.twl[1]), p2 = gemm_row(a.Rwl, 2, xl, yl, z1, a.twl[2]);
    // :1160-1164 (the reference scales the CURRENT pixel by z1 and stacks z2: restated as written)
    const float xc = __fmul_rn(__fmul_rn(__fsub_rn(matchCur[i].x, a.cx), z1), a.invfx);
    const float yc = __fmul_rn(__fmul_rn(__fsub_rn(matchCur[i].y, a.cy), z1), a.invfy);
    const float c0 = gemm_row(a.Rwc, 0, xc, yc, z2, a.Ow[0]), c1 = gemm_row(a.Rwc, 1, xc, yc, z2, a.Ow[1]), c2 = gemm_row(a.Rwc, 2, xc, yc, z2, a.Ow[2]);
    const float fx = __fsub_rn(p0, c0), fz = __fsub_rn(p2, c2);
    o[0] = p0; o[1] = p1; o[2] = p2; o[3] = c0; o[4] = c1; o[5] = c2;
    // sf_norm uses x and z only (:1176).  std::sqrt(float) is correctly rounded; the device's single-precision square root is
    // not, the double one is, and rounding a double square root of a float to float is exact (53 >= 2 * 24 + 2 bits)
    o[6] = (float)__dsqrt_rn((double)__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fz, fz)));
    o[7] = 1.f;
}


// ---------------------------------------------------------------------------------------------------------------
// cv::calcOpticalFlowPyrLK as Tracking::GetSceneFlowObj calls it (Tracking.cc:896: 22 x 22 window, maxLevel 5, 20 iterations /
// epsilon 0.01, minEigThreshold 1e-4), restated from OpenCV 4.5's published algorithm like its CPU checker (lk_oracle.c of the test infrastructure: PARITY
// UNPINNED; the float accumulation order is defined as the scalar path's, row by row, left to right).
//   k_lk_level0 / k_lk_pyrdown   buildOpticalFlowPyramid: levels with a REFLECT_101 border of `win` pixels, pyrDown 5 x 5
//   k_lk_scharr                  calcScharrDeriv into a zero-bordered (dx, dy) plane
//   k_lk_track                   ONE WAVE PER POINT, all levels in one launch: the 484 window pixels are interpolated in
//                                parallel (14-bit fixed point), their products go to LDS and every lane adds them up in the
//                                defined order (same bits in all lanes: no broadcast, no divergence)
constexpr int kLkMaxLevels = 8;

struct LkLevel {
    int w, h, pw;        // level size, padded row pitch (w + 2 win)
    size_t imgOff;       // offset of the padded plane inside one image's pyramid (bytes)
    size_t derivOff;     // offset of the padded (dx, dy) plane (int16 pairs)
};
struct LkArgs {
    LkLevel lv[kLkMaxLevels];
    int top, win, maxCount;
    double epsilon2;
    float minEig;
    const uint8_t *prevPyr, *nextPyr;
    const short *deriv;
};

__device__ __forceinline__ int lk_refl(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

// grid = (ceil(pw * ph / 256), 2 images), block 256: padded level 0 from the gray frames
__global__ __launch_bounds__(256) void k_lk_level0(const uint8_t *__restrict__ g0, size_t s0, const uint8_t *__restrict__ g1, size_t s1, int w, int h, int win,
                                                  uint8_t *__restrict__ p0, uint8_t *__restrict__ p1)
{
    const int pw = w + 2 * win, ph = h + 2 * win, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pw * ph) return;
    const int y = i / pw - win, x = i - (i / pw) * pw - win;
    const uint8_t *g = blockIdx.y ? g1 : g0;
    const size_t st = blockIdx.y ? s1 : s0;
    (blockIdx.y ? p1 : p0)[i] = g[(size_t)lk_refl(y, h) * st + lk_refl(x, w)];
}

// padded level l from padded level l - 1 (both images): cv::pyrDown at the reflected destination coordinate
__global__ __launch_bounds__(256) void k_lk_pyrdown(const uint8_t *__restrict__ s0, const uint8_t *__restrict__ s1, int sw, int sh, int spw, int win,
                                                   uint8_t *__restrict__ d0, uint8_t *__restrict__ d1, int dw, int dh)
{
    const int pw = dw + 2 * win, ph = dh + 2 * win, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pw * ph) return;
    const int y = lk_refl(i / pw - win, dh), x = lk_refl(i - (i / pw) * pw - win, dw);
    const uint8_t *src = (blockIdx.y ? s1 : s0) + (size_t)win * spw + win;
    const int k[5] = {1, 4, 6, 4, 1};
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t *row = src + (size_t)lk_refl(2 * y + j - 2, sh) * spw;
        int hs = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) hs += k[q] * row[lk_refl(2 * x + q - 2, sw)];
        sum += k[j] * hs;
    }
    (blockIdx.y ? d1 : d0)[i] = (uint8_t)((sum + 128) >> 8);
}

// calcScharrDeriv on the interior of the padded previous-image level; the border of the derivative plane stays zero
__global__ __launch_bounds__(256) void k_lk_scharr(const uint8_t *__restrict__ img, int w, int h, int pw, int win, short *__restrict__ deriv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const uint8_t *base = img + (size_t)win * pw + win;
    auto px = [&](int yy, int xx) { return (int)base[(size_t)lk_refl(yy, h) * pw + lk_refl(xx, w)]; };
    auto t0 = [&](int xx) { return (int)(short)((px(y - 1, xx) + px(y + 1, xx)) * 3 + px(y, xx) * 10); };  // vertical [3 10 3]
    auto t1 = [&](int xx) { return (int)(short)(px(y + 1, xx) - px(y - 1, xx)); };                          // vertical [-1 0 1]
    short *d = deriv + 2 * ((size_t)(y + win) * pw + x + win);
    d[0] = (short)(t0(x + 1) - t0(x - 1));
    d[1] = (short)((t1(x + 1) + t1(x - 1)) * 3 + t1(x) * 10);
}

constexpr int kLkMaxWin = 22, kLkWinPx = kLkMaxWin * kLkMaxWin;

__device__ __forceinline__ float lk_div(float a, float b) { return (float)__ddiv_rn((double)a, (double)b); }   // correctly rounded
__device__ __forceinline__ float lk_sqrt(float a) { return (float)__dsqrt_rn((double)a); }

// sum of n floats of an LDS array in index order, identically in every lane
__device__ __forceinline__ float lk_ordered_sum(const float *v, int n)
{
    float s = 0.f;
    for (int i = 0; i < n; i++) s = __fadd_rn(s, v[i]);
    return s;
}

__global__ __launch_bounds__(256) void k_lk_track(const LkArgs a, const float *__restrict__ prevPts, int n, float *__restrict__ nextPts,
                                                 uint8_t *__restrict__ status, float *__restrict__ err)
{
    __shared__ short sI[4][kLkWinPx], sdI[4][2 * kLkWinPx];
    __shared__ float sP[4][3][kLkWinPx + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pt = blockIdx.x * 4 + wv;
    if (pt >= n) return;  // wave-uniform
    short *Iw = sI[wv], *dIw = sdI[wv];
    float *P0 = sP[wv][0], *P1 = sP[wv][1], *P2 = sP[wv][2];
    const int win = a.win, npx = win * win;
    const float half = __fmul_rn((float)(win - 1), 0.5f), scale20 = 1.f / (1 << 20);
    const float ptx = prevPts[2 * pt], pty = prevPts[2 * pt + 1];
    float outx = 0.f, outy = 0.f, errv = 0.f;
    int st = 1;
    auto weights = [](float fa, float fb, int &w00, int &w01, int &w10, int &w11) {
        w00 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), __fsub_rn(1.f, fb)), 16384.f));
        w01 = __float2int_rn(__fmul_rn(__fmul_rn(fa, __fsub_rn(1.f, fb)), 16384.f));
        w10 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), fb), 16384.f));
        w11 = 16384 - w00 - w01 - w10;
    };
    for (int level = a.top; level >= 0; level--) {
        const LkLevel L = a.lv[level];
        const uint8_t *I = a.prevPyr + L.imgOff, *J = a.nextPyr + L.imgOff;
        const short *dI = a.deriv + 2 * L.derivOff;
        const int pw = L.pw;
        const float sc = (float)(1. / (1 << level));
        float px = __fmul_rn(ptx, sc), py = __fmul_rn(pty, sc), nx, ny;
        if (level == a.top) { nx = px; ny = py; }
        else { nx = __fmul_rn(outx, 2.f); ny = __fmul_rn(outy, 2.f); }
        outx = nx; outy = ny;
        px = __fsub_rn(px, half); py = __fsub_rn(py, half);
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h) {
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }
        int w00, w01, w10, w11;
        weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy), w00, w01, w10, w11);
        for (int idx = lane; idx < npx; idx += 64) {  // the window of the first image and its derivatives
            const int y = idx / win, x = idx - y * win;
            const uint8_t *src = I + (size_t)(y + ipy + win) * pw + ipx + win + x;
            const short *ds = dI + 2 * ((size_t)(y + ipy + win) * pw + ipx + win + x);
            const int ival = (src[0] * w00 + src[1] * w01 + src[pw] * w10 + src[pw + 1] * w11 + (1 << 8)) >> 9;
            const int ix = (ds[0] * w00 + ds[2] * w01 + ds[2 * pw] * w10 + ds[2 * pw + 2] * w11 + (1 << 13)) >> 14;
            const int iy = (ds[1] * w00 + ds[3] * w01 + ds[2 * pw + 1] * w10 + ds[2 * pw + 3] * w11 + (1 << 13)) >> 14;
            Iw[idx] = (short)ival; dIw[2 * idx] = (short)ix; dIw[2 * idx + 1] = (short)iy;
            P0[idx] = (float)(ix * ix); P1[idx] = (float)(ix * iy); P2[idx] = (float)(iy * iy);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float A11 = __fmul_rn(lk_ordered_sum(P0, npx), scale20), A12 = __fmul_rn(lk_ordered_sum(P1, npx), scale20),
                    A22 = __fmul_rn(lk_ordered_sum(P2, npx), scale20);
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dA = __fsub_rn(A11, A22);
        const float minEig = lk_div(__fsub_rn(__fadd_rn(A22, A11), lk_sqrt(__fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
                                    (float)(2 * win * win));
        __builtin_amdgcn_wave_barrier();
        if (minEig < a.minEig || D < 1.1920929e-7f) {
            if (level == 0) st = 0;
            continue;
        }
        D = lk_div(1.f, D);
        nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.maxCount; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
                if (level == 0) st = 0;
                break;
            }
            weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny), w00, w01, w10, w11);
            for (int idx = lane; idx < npx; idx += 64) {
                const int y = idx / win, x = idx - y * win;
                const uint8_t *Jp = J + (size_t)(y + iny + win) * pw + inx + win + x;
                const int diff = ((Jp[0] * w00 + Jp[1] * w01 + Jp[pw] * w10 + Jp[pw + 1] * w11 + (1 << 8)) >> 9) - Iw[idx];
                P0[idx] = (float)(diff * dIw[2 * idx]); P1[idx] = (float)(diff * dIw[2 * idx + 1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float b1 = __fmul_rn(lk_ordered_sum(P0, npx), scale20), b2 = __fmul_rn(lk_ordered_sum(P1, npx), scale20);
            __builtin_amdgcn_wave_barrier();
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), D);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), D);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            outx = __fadd_rn(nx, half); outy = __fadd_rn(ny, half);
            if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= a.epsilon2) break;
            if (j > 0 && (double)fabsf(__fadd_rn(dx, pdx)) < 0.01 && (double)fabsf(__fadd_rn(dy, pdy)) < 0.01) {
                outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f)); outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && level == 0) {  // the residual of the final position
            const float ex = __fsub_rn(outx, half), ey = __fsub_rn(outy, half);
            const int iex = (int)floorf(ex), iey = (int)floorf(ey);
            if (iex < -win || iex >= L.w || iey < -win || iey >= L.h) { st = 0; continue; }
            weights(__fsub_rn(ex, (float)iex), __fsub_rn(ey, (float)iey), w00, w01, w10, w11);
            for (int idx = lane; idx < npx; idx += 64) {
                const int y = idx / win, x = idx - y * win;
                const uint8_t *Jp = J + (size_t)(y + iey + win) * pw + iex + win + x;
                const int diff = ((Jp[0] * w00 + Jp[1] * w01 + Jp[pw] * w10 + Jp[pw + 1] * w11 + (1 << 8)) >> 9) - Iw[idx];
                P0[idx] = fabsf((float)diff);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            errv = lk_div(__fmul_rn(lk_ordered_sum(P0, npx), 1.f), (float)(32 * win * win));  // errval * 1.f / (32 * w * h): a division
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) {
        nextPts[2 * pt] = outx; nextPts[2 * pt + 1] = outy;
        status[pt] = (uint8_t)st;
        if (err) err[pt] = errv;
    }
}

}  // namespace amos

using namespace amos;

extern "C" {

int amos_flow_check_device(void *stream, const uint8_t *d_last_gray, size_t last_stride, const uint8_t *d_cur_gray, size_t cur_stride, int cols, int rows,
                           const float *d_pre_xy, const float *d_next_xy, const uint8_t *d_state_in, int n, uint8_t *d_state_out)
{
    if (!d_last_gray || !d_cur_gray || !d_pre_xy || !d_next_xy || !d_state_in || !d_state_out || n < 0 || cols < 11 || rows < 11) { set_error("amos_flow_check_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_flow_check, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_last_gray, d_cur_gray, last_stride, cur_stride, cols, rows,
                       (const FlowPoint *)d_pre_xy, (const FlowPoint *)d_next_xy, d_state_in, n, d_state_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_flow_epipolar_device(void *stream, const double *d_F, const float *d_pre_xy, const float *d_next_xy, const uint8_t *d_state, int n, double *d_dd)
{
    if (!d_F || !d_pre_xy || !d_next_xy || !d_dd || n < 0) { set_error("amos_flow_epipolar_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_epipolar, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_F, (const FlowPoint *)d_pre_xy, (const FlowPoint *)d_next_xy,
                       d_state, n, d_dd);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_flow_fundamental_score_device(void *stream, const double *d_F, int n_hypotheses, const float *d_points1_xy, const float *d_points2_xy, int n,
                                       double threshold, float *d_err, int32_t *d_inliers, uint8_t *d_mask)
{
    if (!d_F || !d_points1_xy || !d_points2_xy || !d_inliers || n < 0 || n_hypotheses < 0 || !(threshold >= 0)) { set_error("amos_flow_fundamental_score_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_hypotheses == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_fundamental_score, dim3(n_hypotheses), dim3(256), 0, (hipStream_t)stream, d_F, (const FlowPoint *)d_points1_xy, (const FlowPoint *)d_points2_xy, n,
                       (float)(threshold * threshold), d_err, d_inliers, d_mask);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_flow_pnp_score_device(void *stream, const double *d_Rt, int n_hypotheses, const float *d_object_xyz, const float *d_image_xy, int n, double fx, double fy,
                               double cx, double cy, double reprojection_error, float *d_err, int32_t *d_inliers, uint8_t *d_mask)
{
    if (!d_Rt || !d_object_xyz || !d_image_xy || !d_inliers || n < 0 || n_hypotheses < 0 || !(reprojection_error >= 0)) { set_error("amos_flow_pnp_score_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_hypotheses == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_pnp_score, dim3(n_hypotheses), dim3(256), 0, (hipStream_t)stream, d_Rt, d_object_xyz, (const FlowPoint *)d_image_xy, n, fx, fy, cx, cy,
                       (float)(reprojection_error * reprojection_error), d_err, d_inliers, d_mask);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_flow_scene_flow_device(void *stream, const float *d_depth_last, size_t last_stride, const float *d_depth_cur, size_t cur_stride,
                                const float *d_match_pre_xy, const float *d_match_cur_xy, int n, const amos_scene_flow_camera *cam, float *d_out)
{
    if (!d_depth_last || !d_depth_cur || !d_match_pre_xy || !d_match_cur_xy || !cam || !d_out || n < 0) { set_error("amos_flow_scene_flow_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n == 0) return AMOS_OK;
    SceneFlowArgs a;
    a.cx = cam->cx; a.cy = cam->cy; a.invfx = cam->invfx; a.invfy = cam->invfy;
    // Rwl = Rlw^T, twl = -Rlw^T * tlw (one gemm, alpha = -1: double accumulation, one rounding)
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) a.Rwl[3 * r + c] = cam->Tlw[4 * c + r];
        a.twl[r] = (float)(-((double)cam->Tlw[r] * cam->Tlw[3] + (double)cam->Tlw[4 + r] * cam->Tlw[7] + (double)cam->Tlw[8 + r] * cam->Tlw[11]));
    }
    for (int k = 0; k < 9; k++) a.Rwc[k] = cam->Rwc[k];
    for (int k = 0; k < 3; k++) a.Ow[k] = cam->Ow[k];
    hipLaunchKernelGGL(k_scene_flow_3d, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_depth_last, d_depth_cur, last_stride, cur_stride,
                       (const FlowPoint *)d_match_pre_xy, (const FlowPoint *)d_match_cur_xy, n, a, d_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}


struct amos_lk {
    int device = 0, w = 0, h = 0, win = 22, maxLevel = 5, top = 0, maxPoints = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    LkArgs args{};
    uint8_t *dPrev = nullptr, *dNext = nullptr;
    short *dDeriv = nullptr;
    size_t pyrBytes = 0;
};

int amos_lk_create(int device, void *stream, int width, int height, int win_size, int max_level, amos_lk **out)
{
    if (!out || win_size < 3 || win_size > kLkMaxWin || max_level < 0 || max_level >= kLkMaxLevels || width <= win_size || height <= win_size) {
        set_error("amos_lk_create: invalid argument (3 <= win_size <= %d, frame larger than the window)", kLkMaxWin);
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_lk *k = new amos_lk();
    k->device = device; k->w = width; k->h = height; k->win = win_size; k->maxLevel = max_level;
    if (stream) k->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete k; return AMOS_ERR_DEVICE; }
        k->ownStream = true;
    }
    size_t off = 0;
    int w = width, h = height, level = 0;
    for (;; level++) {  // buildOpticalFlowPyramid's level count
        LkLevel &L = k->args.lv[level];
        L.w = w; L.h = h; L.pw = w + 2 * win_size;
        L.imgOff = off; L.derivOff = off;
        off += ((size_t)L.pw * (h + 2 * win_size) + 255) / 256 * 256;
        if (level == max_level) break;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_size || h <= win_size) break;
    }
    k->top = k->args.top = level;
    k->args.win = win_size;
    k->pyrBytes = off;
    hipError_t e = hipMalloc((void **)&k->dPrev, off + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&k->dNext, off + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&k->dDeriv, sizeof(short) * 2 * off + 64);
    if (e == hipSuccess) e = hipMemsetAsync(k->dDeriv, 0, sizeof(short) * 2 * off + 64, k->stream);  // the derivative border is zero and never written
    if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
    if (e != hipSuccess) { set_error("amos_lk_create: %s", hipGetErrorString(e)); amos_lk_destroy(k); return AMOS_ERR_DEVICE; }
    k->args.prevPyr = k->dPrev; k->args.nextPyr = k->dNext; k->args.deriv = k->dDeriv;
    *out = k;
    return AMOS_OK;
}

void amos_lk_destroy(amos_lk *k)
{
    if (!k) return;
    (void)hipSetDevice(k->device);
    if (k->stream) (void)hipStreamSynchronize(k->stream);
    for (void *p : {(void *)k->dPrev, (void *)k->dNext, (void *)k->dDeriv}) if (p) (void)hipFree(p);
    if (k->ownStream && k->stream) (void)hipStreamDestroy(k->stream);
    delete k;
}

void *amos_lk_stream(amos_lk *k) { return k ? (void *)k->stream : nullptr; }
int amos_lk_levels(const amos_lk *k) { return k ? k->top : AMOS_ERR_INVALID; }

int amos_lk_track_device(amos_lk *k, const uint8_t *d_prev_gray, size_t prev_stride, const uint8_t *d_next_gray, size_t next_stride, const float *d_prev_xy,
                         int n, int max_count, double epsilon, float min_eig_threshold, float *d_next_xy, uint8_t *d_status, float *d_err)
{
    if (!k || !d_prev_gray || !d_next_gray || !d_prev_xy || !d_next_xy || !d_status || n < 0 || prev_stride < (size_t)k->w || next_stride < (size_t)k->w) {
        set_error("amos_lk_track_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (n == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(k->device));
    const int win = k->win;
    const LkLevel &L0 = k->args.lv[0];
    hipLaunchKernelGGL(k_lk_level0, dim3((L0.pw * (L0.h + 2 * win) + 255) / 256, 2), dim3(256), 0, k->stream, d_prev_gray, prev_stride, d_next_gray, next_stride,
                       k->w, k->h, win, k->dPrev, k->dNext);
    for (int l = 1; l <= k->top; l++) {
        const LkLevel &S = k->args.lv[l - 1], &D = k->args.lv[l];
        hipLaunchKernelGGL(k_lk_pyrdown, dim3((D.pw * (D.h + 2 * win) + 255) / 256, 2), dim3(256), 0, k->stream, k->dPrev + S.imgOff, k->dNext + S.imgOff, S.w,
                           S.h, S.pw, win, k->dPrev + D.imgOff, k->dNext + D.imgOff, D.w, D.h);
    }
    for (int l = 0; l <= k->top; l++) {
        const LkLevel &L = k->args.lv[l];
        hipLaunchKernelGGL(k_lk_scharr, dim3((L.w * L.h + 255) / 256), dim3(256), 0, k->stream, k->dPrev + L.imgOff, L.w, L.h, L.pw, win, k->dDeriv + 2 * L.derivOff);
    }
    LkArgs a = k->args;
    a.maxCount = std::min(std::max(max_count, 0), 100);
    const double eps = std::min(std::max(epsilon, 0.), 10.);
    a.epsilon2 = eps * eps;
    a.minEig = min_eig_threshold;
    hipLaunchKernelGGL(k_lk_track, dim3((n + 3) / 4), dim3(256), 0, k->stream, a, d_prev_xy, n, d_next_xy, d_status, d_err);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
