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

}  // extern "C"
