"""oracle/flow_oracle.py -- numpy restatement of the reference's own per-point arithmetic inside
Tracking::GetSceneFlowObj (src/Tracking.cc:850-1186).  TEST INFRASTRUCTURE ONLY (checker of amos-slam_amd/csrc/amos_flow.hip):
nothing in the product imports it.  Parity: these lines are the reference's code, not OpenCV's, except the two cv::Mat
expressions in scene_flow (one gemm each: double accumulation, one rounding -- restated, unpinned like every OpenCV step)."""
import numpy as np


def flow_check(last, cur, pre, nxt, state):
    """Tracking.cc:902-925: border test (limit_edge_corner = 5) and the 3 x 3 sum of absolute differences (> 2520 rejects)."""
    rows, cols = cur.shape
    out = np.array(state, np.uint8).copy()
    for i in range(len(pre)):
        x1, y1, x2, y2 = int(pre[i][0]), int(pre[i][1]), int(nxt[i][0]), int(nxt[i][1])  # float -> int truncates
        if x1 < 5 or x1 >= cols - 5 or x2 < 5 or x2 >= cols - 5 or y1 < 5 or y1 >= rows - 5 or y2 < 5 or y2 >= rows - 5:
            out[i] = 0
            continue
        a = last[y1 - 1:y1 + 2, x1 - 1:x1 + 2].astype(np.int64)
        b = cur[y2 - 1:y2 + 2, x2 - 1:x2 + 2].astype(np.int64)
        if np.abs(a - b).sum() > 2520:
            out[i] = 0
    return out


def epipolar(F, pre, nxt, state=None):
    """Tracking.cc:930-935: A, B, C = rows of F times (x, y, 1), evaluated left to right in doubles; dd = |A x' + B y' + C| / sqrt(A^2 + B^2)."""
    F = np.asarray(F, np.float64).reshape(3, 3)
    px, py = pre[:, 0].astype(np.float64), pre[:, 1].astype(np.float64)
    qx, qy = nxt[:, 0].astype(np.float64), nxt[:, 1].astype(np.float64)
    A = (F[0, 0] * px + F[0, 1] * py) + F[0, 2]
    B = (F[1, 0] * px + F[1, 1] * py) + F[1, 2]
    C = (F[2, 0] * px + F[2, 1] * py) + F[2, 2]
    dd = np.abs((A * qx + B * qy) + C) / np.sqrt(A * A + B * B)
    if state is not None:
        dd = np.where(np.asarray(state) != 0, dd, -1.0)
    return dd


def _gemm(R, x, t):
    return (R.astype(np.float64) @ x.astype(np.float64) + t.astype(np.float64)).astype(np.float32)


def scene_flow(depth_last, depth_cur, match_pre, match_cur, cx, cy, invfx, invfy, Tlw, Rwc, Ow):
    """Tracking.cc:955-990 and 1153-1183.  Returns [n, 8]: pre_3d, cur_3d, sf_norm, valid."""
    f32 = np.float32
    Tlw = np.asarray(Tlw, f32).reshape(3, 4)
    Rlw, tlw = Tlw[:, :3], Tlw[:, 3]
    Rwl = Rlw.T.copy()
    twl = (-(Rlw.T.astype(np.float64) @ tlw.astype(np.float64))).astype(f32)  # -Rlw.t() * tlw: gemm with alpha = -1
    Rwc, Ow = np.asarray(Rwc, f32).reshape(3, 3), np.asarray(Ow, f32)
    out = np.zeros((len(match_pre), 8), f32)
    for i in range(len(match_pre)):
        z1 = depth_last[int(match_pre[i][1]), int(match_pre[i][0])]
        z2 = depth_cur[int(match_cur[i][1]), int(match_cur[i][0])]
        if not (z1 > 0 and z2 > 0):
            continue
        x = f32(f32(f32(match_pre[i][0] - f32(cx)) * z1) * f32(invfx))
        y = f32(f32(f32(match_pre[i][1] - f32(cy)) * z1) * f32(invfy))
        p = _gemm(Rwl, np.array([x, y, z1], f32), twl)
        xc = f32(f32(f32(match_cur[i][0] - f32(cx)) * z1) * f32(invfx))   # z1, as the reference writes it (:1160)
        yc = f32(f32(f32(match_cur[i][1] - f32(cy)) * z1) * f32(invfy))
        c = _gemm(Rwc, np.array([xc, yc, z2], f32), Ow)
        fx, fz = f32(p[0] - c[0]), f32(p[2] - c[2])
        out[i] = [p[0], p[1], p[2], c[0], c[1], c[2], np.sqrt(f32(f32(fx * fx) + f32(fz * fz)), dtype=f32), 1.0]
    return out


def fundamental_errors(F, p1, p2):
    """OpenCV 4.5 FMEstimatorCallback::computeError (modules/calib3d/src/fundam.cpp) for ONE hypothesis F (3 x 3): the symmetric squared
    epipolar distance of every correspondence, in doubles left to right, stored as float32 -- what cv::findFundamentalMat's RANSAC
    (Tracking.cc:927, 945) tests against threshold^2.  Restated from the published source: parity unpinned like the other OpenCV stages."""
    F = np.asarray(F, np.float64).reshape(9)
    x1, y1 = p1[:, 0].astype(np.float64), p1[:, 1].astype(np.float64)
    x2, y2 = p2[:, 0].astype(np.float64), p2[:, 1].astype(np.float64)
    a = (F[0] * x1 + F[1] * y1) + F[2]
    b = (F[3] * x1 + F[4] * y1) + F[5]
    c = (F[6] * x1 + F[7] * y1) + F[8]
    s2 = 1.0 / (a * a + b * b)
    d2 = (x2 * a + y2 * b) + c
    a = (F[0] * x2 + F[3] * y2) + F[6]
    b = (F[1] * x2 + F[4] * y2) + F[7]
    c = (F[2] * x2 + F[5] * y2) + F[8]
    s1 = 1.0 / (a * a + b * b)
    d1 = (x1 * a + y1 * b) + c
    return np.maximum((d1 * d1) * s1, (d2 * d2) * s2).astype(np.float32)


def pnp_errors(R, t, obj, img, fx, fy, cx, cy):
    """OpenCV 4.5 PnPRansacCallback::computeError without distortion (cv::solvePnPRansac, Tracking.cc:1006): cv::projectPoints' arithmetic in
    doubles (X = R x + t left to right, x' = X * (1 / Z), u = x' fx + cx as float32), then the squared distance to the image point in float32."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3)
    X, Y, Z = (obj[:, k].astype(np.float64) for k in range(3))
    xc = ((R[0, 0] * X + R[0, 1] * Y) + R[0, 2] * Z) + t[0]
    yc = ((R[1, 0] * X + R[1, 1] * Y) + R[1, 2] * Z) + t[1]
    zc = ((R[2, 0] * X + R[2, 1] * Y) + R[2, 2] * Z) + t[2]
    with np.errstate(divide="ignore"):
        iz = np.where(zc != 0.0, 1.0 / zc, 1.0)
    u = ((xc * iz) * fx + cx).astype(np.float32)
    v = ((yc * iz) * fy + cy).astype(np.float32)
    dx, dy = img[:, 0].astype(np.float32) - u, img[:, 1].astype(np.float32) - v
    return (dx * dx + dy * dy).astype(np.float32)
