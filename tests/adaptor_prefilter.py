"""Checker-side restatement (numpy) of what the reference's ten ORBmatcher searches do BEFORE their candidate loops: the
geometric pre-filter of every map point -- projection, image bounds, distance invariance, viewing angle, predicted level
(src/ORBmatcher.cc, the lines cited per function).  The tests feed the queries built here to the oracle's search
restatements (orc_search_*) and hold the reference-signature adaptors (amos-slam_amd/host/ORBmatcher_adaptors.h) to the
result.  Test infrastructure, like oracle/.

Arithmetic: a CV_32F expression `R*x+t` is ONE OpenCV gemm (double accumulation, a single rounding to float); `cv::norm`
and `Mat::dot` accumulate in double; everything the reference writes on `float` variables rounds after every operation
(the reference is built without -ffast-math; products of floats inside a float expression are float).
"""
import numpy as np

f32, f64 = np.float32, np.float64


def _cols(x):
    x = np.asarray(x, f32).reshape(-1, 3)
    return x[:, 0].astype(f64), x[:, 1].astype(f64), x[:, 2].astype(f64)


def transform(R, t, x):
    """R*x+t for n points [n, 3] -> float32 [n, 3]; left-to-right double sums like a gemm row."""
    R, t = np.asarray(R, f32).astype(f64), np.asarray(t, f32).astype(f64)
    a, b, c = _cols(x)
    return np.stack([(R[r, 0] * a + R[r, 1] * b + R[r, 2] * c + t[r]).astype(f32) for r in range(3)], 1)


def centre(R, t):
    """-R^T * t (KeyFrame::SetPose's Ow; `-Rcw.t()*tcw` in the searches)"""
    R, t = np.asarray(R, f32).astype(f64), np.asarray(t, f32).astype(f64)
    return np.array([-(R[0, c] * t[0] + R[1, c] * t[1] + R[2, c] * t[2]) for c in range(3)]).astype(f32)


def pose(T):
    T = np.asarray(T, f32).reshape(4, 4)
    return T[:3, :3].copy(), T[:3, 3].copy()


def unscaled(Scw):
    """Scw -> (Rcw, tcw): scw = sqrt(row0 . row0); Rcw = sRcw / scw, tcw = st / scw (Mat / float multiplies by 1 / s), :397-400, :1188-1191"""
    R, t = pose(Scw)
    r0 = R[0].astype(f64)
    scw = f32(np.sqrt(r0[0] * r0[0] + r0[1] * r0[1] + r0[2] * r0[2]))
    inv = f32(1.0 / f64(scw))
    return (R * inv).astype(f32), (t * inv).astype(f32)


def norm(v):
    a, b, c = _cols(v)
    return np.sqrt(a * a + b * b + c * c)   # double


def project_kf(cam, pc):
    """invz = 1 / z; x = X * invz; u = fx * x + cx (:433-438, :1051-1055, :1218-1223, :1368-1373)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (1.0 / pc[:, 2].astype(f64)).astype(f32)
        x, y = pc[:, 0] * invz, pc[:, 1] * invz
        u, v = f32(cam.fx) * x + f32(cam.cx), f32(cam.fy) * y + f32(cam.cy)
    return u.astype(f32), v.astype(f32), invz


def in_image(cam, u, v):
    """KeyFrame::IsInImage"""
    return (u >= f32(cam.min_x)) & (u < f32(cam.max_x)) & (v >= f32(cam.min_y)) & (v < f32(cam.max_y))


def keyframe_queries(hb, cam, R, t, Ow, world, normal, min_dist, max_dist, desc, order, keep, with_right, sf1, n_levels):
    """The pre-filter three keyframe-side searches share (Fuse :1033-1080, Fuse(Scw) :1199-1244, SearchByProjection(pKF, Scw) :410-455).
    `order`: point indices in call order; `keep[j]`: entry j survives the isBad / already-found / IsInKeyFrame tests.
    Returns (queries as hb.WINDOW_QUERY with src = position in `order`, positions)."""
    order = np.asarray(order)
    pw = world[order]
    pc = transform(R, t, pw)
    u, v, invz = project_kf(cam, pc)
    PO = (pw - Ow[None, :]).astype(f32)
    dist = norm(PO).astype(f32)
    lo, hi = f32(0.8) * min_dist[order], f32(1.2) * max_dist[order]
    a, b, c = _cols(PO)
    n1, n2, n3 = _cols(normal[order])
    cosine_ok = ~((a * n1 + b * n2 + c * n3) < 0.5 * dist.astype(f64))
    ok = np.asarray(keep, bool) & ~(pc[:, 2] < 0) & in_image(cam, u, v) & ~(dist < lo) & ~(dist > hi) & cosine_ok
    pos = np.nonzero(ok)[0]
    q = np.zeros(len(pos), hb.WINDOW_QUERY)
    q["u"], q["v"] = u[pos], v[pos]
    q["ur"] = (u[pos] - f32(cam.mbf) * invz[pos]).astype(f32) if with_right else 0
    q["level"] = hb.standin_predict_scale(max_dist[order][pos], dist[pos], sf1, n_levels)
    q["src"], q["desc"] = pos, desc[order][pos]
    return q, pos


def reloc_queries(hb, cam, T_cur, world, min_dist, max_dist, desc, feat_point, feat_ok, kf_angles, sf1, n_levels):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), :1745-1790: one query per keyframe feature with a usable point."""
    R, t = pose(T_cur)
    Ow = centre(R, t)
    feats = np.nonzero(feat_ok)[0]
    pidx = feat_point[feats]
    pw = world[pidx]
    pc = transform(R, t, pw)
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (1.0 / pc[:, 2].astype(f64)).astype(f32)
        u = (f32(cam.fx) * pc[:, 0]) * invz + f32(cam.cx)
        v = (f32(cam.fy) * pc[:, 1]) * invz + f32(cam.cy)
    dist = norm((pw - Ow[None, :]).astype(f32)).astype(f32)
    ok = ~(u < f32(cam.min_x)) & ~(u > f32(cam.max_x)) & ~(v < f32(cam.min_y)) & ~(v > f32(cam.max_y))
    ok &= ~(dist < f32(0.8) * min_dist[pidx]) & ~(dist > f32(1.2) * max_dist[pidx])
    sel = np.nonzero(ok)[0]
    q = np.zeros(len(sel), hb.KF_QUERY)
    q["u"], q["v"] = u[sel], v[sel]
    q["level"] = hb.standin_predict_scale(max_dist[pidx][sel], dist[sel], sf1, n_levels)
    q["angle"], q["desc"] = kf_angles[feats][sel], desc[pidx][sel]
    return q, pidx[sel]


def sim3_hops(s12, R12, t12):
    """sR12 = s12 * R12; sR21 = (1.0 / s12) * R12.t(); t21 = -sR21 * t12 (:1331-1335).  Returns (c12 R, t), (c21 R, t)."""
    R12, t12 = np.asarray(R12, f32).reshape(3, 3), np.asarray(t12, f32).reshape(3)
    sR12 = (f64(f32(s12)) * R12.astype(f64)).astype(f32)
    sR21 = ((1.0 / f64(f32(s12))) * R12.T.astype(f64)).astype(f32)
    a = sR21.astype(f64)
    t21 = np.array([-(a[r, 0] * f64(t12[0]) + a[r, 1] * f64(t12[1]) + a[r, 2] * f64(t12[2])) for r in range(3)]).astype(f32)
    return (sR12, t12.copy()), (sR21, t21)


def sim3_direction(hb, cam1, cam_into, T_from, hop, world, min_dist, max_dist, desc, feat_point, feat_ok, sf1, n_levels):
    """One direction of SearchBySim3 (:1361-1393 / :1441-1473): the points of one keyframe through its world pose and the Sim3 hop into
    the other; the projection uses pKF1's intrinsics both ways (:1316-1319), the bounds are the target's."""
    R, t = pose(T_from)
    feats = np.nonzero(feat_ok)[0]
    pidx = feat_point[feats]
    pc = transform(hop[0], hop[1], transform(R, t, world[pidx]))
    u, v, _ = project_kf(cam1, pc)
    dist = norm(pc).astype(f32)
    ok = ~(pc[:, 2] < 0) & in_image(cam_into, u, v) & ~(dist < f32(0.8) * min_dist[pidx]) & ~(dist > f32(1.2) * max_dist[pidx])
    sel = np.nonzero(ok)[0]
    q = np.zeros(len(sel), hb.WINDOW_QUERY)
    q["u"], q["v"], q["ur"] = u[sel], v[sel], 0
    q["level"] = hb.standin_predict_scale(max_dist[pidx][sel], dist[sel], sf1, n_levels)
    q["src"], q["desc"] = feats[sel], desc[pidx][sel]
    return q


def epipole(cam2, T1, T2):
    """SearchForTriangulation :818-826: pKF1's camera centre in pKF2's image."""
    R1, t1 = pose(T1)
    R2, t2 = pose(T2)
    C2 = transform(R2, t2, centre(R1, t1)[None, :])[0]
    invz = f32(1.0) / C2[2]
    return f32(f32(cam2.fx) * C2[0] * invz + f32(cam2.cx)), f32(f32(cam2.fy) * C2[1] * invz + f32(cam2.cy))
