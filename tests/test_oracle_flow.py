"""CPU: hand-computable cases for oracle/flow_oracle.py (the numpy restatement of Tracking::GetSceneFlowObj's own point
arithmetic, src/Tracking.cc:902-946, 955-990, 1153-1183) that checks amos_flow_*_device on the GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import flow_oracle as fo  # noqa: E402


def test_flow_check_border_and_sad():
    last = np.full((40, 60), 100, np.uint8)
    cur = last.copy()
    cur[20:23, 30:33] = 100 + 28  # 9 x 28 = 252 < 2520: kept; the threshold needs a mean difference above 280 (never, for u8)
    pre = np.array([[10, 10], [4.9, 10], [5, 10], [54.9, 10], [55, 10], [10, 34.9], [10, 35.0], [31, 21]], np.float32)
    nxt = pre.copy()
    got = fo.flow_check(last, cur, pre, nxt, np.ones(len(pre), np.uint8))
    assert got.tolist() == [1, 0, 1, 1, 0, 1, 0, 1]
    assert fo.flow_check(last, cur, pre, nxt, np.zeros(len(pre), np.uint8)).sum() == 0  # a lost track stays lost
    # 9 x 255 = 2295 <= 2520: the SAD test of the reference can never fire on 8-bit images; restated as written
    hi = np.full((40, 60), 255, np.uint8)
    lo = np.zeros((40, 60), np.uint8)
    assert fo.flow_check(hi, lo, pre[:1], pre[:1], np.ones(1, np.uint8))[0] == 1


def test_epipolar_distance_known_answers():
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float64)  # horizontal epipolar lines: l = (0, -1, y)
    pre = np.array([[10, 20], [300, 200.5]], np.float32)
    nxt = np.array([[50, 23], [10, 200.5]], np.float32)
    dd = fo.epipolar(F, pre, nxt)
    assert dd.tolist() == [3.0, 0.0]
    assert fo.epipolar(F, pre, nxt, np.array([0, 1])).tolist() == [-1.0, 0.0]


def test_scene_flow_identity_pose():
    depth = np.full((480, 640), 2.0, np.float32)
    depth[100, 100] = 0
    pre = np.array([[320, 240], [100, 100], [420.5, 240]], np.float32)
    cur = np.array([[320, 240], [200, 200], [420.5, 240]], np.float32)
    eye = np.eye(3, 4, dtype=np.float32)
    out = fo.scene_flow(depth, depth, pre, cur, 320.0, 240.0, 0.002, 0.002, eye, np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    assert out[0].tolist() == [0, 0, 2, 0, 0, 2, 0, 1]
    assert out[1].tolist() == [0] * 8  # z1 == 0
    np.testing.assert_allclose(out[2, :3], [100.5 * 2 * 0.002, 0, 2], rtol=1e-6)
    assert out[2, 6] == 0 and out[2, 7] == 1


def test_fundamental_errors_known_answers():
    """horizontal epipolar lines (pure x translation: F = [t]_x with t = (1, 0, 0)): the symmetric distance is the squared row difference."""
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float64)
    p1 = np.array([[10, 20], [300, 200.5], [5, 5]], np.float32)
    p2 = np.array([[50, 23], [10, 200.5], [7, 4]], np.float32)
    assert fo.fundamental_errors(F, p1, p2).tolist() == [9.0, 0.0, 1.0]
    # scale invariance of the distance and the float32 result type
    e = fo.fundamental_errors(3.5 * F, p1, p2)
    assert e.dtype == np.float32 and np.allclose(e, [9.0, 0.0, 1.0], rtol=1e-6)
    # a general F: the error is the larger of the two point-to-line squared distances, each computed independently here
    rng = np.random.default_rng(4)
    F = rng.normal(size=(3, 3))
    p1, p2 = rng.uniform(0, 640, (50, 2)).astype(np.float32), rng.uniform(0, 480, (50, 2)).astype(np.float32)
    h1, h2 = np.c_[p1, np.ones(50)].astype(np.float64), np.c_[p2, np.ones(50)].astype(np.float64)
    l2, l1 = h1 @ F.T, h2 @ F            # epipolar lines in image 2 of the points of image 1, and the other way round
    d2 = (l2 * h2).sum(1) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
    d1 = (l1 * h1).sum(1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    np.testing.assert_allclose(fo.fundamental_errors(F, p1, p2), np.maximum(d1, d2), rtol=1e-5)


def test_pnp_errors_known_answers():
    obj = np.array([[0, 0, 2], [1, 0, 2], [0, 1, 4], [0, 0, 0]], np.float32)
    R, t = np.eye(3), np.zeros(3)
    img = np.array([[320, 240], [570, 240], [320, 365], [321, 243]], np.float32)   # fx = fy = 500, principal point (320, 240)
    assert fo.pnp_errors(R, t, obj, img, 500.0, 500.0, 320.0, 240.0).tolist() == [0.0, 0.0, 0.0, 10.0]   # the last point has Z = 0: projects to the principal point
    e = fo.pnp_errors(R, np.array([0.01, 0, 0]), obj[:3], img[:3], 500.0, 500.0, 320.0, 240.0)
    np.testing.assert_allclose(e, [2.5 ** 2, 2.5 ** 2, 1.25 ** 2], rtol=1e-4)   # a 1 cm shift at 2 m / 4 m depth: 2.5 / 1.25 px
