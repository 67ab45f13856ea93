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
