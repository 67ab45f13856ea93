"""GPU: the point arithmetic of Tracking::GetSceneFlowObj (amos_flow_*_device) against the numpy
restatement in oracle/flow_oracle.py: states and validity flags exact, doubles / floats bit for bit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _scene(synth, rng, n):
    last, cur = synth.frame(31, 4), synth.frame(31, 5)
    pre = np.stack([rng.uniform(-3, 645, n), rng.uniform(-3, 485, n)], 1).astype(np.float32)
    nxt = (pre + np.array([2.0, 1.0], np.float32) + rng.normal(0, 0.7, (n, 2))).astype(np.float32)
    nxt[::17] += rng.uniform(-40, 40, (len(nxt[::17]), 2)).astype(np.float32)  # lost tracks: large SAD
    pre[:8] = [[4.99, 50], [5.0, 50], [634.99, 50], [635.0, 50], [50, 4.5], [50, 5.2], [50, 474.9], [50, 475.0]]  # the 5-px border
    state = (rng.random(n) < 0.9).astype(np.uint8)
    return last, cur, pre, nxt, state


def test_flow_check_vs_numpy(gpu_lib, synth):
    import torch
    import flow_oracle as fo
    rng = np.random.default_rng(1)
    last, cur, pre, nxt, state = _scene(synth, rng, 3000)
    d = [torch.from_numpy(a).cuda() for a in (last, cur, pre, nxt, state)]
    out = torch.full((len(pre),), 7, dtype=torch.uint8, device="cuda")
    gpu_lib.flow_check(torch.cuda.current_stream().cuda_stream, d[0].data_ptr(), 640, d[1].data_ptr(), 640, 640, 480, d[2].data_ptr(), d[3].data_ptr(),
                       d[4].data_ptr(), len(pre), out.data_ptr())
    torch.cuda.synchronize()
    want = fo.flow_check(last, cur, pre, nxt, state)
    got = out.cpu().numpy()
    assert np.array_equal(got, want)
    assert 0.3 < want.mean() < 0.95 and (want[:8] == [0, state[1], state[2], 0, 0, state[5], state[6], 0]).all()


def test_epipolar_distance_vs_numpy(gpu_lib, synth):
    import torch
    import flow_oracle as fo
    rng = np.random.default_rng(2)
    _, _, pre, nxt, state = _scene(synth, rng, 4096)
    # the fundamental matrix of a pure image translation by (2, 1) px, [t]_x, slightly perturbed: epipolar lines run along the motion
    F = np.array([[1e-9, 2e-9, 1.0], [-3e-9, 1e-9, -2.0], [-1.0, 2.0, 1e-4]], np.float64)
    d_F, d_pre, d_nxt, d_state = (torch.from_numpy(a).cuda() for a in (F, pre, nxt, state))
    dd = torch.zeros(len(pre), dtype=torch.float64, device="cuda")
    gpu_lib.flow_epipolar(torch.cuda.current_stream().cuda_stream, d_F.data_ptr(), d_pre.data_ptr(), d_nxt.data_ptr(), d_state.data_ptr(), len(pre), dd.data_ptr())
    torch.cuda.synchronize()
    want = fo.epipolar(F, pre, nxt, state)
    got = dd.cpu().numpy()
    assert got.tobytes() == want.tobytes()
    assert ((want >= 0) & (want <= 0.5)).any() and (want > 1).any() and (want == -1).sum() == (state == 0).sum()


def test_scene_flow_vs_numpy(gpu_lib, synth):
    import torch
    import flow_oracle as fo
    rng = np.random.default_rng(3)
    n = 2500
    pre = np.stack([rng.uniform(0, 639.9, n), rng.uniform(0, 479.9, n)], 1).astype(np.float32)
    cur = np.clip(pre + rng.normal(0, 2, (n, 2)), 0, [639.9, 479.9]).astype(np.float32)
    yy, xx = np.mgrid[0:480, 0:640]
    d_last = (1.5 + 0.5 * np.sin(xx / 90.0) + 0.3 * np.cos(yy / 60.0)).astype(np.float32)
    d_cur = (d_last + 0.02).astype(np.float32)
    d_last[rng.random(d_last.shape) < 0.1] = 0  # invalid depth
    th = 0.03
    Rlw = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    Tlw = np.concatenate([Rlw, np.array([[0.1], [-0.05], [0.2]], np.float32)], 1)
    Rwc = Rlw.T.copy()
    Ow = np.array([-0.08, 0.04, -0.25], np.float32)
    cam = gpu_lib.SceneFlowCamera(320.1, 247.6, 1 / 535.4, 1 / 539.2)
    for i, v in enumerate(Tlw.reshape(-1)):
        cam.Tlw[i] = float(v)
    for i, v in enumerate(Rwc.reshape(-1)):
        cam.Rwc[i] = float(v)
    for i, v in enumerate(Ow):
        cam.Ow[i] = float(v)
    t = [torch.from_numpy(a).cuda() for a in (d_last, d_cur, pre, cur)]
    out = torch.zeros((n, 8), dtype=torch.float32, device="cuda")
    gpu_lib.flow_scene_flow(torch.cuda.current_stream().cuda_stream, t[0].data_ptr(), 640, t[1].data_ptr(), 640, t[2].data_ptr(), t[3].data_ptr(), n, cam, out.data_ptr())
    torch.cuda.synchronize()
    want = fo.scene_flow(d_last, d_cur, pre, cur, np.float32(cam.cx), np.float32(cam.cy), np.float32(cam.invfx), np.float32(cam.invfy), Tlw, Rwc, Ow)
    got = out.cpu().numpy()
    bad = np.nonzero((got != want).any(1))[0]
    assert len(bad) == 0, (len(bad), np.nonzero((got != want).any(0))[0].tolist(), got[bad[:2]].tolist(), want[bad[:2]].tolist())
    assert got.tobytes() == want.tobytes()
    assert 0.8 < want[:, 7].mean() < 0.95 and want[want[:, 7] > 0, 6].max() > 0.01


def test_ransac_hypothesis_scorers_vs_numpy(gpu_lib, synth):
    """amos_flow_fundamental_score_device / amos_flow_pnp_score_device (the device side of GetSceneFlowObj's three RANSACs: every
    correspondence's error under every hypothesis, inlier test, inlier count) against oracle/flow_oracle.py: errors bit for bit, masks and
    counts exact, for 64 hypotheses x 1 000 correspondences incl. exact ones (error 0), gross outliers and a point at depth zero."""
    import torch
    import flow_oracle as fo
    rng = np.random.default_rng(6)
    st = torch.cuda.current_stream().cuda_stream
    n, nh = 1000, 64
    # two views of random 3-D points: a small rotation + translation; correspondences with 0.3 px noise, 10 % gross outliers
    K = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1]])
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.5, 6, n)]

    def rot(rx, ry, rz):
        cx_, sx, cy_, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        return np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]]) @ np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    R0, t0 = rot(0.01, -0.02, 0.005), np.array([0.05, 0.01, -0.02])
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    X2 = X @ R0.T + t0
    x2 = (K @ X2.T).T
    x2 = x2[:, :2] / x2[:, 2:]
    p1 = np.ascontiguousarray(x1, np.float32)   # (x1, x2 are transposed views: the device wants [n][2] rows)
    p2 = np.ascontiguousarray(x2 + rng.normal(0, 0.3, x2.shape), np.float32)
    p2[::10] += rng.uniform(-60, 60, (len(p2[::10]), 2)).astype(np.float32)
    tx = np.array([[0, -t0[2], t0[1]], [t0[2], 0, -t0[0]], [-t0[1], t0[0], 0]])
    Ftrue = np.linalg.inv(K).T @ tx @ R0 @ np.linalg.inv(K)
    Fs = np.stack([Ftrue * (1 + 0.0) if h == 0 else Ftrue + rng.normal(0, 1e-7 * (h % 5), (3, 3)) for h in range(nh)]).astype(np.float64)
    d_F, d_p1, d_p2 = torch.from_numpy(Fs.reshape(nh, 9).copy()).cuda(), torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda()
    err = torch.full((nh, n), float("nan"), device="cuda")
    cnt = torch.full((nh,), -1, dtype=torch.int32, device="cuda")
    msk = torch.full((nh, n), 7, dtype=torch.uint8, device="cuda")
    gpu_lib.flow_fundamental_score(st, d_F.data_ptr(), nh, d_p1.data_ptr(), d_p2.data_ptr(), n, 0.1, err.data_ptr(), cnt.data_ptr(), msk.data_ptr())
    torch.cuda.synchronize()
    thr2 = np.float32(0.1 * 0.1)
    for h in range(nh):
        want = fo.fundamental_errors(Fs[h], p1, p2)
        assert err[h].cpu().numpy().tobytes() == want.tobytes(), h
        assert np.array_equal(msk[h].cpu().numpy(), (want <= thr2).astype(np.uint8)) and int(cnt[h]) == int((want <= thr2).sum()), h
    assert 0 < int(cnt[0]) < n and int(cnt.max()) >= int(cnt[0]) - 50
    # counts only (no error / mask buffers)
    cnt2 = torch.full((nh,), -1, dtype=torch.int32, device="cuda")
    gpu_lib.flow_fundamental_score(st, d_F.data_ptr(), nh, d_p1.data_ptr(), d_p2.data_ptr(), n, 0.1, None, cnt2.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal(cnt, cnt2)

    # poses: the true one, perturbed ones, one that puts a point at depth exactly zero
    obj = X.astype(np.float32)
    obj[5] = [0.3, 0.2, 0.0]
    img = p2.copy()
    Rts = np.zeros((nh, 12))
    for h in range(nh):
        Rh = rot(0.01 + 1e-4 * (h % 7), -0.02, 0.005 + 2e-4 * (h % 3))
        th = t0 + (0 if h == 0 else rng.normal(0, 1e-3, 3))
        Rts[h, :9], Rts[h, 9:] = Rh.reshape(9), th
    Rts[1, :9], Rts[1, 9:] = np.eye(3).reshape(9), 0.0   # identity: obj[5] has Z = 0 exactly
    d_Rt, d_obj, d_img = torch.from_numpy(Rts.copy()).cuda(), torch.from_numpy(obj).cuda(), torch.from_numpy(img).cuda()
    err.fill_(float("nan")); cnt.fill_(-1); msk.fill_(7)
    gpu_lib.flow_pnp_score(st, d_Rt.data_ptr(), nh, d_obj.data_ptr(), d_img.data_ptr(), n, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.4, err.data_ptr(), cnt.data_ptr(),
                           msk.data_ptr())
    torch.cuda.synchronize()
    thr2 = np.float32(0.4 * 0.4)
    for h in range(nh):
        want = fo.pnp_errors(Rts[h, :9], Rts[h, 9:], obj, img, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        assert err[h].cpu().numpy().tobytes() == want.tobytes(), h
        assert np.array_equal(msk[h].cpu().numpy(), (want <= thr2).astype(np.uint8)) and int(cnt[h]) == int((want <= thr2).sum()), h
    assert int(cnt[0]) > 100 and int(cnt[1]) < int(cnt[0])
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.flow_pnp_score(st, d_Rt.data_ptr(), nh, d_obj.data_ptr(), d_img.data_ptr(), n, K[0, 0], K[1, 1], K[0, 2], K[1, 2], -1.0, None, cnt.data_ptr(), None)
