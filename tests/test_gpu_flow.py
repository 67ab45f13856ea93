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
