"""cv::calcOpticalFlowPyrLK as Tracking::GetSceneFlowObj calls it (src/Tracking.cc:896), restated (PARITY UNPINNED, see
oracle/lk_oracle.c): the oracle's pyramid and derivatives against independent numpy code, its tracker on known motions (CPU);
the HIP kernels against the oracle, bit for bit (GPU)."""
import numpy as np
import pytest


def _np_pyr_down(img):
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)

    def refl(i, n):
        i = np.abs(i)
        return np.where(i >= n, 2 * n - 2 - i, i)

    ys = refl(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)      # [dh, 5]
    xs = refl(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)      # [dw, 5]
    a = img.astype(np.int64)
    hs = (a[:, xs] * k).sum(2)                                                # [h, dw]
    out = (hs[ys, :] * k[None, :, None]).sum(1)                               # [dh, dw]
    return ((out + 128) >> 8).astype(np.uint8)


def _np_scharr(img):
    a = np.pad(img.astype(np.int64), 1, mode="reflect")
    t0 = (a[:-2] + a[2:]) * 3 + a[1:-1] * 10          # vertical [3 10 3], columns padded
    t1 = a[2:] - a[:-2]                                # vertical [-1 0 1]
    dx = t0[:, 2:] - t0[:, :-2]
    dy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return np.stack([dx, dy], -1).astype(np.int16)


def test_pyramid_and_derivatives_vs_numpy(ob, synth):
    img = synth.frame(40, 3)
    level = img
    for l in range(5):
        got, deriv, top = ob.lk_pyramid_level(img, l)
        assert top == 4  # 640 x 480, 22 x 22 window: the 20 x 15 level is not built
        assert np.array_equal(got, level), l
        assert np.array_equal(deriv, _np_scharr(level)), l
        level = _np_pyr_down(level)
    odd = synth.frame(41, 0, 241, 323)  # odd sizes: (w + 1) / 2
    got, deriv, top = ob.lk_pyramid_level(odd, 2)
    want = _np_pyr_down(_np_pyr_down(odd))
    assert top == 3 and got.shape == (61, 81) and np.array_equal(got, want) and np.array_equal(deriv, _np_scharr(want))


def test_tracker_recovers_known_translations(ob, synth):
    rng = np.random.default_rng(0)
    a = synth.frame(31, 4)
    pts = np.stack([rng.uniform(40, 600, 400), rng.uniform(40, 440, 400)], 1).astype(np.float32)
    for k, shift in ((5, (-2.0, -1.0)), (7, (-6.0, -3.0)), (12, (-16.0, -8.0))):  # frame k of a stream = the scene moved by (2, 1) per frame
        b = synth.frame(31, k)
        out, st, err, top = ob.lk_track(a, b, pts)
        assert top == 4 and st.mean() > 0.9
        flow = (out - pts)[st > 0]
        assert np.abs(np.median(flow, 0) - np.array(shift)).max() < 0.05
        assert (np.abs(flow - np.array(shift)).max(1) < 0.5).mean() > 0.9
    # identical frames: zero flow, zero residual
    out, st, err, _ = ob.lk_track(a, a, pts)
    assert st.mean() > 0.9 and np.abs(out - pts)[st > 0].max() < 1e-3 and err[st > 0].max() < 1e-3  # (windows without texture fail the eigenvalue test; float position round-off leaves residuals of a few 1e-4)
    # a flat image: the minimum-eigenvalue test rejects every point at level 0
    flat = np.full((480, 640), 77, np.uint8)
    out, st, err, _ = ob.lk_track(flat, flat, pts)
    assert not st.any()
    # points outside the padded image are rejected at level 0, the others are untouched by them
    far = np.array([[-40.0, 100.0], [700.0, 100.0], [320.0, 240.0]], np.float32)
    out, st, _, _ = ob.lk_track(a, a, far)
    assert st.tolist() == [0, 0, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,k", [(640, 480, 5), (640, 480, 9), (323, 241, 6)])
def test_gpu_lk_vs_oracle(gpu_lib, ob, synth, w, h, k):
    import torch
    rng = np.random.default_rng(w + k)
    a, b = synth.frame(33, 4, h, w), synth.frame(33, k, h, w)
    n = 1000
    pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
    pts[:50] = np.stack([rng.uniform(0, w, 50), rng.uniform(0, h, 50)], 1).round()  # integer positions: zero fractional weights
    want, wst, werr, top = ob.lk_track(a, b, pts)
    lk = gpu_lib.LkTracker(w, h)
    assert lk.levels == top
    d_a, d_b, d_p = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(pts).cuda()
    d_out = torch.full((n, 2), -9.0, dtype=torch.float32, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    d_err = torch.full((n,), -9.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    lk.track_device(d_a.data_ptr(), w, d_b.data_ptr(), w, d_p.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), d_err.data_ptr())
    torch.cuda.ExternalStream(lk.stream).synchronize()
    got, gst, gerr = d_out.cpu().numpy(), d_st.cpu().numpy(), d_err.cpu().numpy()
    assert np.array_equal(gst, wst)
    bad = np.nonzero((got != want).any(1))[0]
    assert len(bad) == 0, (len(bad), bad[:3], got[bad[:3]], want[bad[:3]])
    assert got.tobytes() == want.tobytes() and gerr.tobytes() == werr.tobytes()
    assert 0.5 < wst.mean() <= 1.0
