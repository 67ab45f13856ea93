"""The corner source of Tracking::GetSceneFlowObj (src/Tracking.cc:894-895): cv::goodFeaturesToTrack (Harris) + cv::cornerSubPix.
CPU: the oracle (oracle/corner_oracle.c) against an independent numpy restatement of the Harris response, the defining properties of
the selection, hand cases of the sub-pixel refinement.  GPU: amos_corners_* bit for bit against the oracle, and the chain into the
pyramidal LK tracker with everything resident.  OpenCV-derived: parity unpinned (the oracle's header says what was restated)."""
import numpy as np
import pytest

f32 = np.float32


def _refl(a, n):
    a = np.abs(a)
    return np.where(a >= n, 2 * n - 2 - a, a)


def _harris_numpy(img, k=0.04):
    """corner.cpp's cornerHarris for blockSize 3 / aperture 3 on an 8-bit image, written with array operations: Sobel with the scale on the
    smoothing taps (float), products (float), 3 x 3 box sums (double, one rounding), response in float; REFLECT_101 at both filters."""
    h, w = img.shape
    P = img[np.ix_(_refl(np.arange(-1, h + 1), h), _refl(np.arange(-1, w + 1), w))].astype(np.int32)
    s = 1.0 / (4 * 3 * 255)
    k1, k2 = f32(s), f32(2 * s)
    a = (P[:, 2:] - P[:, :-2]).astype(f32)
    dx = ((a[:-2] + a[2:]) * k1 + a[1:-1] * k2).astype(f32)
    b = (P[:, 1:-1].astype(f32) * k2 + (P[:, :-2] + P[:, 2:]).astype(f32) * k1).astype(f32)
    dy = (b[2:] - b[:-2]).astype(f32)

    def box(v):
        V = v[np.ix_(_refl(np.arange(-1, h + 1), h), _refl(np.arange(-1, w + 1), w))].astype(np.float64)
        rows = V[:, :-2] + V[:, 1:-1] + V[:, 2:]
        return (rows[:-2] + rows[1:-1] + rows[2:]).astype(f32)
    A, B, Cc = box((dx * dx).astype(f32)), box((dx * dy).astype(f32)), box((dy * dy).astype(f32))
    tr = (A + Cc).astype(f32)
    return ((A * Cc).astype(f32) - (B * B).astype(f32)).astype(f32) - ((f32(k) * tr).astype(f32) * tr).astype(f32)


def _corner_image(cx, cy, w=160, h=120, lo=30, hi=200):
    """a bright quadrant whose corner sits at the sub-pixel position (cx, cy): 4 x 4 supersampled"""
    big = np.full((h * 4, w * 4), lo, f32)
    big[int(round(cy * 4 + 2)):, int(round(cx * 4 + 2)):] = hi
    return big.reshape(h, 4, w, 4).mean((1, 3)).astype(np.uint8)


def test_harris_response_against_numpy(ob, synth):
    for img in (synth.frame(5, 2), synth.frame(6, 1, 120, 200), np.zeros((40, 50), np.uint8)):
        assert np.array_equal(ob.corner_harris(img), _harris_numpy(img))


def test_good_features_selection_rules(ob, synth):
    img = synth.frame(5, 2)
    xy, R = ob.good_features_to_track(img, with_response=True)
    assert 100 < len(xy) <= 1000 and xy.dtype == np.float32
    xi, yi = xy[:, 0].astype(int), xy[:, 1].astype(int)
    assert (xi >= 1).all() and (xi <= 638).all() and (yi >= 1).all() and (yi <= 478).all()          # the border row / column never holds a corner
    vals = R[yi, xi]
    thr = f32(np.float64(R.max()) * 0.01)
    assert (vals > thr).all() and (np.diff(vals) <= 0).all()                                        # above the quality threshold, strongest first
    Rt = np.where(R > thr, R, 0)
    for x, y in zip(xi[:200], yi[:200]):
        assert Rt[y, x] == Rt[y - 1:y + 2, x - 1:x + 2].max()                                        # 3 x 3 local maxima of the thresholded response
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, 1e9)
    assert d2.min() >= 64                                                                           # no two corners closer than minDistance
    # greedy in order: every local maximum that was NOT taken lies within minDistance of a stronger taken one (checked on a sample)
    cand = np.argwhere((Rt[1:-1, 1:-1] != 0) & (Rt[1:-1, 1:-1] == np.max([Rt[1 + j:479 + j, 1 + i:639 + i] for j in (-1, 0, 1) for i in (-1, 0, 1)], 0))) + 1
    taken = {(int(x), int(y)) for x, y in zip(xi, yi)}
    rng = np.random.default_rng(0)
    for y, x in cand[rng.permutation(len(cand))[:300]]:
        if (x, y) in taken:
            continue
        near = ((xy[:, 0] - x) ** 2 + (xy[:, 1] - y) ** 2) < 64
        assert near.any() and (vals[near] >= R[y, x]).any()
    # maxCorners cuts the same list; a larger minimum distance gives fewer corners
    assert np.array_equal(ob.good_features_to_track(img, max_corners=50), xy[:50])
    assert len(ob.good_features_to_track(img, min_distance=20.0)) < len(xy)
    assert len(ob.good_features_to_track(np.full((60, 80), 77, np.uint8))) == 0                      # a flat image has no corners


def test_equal_responses_take_the_later_pixel_first(ob):
    """two identical isolated blobs give exactly equal responses; the library sorts pointers descending on ties"""
    img = np.zeros((60, 100), np.uint8)
    img[20:24, 20:24] = 200
    img[20:24, 70:74] = 200
    xy = ob.good_features_to_track(img, min_distance=3.0)
    R = ob.corner_harris(img)
    v = R[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    ties = np.nonzero(np.diff(v) == 0)[0]
    assert len(ties) > 0
    for t in ties:
        assert xy[t, 1] * 100 + xy[t, 0] > xy[t + 1, 1] * 100 + xy[t + 1, 0]


def test_subpix_finds_a_known_corner(ob):
    for cx, cy in ((80.3, 60.6), (70.0, 50.5), (90.75, 40.25)):
        img = _corner_image(cx, cy)
        start = np.array([[round(cx) + 1, round(cy) - 1]], f32)
        got = ob.corner_subpix(img, start)
        assert abs(got[0, 0] - cx) < 0.15 and abs(got[0, 1] - cy) < 0.15, (cx, cy, got)
    # a point on a flat region has a singular system: it stays where it was; a point that would leave the window is put back
    flat = np.full((120, 160), 90, np.uint8)
    assert np.array_equal(ob.corner_subpix(flat, np.array([[40.0, 50.0]], f32)), np.array([[40.0, 50.0]], f32))
    m = ob.corner_subpix_mask(10)
    assert m.shape == (21, 21) and m[10, 10] == 1.0 and np.allclose(m[0, 10], np.exp(-1.0), rtol=1e-6) and np.array_equal(m, m.T)


def test_subpix_near_the_border_uses_replicated_pixels(ob, synth):
    """patches that stick out of the image on every side (getRectSubPix's border branch): results are finite and inside the image"""
    img = synth.frame(7, 3, 100, 140)
    pts = np.array([[1, 1], [138, 1], [1, 98], [138, 98], [70, 1], [1, 50], [138.5, 50.25], [69.5, 98.75], [0.25, 0.25]], f32)
    got = ob.corner_subpix(img, pts)
    assert np.isfinite(got).all() and (np.abs(got - pts) <= 10).all()


@pytest.fixture(scope="module")
def torch_gpu(gpu_lib):
    import torch
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["synth", "ref_frame", "small", "blobs"])
def test_gpu_corners_equal_the_oracle(gpu_lib, ob, synth, torch_gpu, case):
    torch = torch_gpu
    if case == "synth":
        img, kw = synth.frame(5, 2), {}
    elif case == "ref_frame":
        import os
        from PIL import Image
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        img = np.array(Image.open(os.path.join(root, "tests", "golden", "ref_data", "1341846313.553992.png")).convert("L"))
        kw = {}
    elif case == "small":
        img, kw = synth.frame(6, 1, 97, 131), dict(max_corners=40, quality=0.02, min_distance=5.0)
    else:
        img = np.zeros((60, 100), np.uint8)
        img[20:24, 20:24] = 200
        img[20:24, 70:74] = 200
        kw = dict(min_distance=3.0)
    img = np.ascontiguousarray(img)
    h, w = img.shape
    want_xy, want_R = ob.good_features_to_track(img, with_response=True, **kw)
    det = gpu_lib.CornerDetector(max_width=max(w, 640), max_height=max(h, 480))
    st = torch.cuda.ExternalStream(det.stream)
    d_img = torch.from_numpy(img).cuda()
    d_xy = torch.full((2000, 2), -1.0, dtype=torch.float32, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_R = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    det.good_features_device(d_img.data_ptr(), w, w, h, d_xy.data_ptr(), 2000, d_n.data_ptr(), response_ptr=d_R.data_ptr(), **kw)
    st.synchronize()
    assert det.candidate_count() >= len(want_xy)
    assert np.array_equal(d_R.cpu().numpy(), want_R), "Harris response"
    n = int(d_n.item())
    assert n == len(want_xy) and np.array_equal(d_xy[:n].cpu().numpy(), want_xy), "corner list"
    # refinement in place, count read on the device
    want_sub = ob.corner_subpix(img, want_xy)
    det.subpix_device(d_img.data_ptr(), w, w, h, d_xy.data_ptr(), count_ptr=d_n.data_ptr(), n=2000)
    st.synchronize()
    got_sub = d_xy[:n].cpu().numpy()
    assert got_sub.tobytes() == want_sub.tobytes(), "refined positions"
    assert (d_xy[n:] == -1).all()
    if case in ("synth", "ref_frame"):
        assert n > 100
    det.close()


@pytest.mark.gpu
def test_gpu_subpix_on_given_points_incl_borders(gpu_lib, ob, synth, torch_gpu):
    torch = torch_gpu
    img = synth.frame(7, 3, 100, 140)
    rng = np.random.default_rng(3)
    pts = np.concatenate([np.array([[1, 1], [138, 1], [1, 98], [138, 98], [70, 1], [1, 50], [138.5, 50.25], [69.5, 98.75], [0.25, 0.25]], f32),
                          (rng.random((300, 2)) * [139, 99]).astype(f32)])
    det = gpu_lib.CornerDetector(max_width=640, max_height=480)
    st = torch.cuda.ExternalStream(det.stream)
    d_img = torch.from_numpy(img).cuda()
    for win, iters, eps in ((10, 20, 0.03), (5, 40, 0.001), (3, 1, 0.0)):
        d_xy = torch.from_numpy(pts).cuda()
        torch.cuda.synchronize()
        det.subpix_device(d_img.data_ptr(), 140, 140, 100, d_xy.data_ptr(), n=len(pts), win=win, max_count=iters, epsilon=eps)
        st.synchronize()
        assert d_xy.cpu().numpy().tobytes() == ob.corner_subpix(img, pts, win, iters, eps).tobytes(), win
    det.close()


@pytest.mark.gpu
def test_gpu_corners_feed_the_lk_tracker_resident(gpu_lib, ob, synth, torch_gpu):
    """Tracking.cc:894-896 as one resident chain: corners of the last frame -> sub-pixel -> pyramidal LK into the current frame; nothing
    but the final arrays crosses the host link.  Expected: the oracle's three functions chained on the host."""
    torch = torch_gpu
    f0, f1 = synth.frame(9, 10), synth.frame(9, 11)
    want_xy = ob.corner_subpix(f0, ob.good_features_to_track(f0))
    want_next, want_st, want_err, _top = ob.lk_track(f0, f1, want_xy)
    det = gpu_lib.CornerDetector()
    lk = gpu_lib.LkTracker(640, 480, stream=det.stream)
    st = torch.cuda.ExternalStream(det.stream)
    d0, d1 = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
    d_xy = torch.zeros((1000, 2), dtype=torch.float32, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    det.good_features_device(d0.data_ptr(), 640, 640, 480, d_xy.data_ptr(), 1000, d_n.data_ptr())
    det.subpix_device(d0.data_ptr(), 640, 640, 480, d_xy.data_ptr(), count_ptr=d_n.data_ptr(), n=1000)
    st.synchronize()
    n = int(d_n.item())
    assert n == len(want_xy) and d_xy[:n].cpu().numpy().tobytes() == want_xy.tobytes()
    d_next = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_err = torch.zeros(n, dtype=torch.float32, device="cuda")
    lk.track_device(d0.data_ptr(), 640, d1.data_ptr(), 640, d_xy.data_ptr(), n, d_next.data_ptr(), d_st.data_ptr(), d_err.data_ptr())
    st.synchronize()
    assert np.array_equal(d_st.cpu().numpy(), want_st) and d_next.cpu().numpy().tobytes() == want_next.tobytes()
    assert int(want_st.sum()) > 0.8 * n   # the stream moves by (2, 1) px per frame: nearly everything is tracked
    moved = (want_next - want_xy)[want_st > 0]
    assert abs(np.median(moved[:, 0]) + 2) < 0.3 and abs(np.median(moved[:, 1]) + 1) < 0.3
