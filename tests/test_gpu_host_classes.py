"""GPU: the C++ drop-in classes (ORB_SLAM2::ORBextractor / ORBmatcher in amos-slam_amd/host) driven
through their reference-shaped methods, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hb(gpu_lib):
    import host_binding
    host_binding.host()
    return host_binding


def _same(a, b, what):
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    assert a.tobytes() == b.tobytes(), f"{what} differs"


def test_extractor_call_operator(hb, ob, synth):
    """ORBextractor::operator()(image, mask, keypoints, descriptors) + mvImagePyramid ROI."""
    img = synth.frame(13, 4)
    kps, desc, pyr = hb.host_extract(img, pyr_level=2)
    orc = ob.Oracle()
    ko, do = orc.extract(img)
    _same(kps, ko, "keypoints")
    _same(desc, do, "descriptors")
    _same(pyr, orc.level_image(2, padded=True), "mvImagePyramid[2] incl. border")


def test_extractor_other_params(hb, ob, synth):
    img = synth.frame(14, 0, 360, 480)
    kps, desc, _ = hb.host_extract(img, nf=600, sf=1.3, nl=5, ini=25, mn=9)
    ko, do = ob.Oracle(600, 1.3, 5, 25, 9).extract(img)
    _same(kps, ko, "keypoints")
    _same(desc, do, "descriptors")


@pytest.mark.parametrize("with_labels", [False, True])
def test_amos_rgbd_flow(hb, ob, synth, with_labels):
    """operator()(3-arg) -> MovingKeyPoints -> ProcessDesp, as Frame.cc:480-496,633 calls them."""
    img, mask = synth.frame(15, 2), synth.person_mask(15, 2)
    labels = ids = rm = None
    if with_labels:
        rng = np.random.default_rng(3)
        labels = np.kron(rng.integers(1, 16, (15, 20)), np.ones((32, 32))).astype(np.float64)
        ids = rng.permutation(15).astype(np.int32)
        rm = np.zeros(15, np.int32)
        rm[[1, 5, 9]] = 1
    removed, kps, desc, lists, counts = hb.host_amos_flow(img, mask, labels, ids, rm)
    orc = ob.Oracle()
    orc.detect(img)
    ro = orc.gate(mask, labels, ids, rm)
    ko, do = orc.describe()
    _same(removed, ro, "DynaPt")
    _same(kps, ko, "mvKeys")
    _same(desc, do, "mDescriptors")
    # the caller's per-level vectors come back rescaled in place (ORBextractor.cc:1804-1813)
    assert [int(c) for c in counts] == [len(orc.level_keypoints(l)) for l in range(8)]
    _same(lists, ko, "mvKeysTemp after ProcessDesp")
    assert len(removed) > 0


def test_descriptor_distance_static(hb, ob):
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        assert hb.host_descriptor_distance(a, b) == ob.descriptor_distance(a, b)
    assert hb.host_descriptor_distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def _two_frames(ob, synth, stream=16):
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(stream, 20))
    k1, d1 = orc.extract(synth.frame(stream, 21))
    return k0, d0, k1, d1, orc.tables()["scale"]


def test_features_in_area(hb, ob, synth):
    k0, d0, _, _, _ = _two_frames(ob, synth)
    view, keep = hb.frame_view(k0, d0)
    rng = np.random.default_rng(1)
    for _ in range(200):
        x, y = float(rng.uniform(-30, 670)), float(rng.uniform(-30, 510))
        r = float(rng.uniform(1, 60))
        lo, hi = (int(rng.integers(-1, 6)), int(rng.integers(-1, 8)))
        got = hb.host_features_in_area(view, x, y, r, lo, hi)
        want = hb.oracle_features_in_area(view, x, y, r, lo, hi)
        assert np.array_equal(got, want)
        if lo < 0 and hi < 0:  # brute force cross-check of the grid
            inside = np.nonzero((np.abs(k0["x"] - np.float32(x)) < np.float32(r)) & (np.abs(k0["y"] - np.float32(y)) < np.float32(r)))[0]
            assert set(inside.tolist()) >= set(got.tolist())


@pytest.mark.parametrize("forward,backward", [(0, 0), (1, 0), (0, 1)])
def test_search_by_projection_frame(hb, ob, synth, forward, backward):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) incl. the greedy
    already-matched skip, right-coordinate gate and rotation histogram."""
    k0, d0, k1, d1, sf = _two_frames(ob, synth)
    rng = np.random.default_rng(2)
    ur = np.where(rng.random(len(k1)) < 0.7, k1["x"] - rng.uniform(5, 40, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur)
    q = np.zeros(len(k0), hb.PROJ_QUERY)
    q["u"] = k0["x"] - 2 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["v"] = k0["y"] - 1 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["invz"] = rng.uniform(0.2, 2.0, len(k0)).astype(np.float32)
    q["octave"], q["angle"], q["desc"] = k0["octave"], k0["angle"], d0
    q["has_obs"] = rng.random(len(k0)) < 0.6
    match0 = np.full(len(k1), -1, np.int32)
    for th in (7.0, 15.0):
        rh, mh = hb.search_frame("host", view, q, match0, sf, 40.0, th, forward, backward)
        ro, mo = hb.search_frame("oracle", view, q, match0, sf, 40.0, th, forward, backward)
        assert rh == ro and np.array_equal(mh, mo)
        assert rh > 50
    rh, mh = hb.search_frame("host", view, q, match0, sf, 40.0, 15.0, forward, backward, check_ori=False)
    ro, mo = hb.search_frame("oracle", view, q, match0, sf, 40.0, 15.0, forward, backward, check_ori=False)
    assert rh == ro and np.array_equal(mh, mo)


def test_search_by_projection_points(hb, ob, synth):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th): best / second best with the same-level
    ratio rule (ORBmatcher.cc:163-167)."""
    k0, d0, k1, d1, sf = _two_frames(ob, synth, stream=17)
    rng = np.random.default_rng(4)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 20, -1).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur)
    q = np.zeros(len(k0), hb.MAP_QUERY)
    q["proj_x"], q["proj_y"] = k0["x"] - 2, k0["y"] - 1
    q["proj_xr"] = q["proj_x"] - 20 + rng.normal(0, 3, len(k0)).astype(np.float32)
    q["view_cos"] = rng.choice([0.9, 0.999], len(k0)).astype(np.float32)
    q["level"] = np.maximum(k0["octave"], 0)
    q["has_obs"] = rng.random(len(k0)) < 0.5
    q["desc"] = d0
    match0 = np.full(len(k1), -1, np.int32)
    obs0 = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    for th in (1.0, 3.0, 5.0):
        rh, mh, oh = hb.search_points("host", view, q, match0, obs0, sf, th)
        ro, mo, oo = hb.search_points("oracle", view, q, match0, obs0, sf, th)
        assert rh == ro and np.array_equal(mh, mo) and np.array_equal(oh, oo)
    assert rh > 50


def test_search_for_initialization(hb, ob, synth):
    k0, d0, k1, d1, sf = _two_frames(ob, synth, stream=18)
    v1, keep1 = hb.frame_view(k0, d0)
    v2, keep2 = hb.frame_view(k1, d1)
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32)
    for window in (20, 100):
        rh, mh, ph = hb.search_init("host", v1, v2, prev, window)
        ro, mo, po = hb.search_init("oracle", v1, v2, prev, window)
        assert rh == ro and np.array_equal(mh, mo) and np.array_equal(ph, po)
    assert rh > 30
    assert (mh[k0["octave"] > 0] == -1).all()  # only level-0 features take part


def test_search_by_projection_keyframe(hb, ob, synth):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) -- relocalisation."""
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(21, 2))
    k1, d1 = orc.extract(synth.frame(21, 3))
    sf = orc.tables()["scale"]
    rng = np.random.default_rng(5)
    view, keep = hb.frame_view(k1, d1)
    q = np.zeros(len(k0), hb.KF_QUERY)
    q["u"] = k0["x"] - 2 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["v"] = k0["y"] - 1 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["level"], q["angle"], q["desc"] = k0["octave"], k0["angle"], d0
    m0 = np.where(rng.random(len(k1)) < 0.1, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    for th, orb_dist in ((10.0, 100), (3.0, 64)):
        n_h, m_h = hb.search_kf("host", view, q, m0, sf, th, orb_dist)
        n_o, m_o = hb.search_kf("oracle", view, q, m0, sf, th, orb_dist)
        assert n_h == n_o and np.array_equal(m_h, m_o)
    assert n_o > 50


def test_search_by_bow(hb, ob, synth):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) over feature vectors (stand-in vocabulary)."""
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(22, 2))
    k1, d1 = orc.extract(synth.frame(22, 3))
    rng = np.random.default_rng(6)

    def nodes(desc):
        ids = (desc[:, 0].astype(np.uint32) >> 2) * 3 + 5
        out = {}
        for i in rng.permutation(len(desc)):
            out.setdefault(int(ids[i]), []).append(int(i))
        return out
    n0, n1 = nodes(d0), nodes(d1)
    for drop in list(n1)[::6]:
        del n1[drop]
    has = (rng.random(len(k0)) < 0.85).astype(np.uint8)
    vkf, keep0 = hb.bow_view(k0, d0, n0, has)
    vf, keep1 = hb.bow_view(k1, d1, n1)
    for ratio, ori in ((0.7, True), (0.9, False)):
        n_h, m_h = hb.search_bow("host", vkf, vf, ratio, ori)
        n_o, m_o = hb.search_bow("oracle", vkf, vf, ratio, ori)
        assert n_h == n_o and np.array_equal(m_h, m_o)
    assert n_o > 30


def _bow_nodes(desc, rng):
    ids = (desc[:, 0].astype(np.uint32) >> 2) * 3 + 5
    out = {}
    for i in rng.permutation(len(desc)):
        out.setdefault(int(ids[i]), []).append(int(i))
    return out


def test_search_by_bow_keyframes(hb, ob, synth):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (loop closing)."""
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(23, 2))
    k1, d1 = orc.extract(synth.frame(23, 3))
    rng = np.random.default_rng(7)
    has0, has1 = (rng.random(len(k0)) < 0.85).astype(np.uint8), (rng.random(len(k1)) < 0.85).astype(np.uint8)
    v0, keep0 = hb.bow_view(k0, d0, _bow_nodes(d0, rng), has0)
    v1, keep1 = hb.bow_view(k1, d1, _bow_nodes(d1, rng), has1)
    for ratio, ori in ((0.75, True), (0.95, False)):
        n_h, m_h = hb.search_bow_kf("host", v0, v1, ratio, ori)
        n_o, m_o = hb.search_bow_kf("oracle", v0, v1, ratio, ori)
        assert n_h == n_o and np.array_equal(m_h, m_o)
    assert n_o > 30


@pytest.mark.parametrize("only_stereo", [0, 1])
def test_search_for_triangulation(hb, ob, synth, only_stereo):
    """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (local mapping)."""
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(24, 2))
    k1, d1 = orc.extract(synth.frame(24, 3))
    sf = orc.tables()["scale"]
    sg = orc.tables()["sigma2"]
    rng = np.random.default_rng(8 + only_stereo)
    has0, has1 = (rng.random(len(k0)) < 0.3).astype(np.uint8), (rng.random(len(k1)) < 0.3).astype(np.uint8)
    ur0 = np.where(rng.random(len(k0)) < 0.5, k0["x"] - 10, -1).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 10, -1).astype(np.float32)
    f12 = np.array([[0, 0, 1.0], [0, 0, -2.0], [-1.0, 2.0, 0]], np.float32)   # pure image translation by (2, 1)
    v0, keep0 = hb.bow_view(k0, d0, _bow_nodes(d0, rng), has0, ur0)
    v1, keep1 = hb.bow_view(k1, d1, _bow_nodes(d1, rng), has1, ur1)
    n_h, p_h = hb.search_triangulation("host", v0, v1, f12, 320.0, 200.0, sf, sg, only_stereo)
    n_o, p_o = hb.search_triangulation("oracle", v0, v1, f12, 320.0, 200.0, sf, sg, only_stereo)
    assert n_h == n_o and np.array_equal(p_h, p_o)
    assert n_o > 10


def _win_queries(hb, k_src, d_src, rng, dx=-2.0, dy=-1.0):
    q = np.zeros(len(k_src), hb.WINDOW_QUERY)
    q["u"] = k_src["x"] + dx + rng.normal(0, 1.5, len(k_src)).astype(np.float32)
    q["v"] = k_src["y"] + dy + rng.normal(0, 1.5, len(k_src)).astype(np.float32)
    q["ur"] = q["u"] - rng.uniform(5, 30, len(k_src)).astype(np.float32)
    q["level"] = np.minimum(k_src["octave"] + (rng.random(len(k_src)) < 0.3), 7)
    q["src"], q["desc"] = np.arange(len(k_src)), d_src
    return q


def test_fuse_and_sim3_searches(hb, ob, synth):
    """ORBmatcher::Fuse (both), SearchByProjection(pKF, Scw, ...) and SearchBySim3 (mapping / loop-closing threads)."""
    orc = ob.Oracle()
    k0, d0 = orc.extract(synth.frame(25, 2))
    k1, d1 = orc.extract(synth.frame(25, 3))
    t = orc.tables()
    sf, inv_s2 = t["scale"], t["inv_sigma2"]
    rng = np.random.default_rng(9)
    ur1 = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(5, 30, len(k1)), -1).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur1)
    q = _win_queries(hb, k0, d0, rng)
    for args in ((3.0, inv_s2), (4.0, None)):
        n_h, b_h = hb.fuse("host", view, q, sf, *args)
        n_o, b_o = hb.fuse("oracle", view, q, sf, *args)
        assert n_h == n_o and np.array_equal(b_h, b_o) and n_o > 50
    m0 = np.where(rng.random(len(k1)) < 0.2, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    n_h, m_h = hb.search_projection_sim("host", view, q, m0, sf, 10)
    n_o, m_o = hb.search_projection_sim("oracle", view, q, m0, sf, 10)
    assert n_h == n_o and np.array_equal(m_h, m_o) and n_o > 50
    v0, keep0 = hb.frame_view(k0, d0)
    q21 = _win_queries(hb, k1, d1, rng, 2.0, 1.0)
    n_h, s_h = hb.search_sim3("host", v0, view, q, q21, sf, sf, 7.5)
    n_o, s_o = hb.search_sim3("oracle", v0, view, q, q21, sf, sf, 7.5)
    assert n_h == n_o and np.array_equal(s_h, s_o) and n_o > 50


@pytest.mark.parametrize("mode,stereo,shifted", [(0, False, False), (0, True, True), (1, False, True), (2, True, False)])
def test_resident_grid_and_window_search(gpu_lib, ob, synth, mode, stereo, shifted):
    """8f-1: AssignFeaturesToGrid + GetFeaturesInArea + best/second loop on resident batch results."""
    import torch
    import host_binding as hb
    n, bounds, mbf, th = 4, (0.0, 640.0, 0.0, 480.0), 40.0, 15.0
    rng = np.random.default_rng(77 + mode)
    frames = synth.frames(5, 0, n)
    depth = (1.0 + 2.0 * rng.random((n, 480, 640))).astype(np.float32)
    depth[rng.random(depth.shape) < 0.2] = 0
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d_frames, d_depth = torch.from_numpy(frames).cuda(), torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    ext.extract_batch_device(d_frames.data_ptr(), 480 * 640, 640, 640, 480, n)
    d_kps, d_desc, d_counts, cap = ext.batch_results_device()
    d_ur = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    d_dep = torch.zeros_like(d_ur)
    d_cell = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    ext.rgbd_glue_batch_device(d_depth.data_ptr(), False, 1.0, 480 * 640 * 4, 640 * 4, mbf, bounds, d_ur.data_ptr(), d_dep.data_ptr(),
                               d_cell.data_ptr())
    ext.sync()
    mt = gpu_lib.OrbMatcher()
    d_start = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_items = torch.full((n, cap), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    mt.grid_build_batch_device(d_cell.data_ptr(), d_counts, n, cap, d_start.data_ptr(), d_items.data_ptr())
    pq = torch.arange(0, n - 1, dtype=torch.int32, device="cuda")
    pt = torch.arange(1, n, dtype=torch.int32, device="cuda")
    uv = rng.uniform(-30, 700, (n - 1, cap, 2)).astype(np.float32) if shifted else None
    host = [ext.batch_fetch(f) for f in range(n)]
    if shifted:  # most queries near their own position, a few far outside the image
        for p in range(n - 1):
            m = len(host[p][0])
            near = rng.random(m) < 0.9
            uv[p, :m, 0] = np.where(near, host[p][0]["x"] + rng.uniform(-6, 6, m).astype(np.float32), uv[p, :m, 0])
            uv[p, :m, 1] = np.where(near, host[p][0]["y"] + rng.uniform(-6, 6, m).astype(np.float32), uv[p, :m, 1])
    invz = (1.0 / (1.0 + 2.0 * rng.random((n - 1, cap)))).astype(np.float32) if stereo else None
    d_uv = torch.from_numpy(uv).cuda() if shifted else None
    d_invz = torch.from_numpy(invz).cuda() if stereo else None
    d_out = torch.zeros((n - 1, cap, 4), dtype=torch.int32, device="cuda")
    sf = ext.tables()["scale"]
    torch.cuda.synchronize()
    mt.window_best2_batch_device(d_kps, d_desc, d_counts, d_start.data_ptr(), d_items.data_ptr(), pq.data_ptr(), pt.data_ptr(), n - 1, cap, sf,
                                 th, d_out.data_ptr(), mode=mode, bounds=bounds, d_query_uv=d_uv.data_ptr() if shifted else None,
                                 d_query_invz=d_invz.data_ptr() if stereo else None, d_u_right=d_ur.data_ptr() if stereo else None, mbf=mbf)
    mt.sync()
    torch.cuda.synchronize()
    out, start, items, cell, ur = d_out.cpu().numpy(), d_start.cpu().numpy(), d_items.cpu().numpy(), d_cell.cpu().numpy(), d_ur.cpu().numpy()
    nfound = 0
    for f in range(n):  # the CSR itself: every keypoint in its cell, ascending inside a cell
        m = len(host[f][0])
        assert start[f, -1] == (cell[f, :m] >= 0).sum()
        for c in np.unique(cell[f, :m]):
            if c >= 0:
                assert np.array_equal(items[f, start[f, c]:start[f, c + 1]], np.nonzero(cell[f, :m] == c)[0])
    for p in range(n - 1):
        qk, qd = host[p]
        tk, td = host[p + 1]
        view, keep = hb.frame_view(tk, td, ur[p + 1, :len(tk)] if stereo else None, bounds)
        want = hb.window_best2(view, qk, qd, sf, th, mode=mode, query_uv=uv[p, :len(qk)] if shifted else None,
                               query_invz=invz[p, :len(qk)] if stereo else None, mbf=mbf)
        got = out[p, :len(qk)]
        for k, name in enumerate(("best_idx", "best_dist", "second_idx", "second_dist")):
            assert np.array_equal(got[:, k], want[name]), (p, name)
        nfound += int((want["best_idx"] >= 0).sum())
    assert nfound > 500


def test_resident_search_with_distorted_camera(gpu_lib, ob, synth):
    """Undistort -> glue (cells from mvKeysUn, bounds from ComputeImageBounds) -> CSR grid -> window search
    around the undistorted positions; the oracle gets a frame view of the undistorted keypoints and the same bounds."""
    import torch
    import host_binding as hb
    n, th = 3, 12.0
    fx, fy, cx, cy = 517.306408, 516.469215, 318.643040, 255.313989
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    bounds = gpu_lib.image_bounds(640, 480, fx, fy, cx, cy, dist)
    assert bounds != (0.0, 640.0, 0.0, 480.0)
    frames = synth.frames(6, 0, n)
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d_frames = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ext.extract_batch_device(d_frames.data_ptr(), 480 * 640, 640, 640, 480, n)
    d_kps, d_desc, d_counts, cap = ext.batch_results_device()
    d_un = torch.zeros((n, cap, 7), dtype=torch.float32, device="cuda")
    d_cell = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ext.undistort_batch_device(fx, fy, cx, cy, dist, d_un.data_ptr())
    ext.rgbd_glue_batch_device(None, False, 1.0, 0, 0, 0.0, bounds, None, None, d_cell.data_ptr(), d_kps_un=d_un.data_ptr())
    ext.sync()
    mt = gpu_lib.OrbMatcher()
    d_start = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_items = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    mt.grid_build_batch_device(d_cell.data_ptr(), d_counts, n, cap, d_start.data_ptr(), d_items.data_ptr())
    pq = torch.arange(0, n - 1, dtype=torch.int32, device="cuda")
    pt = torch.arange(1, n, dtype=torch.int32, device="cuda")
    d_uv = d_un[:n - 1, :, :2].contiguous()                     # query positions = the undistorted keypoints of the query frame
    d_out = torch.zeros((n - 1, cap, 4), dtype=torch.int32, device="cuda")
    sf = ext.tables()["scale"]
    torch.cuda.synchronize()
    mt.window_best2_batch_device(d_un.data_ptr(), d_desc, d_counts, d_start.data_ptr(), d_items.data_ptr(), pq.data_ptr(), pt.data_ptr(),
                                 n - 1, cap, sf, th, d_out.data_ptr(), mode=0, bounds=bounds, d_query_uv=d_uv.data_ptr())
    mt.sync()
    torch.cuda.synchronize()
    un = d_un.cpu().numpy().view(np.uint8).reshape(n, cap, 28)
    out = d_out.cpu().numpy()
    found = 0
    for p in range(n - 1):
        (kq, dq), (kt, dt) = ext.batch_fetch(p), ext.batch_fetch(p + 1)
        uq = np.frombuffer(un[p, :len(kq)].tobytes(), ob.KP_DTYPE)
        ut = np.frombuffer(un[p + 1, :len(kt)].tobytes(), ob.KP_DTYPE)
        view, keep = hb.frame_view(ut, dt, None, bounds)
        want = hb.window_best2(view, uq, dq, sf, th, mode=0, query_uv=np.stack([uq["x"], uq["y"]], 1))
        got = out[p, :len(kq)]
        for c, name in enumerate(("best_idx", "best_dist", "second_idx", "second_dist")):
            assert np.array_equal(got[:, c], want[name]), (p, name)
        found += int((want["best_idx"] >= 0).sum())
    assert found > 500


# ---------------------------------------------------------------------------------------------
# The reference-signature adaptors (amos-slam_amd/host/ORBmatcher_adaptors.h), end to end on stand-in
# Frame / MapPoint objects (tests/host/ref_standins.h) against the oracle.

def _project_like_the_adaptor(T, world, fx, fy, cx, cy):
    """`Rcw*x3Dw+tcw` as OpenCV evaluates it for CV_32F (one gemm: double accumulation, one rounding), then the
    reference's float arithmetic (ORBmatcher.cc:1613-1622): xc, yc, invzc = 1.0/zc (double division stored to float),
    u = fx*xc*invzc+cx with every product and sum rounded to float (the host library is built with -ffp-contract=off)."""
    T = np.asarray(T, np.float32).reshape(4, 4)
    R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    w = np.asarray(world, np.float32).astype(np.float64)
    cam = (w @ R.T + t).astype(np.float32)
    xc, yc, zc = cam[:, 0], cam[:, 1], cam[:, 2]
    with np.errstate(divide="ignore"):
        invz = (1.0 / zc.astype(np.float64)).astype(np.float32)
    u = (np.float32(fx) * xc) * invz + np.float32(cx)
    v = (np.float32(fy) * yc) * invz + np.float32(cy)
    return u.astype(np.float32), v.astype(np.float32), invz


def _pose(rx, ry, rz, t):
    cx_, sx = np.cos(rx), np.sin(rx)
    cy_, sy = np.cos(ry), np.sin(ry)
    cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(np.float32)


@pytest.mark.parametrize("motion", ["sideways", "forward", "backward", "mono"])
def test_reference_signature_search_by_projection_last_frame(hb, ob, synth, motion):
    """int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    (src/ORBmatcher.cc:1569) exactly as Tracking::TrackWithMotionModel calls it: the adaptor projects LastFrame's map
    points with CurrentFrame.mTcw, derives bForward / bBackward from the two poses, runs the search and writes
    CurrentFrame.mvpMapPoints.  Expected result: the oracle's orc_search_by_projection_frame on queries computed here."""
    k0, d0, k1, d1, sf = _two_frames(ob, synth, stream=23)
    rng = np.random.default_rng(11)
    fx, fy, cx, cy, mb, mbf = 535.4, 539.2, 320.1, 247.6, 0.08, 40.0
    n0, n1 = len(k0), len(k1)
    # last frame at the origin; its map points = back-projected keypoints at random depths
    T_last = _pose(0.002, -0.003, 0.001, [0.01, -0.02, 0.005])
    depth = rng.uniform(0.8, 6.0, n0)
    ray = np.stack([(k0["x"] - cx) / fx, (k0["y"] - cy) / fy, np.ones(n0)], 1) * depth[:, None]
    Tl = T_last.astype(np.float64)
    world = ((ray - Tl[:3, 3]) @ Tl[:3, :3]).astype(np.float32)  # Rlw^T (x_l - tlw)
    dz = {"sideways": 0.0, "forward": -0.5, "backward": 0.5, "mono": -0.5}[motion]  # camera moves along +z <=> tcw.z decreases
    T_cur = (_pose(0.001, 0.002, -0.001, [0.004, 0.002, dz]).astype(np.float64) @ Tl).astype(np.float32)
    has_point = rng.random(n0) < 0.8
    outlier = rng.random(n0) < 0.1
    obs = np.where(rng.random(n0) < 0.6, rng.integers(1, 5, n0), 0).astype(np.int32)
    world[5] = [0, 0, -3]  # behind the camera: invzc < 0
    ur = np.where(rng.random(n1) < 0.7, k1["x"] - rng.uniform(5, 40, n1).astype(np.float32), np.float32(-1)).astype(np.float32)
    cur_cam, last_cam = hb.test_camera(T_cur, sf, fx, fy, cx, cy, mb, mbf), hb.test_camera(T_last, sf, fx, fy, cx, cy, mb, mbf)
    # what the oracle is given: the queries the adaptor should have built (ORBmatcher.cc:1604-1625), in feature order
    u, v, invz = _project_like_the_adaptor(T_cur, world, fx, fy, cx, cy)
    ok = has_point & ~outlier & ~(invz < 0) & ~(u < 0) & ~(u > 640) & ~(v < 0) & ~(v > 480)
    src = np.nonzero(ok)[0]
    q = np.zeros(len(src), hb.PROJ_QUERY)
    q["u"], q["v"], q["invz"] = u[src], v[src], invz[src]
    q["octave"], q["angle"], q["desc"], q["has_obs"] = k0["octave"][src], k0["angle"][src], d0[src], obs[src] > 0
    # bForward / bBackward (ORBmatcher.cc:1588-1599): tlc = Rlw * twc + tlw, twc = -Rcw^T tcw
    Tc, Tl64 = T_cur.astype(np.float64), T_last.astype(np.float64)
    twc = (-(Tc[:3, :3].T @ Tc[:3, 3])).astype(np.float32).astype(np.float64)
    tlc = (Tl64[:3, :3] @ twc + Tl64[:3, 3]).astype(np.float32)
    mono = motion == "mono"
    forward, backward = bool(tlc[2] > mb and not mono), bool(-tlc[2] > mb and not mono)
    assert (forward, backward) == {"sideways": (False, False), "forward": (True, False), "backward": (False, True), "mono": (False, False)}[motion]
    view, keep = hb.frame_view(k1, d1, ur)
    for th in (7.0, 15.0):
        want_n, want = hb.search_frame("oracle", view, q, np.full(n1, -1, np.int32), sf, mbf, th, int(forward), int(backward))
        got_n, got = hb.ref_search_last_frame(cur_cam, k1, d1, ur, None, last_cam, k0, k0, has_point, outlier, world, d0, obs, th, mono)
        want_feat = np.where(want >= 0, src[np.maximum(want, 0)], -1)  # query index -> last-frame feature index
        assert got_n == want_n and np.array_equal(got, want_feat)
        assert got_n > 30
    # occupants present on entry (not what Tracking does, but what the signature allows): those with observations block
    # their feature, the others may be overwritten
    occ = np.where(rng.random(n1) < 0.15, rng.integers(0, 3, n1), -1).astype(np.int32)
    cur0 = np.full(n1, -1, np.int32)
    q2 = np.concatenate([q, np.zeros(int((occ >= 0).sum()), hb.PROJ_QUERY)])
    q2["u"][len(q):], q2["v"][len(q):] = -1.0e9, -1.0e9
    q2["has_obs"][len(q):] = occ[occ >= 0] > 0
    cur0[occ >= 0] = len(q) + np.arange(int((occ >= 0).sum()))
    want_n, want = hb.search_frame("oracle", view, q2, cur0, sf, mbf, 15.0, int(forward), int(backward))
    got_n, got = hb.ref_search_last_frame(cur_cam, k1, d1, ur, occ, last_cam, k0, k0, has_point, outlier, world, d0, obs, 15.0, mono)
    want_feat = np.where(want >= len(q), -2, np.where(want >= 0, src[np.clip(want, 0, len(src) - 1)], -1))
    assert got_n == want_n and np.array_equal(got, want_feat)
    assert ((occ > 0) <= (got == -2)).all(), "an occupant with observations is never displaced"


def test_reference_signature_search_by_projection_local_points(hb, ob, synth):
    """int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th)
    (src/ORBmatcher.cc:70): mbTrackInView / isBad filter, the mTrack* members, F.mvpMapPoints written back."""
    k0, d0, k1, d1, sf = _two_frames(ob, synth, stream=24)
    rng = np.random.default_rng(12)
    n0, n1 = len(k0), len(k1)
    ur = np.where(rng.random(n1) < 0.5, k1["x"] - 20, -1).astype(np.float32)
    pts = np.zeros(n0, hb.MAP_QUERY)
    pts["proj_x"], pts["proj_y"] = k0["x"] - 2, k0["y"] - 1
    pts["proj_xr"] = pts["proj_x"] - 20 + rng.normal(0, 3, n0).astype(np.float32)
    pts["view_cos"] = rng.choice([0.9, 0.999], n0).astype(np.float32)
    pts["level"] = np.maximum(k0["octave"], 0)
    pts["has_obs"] = rng.integers(0, 3, n0)
    pts["desc"] = d0
    in_view, bad = rng.random(n0) < 0.85, rng.random(n0) < 0.05
    src = np.nonzero(in_view & ~bad)[0]
    cam = hb.test_camera(np.eye(4), sf)
    view, keep = hb.frame_view(k1, d1, ur)
    q = pts[src].copy()
    q["has_obs"] = q["has_obs"] > 0
    for th in (1.0, 3.0):
        want_n, want, _ = hb.search_points("oracle", view, q, np.full(n1, -1, np.int32), np.zeros(n1, np.uint8), sf, th)
        got_n, got = hb.ref_search_local_points(cam, k1, d1, ur, pts, in_view, bad, th)
        assert got_n == want_n and np.array_equal(got, np.where(want >= 0, src[np.maximum(want, 0)], -1))
    assert got_n > 50


def test_pyramid_is_downloaded_on_demand_and_the_device_is_the_callers(hb, ob, synth, monkeypatch):
    """Default pyramid mode: the 3-arg operator() (RGB-D Amos flow, Frame.cc:484) copies no pixel, yet every mvImagePyramid[l] has the level's
    rows / cols (Frame.cc:1197 reads mvImagePyramid[0].rows); DownloadPyramid() then gives the padded planes (stereo, Frame.cc:1401,1434;
    the 4-arg operator() does it by itself: test_extractor_call_operator).  The handle sits on the calling thread's current device; AMOS_DEVICE
    overrides it (a device that does not exist is refused: the variable is honoured, nothing is hard-wired to device 0)."""
    img = synth.frame(13, 5)
    rc, pyr, dev = hb.host_pyramid_on_demand(img, pyr_level=3)
    orc = ob.Oracle()
    orc.extract(img)
    lw, lh = orc.level_sizes(640, 480)
    assert rc[:, 0].tolist() == [int(v) for v in lh] and rc[:, 1].tolist() == [int(v) for v in lw]
    _same(pyr, orc.level_image(3, padded=True), "mvImagePyramid[3] after DownloadPyramid()")
    import torch
    assert dev == torch.cuda.current_device() == 0
    monkeypatch.setenv("AMOS_DEVICE", "5")
    with pytest.raises(RuntimeError, match="(?i)device|ordinal"):
        hb.host_pyramid_on_demand(img)
    monkeypatch.setenv("AMOS_DEVICE", "0")
    assert hb.host_pyramid_on_demand(img, pyr_level=3)[2] == 0


def test_extractors_and_matchers_on_separate_threads(hb, ob, synth):
    """SURVEY 8b threading: two ORBextractor instances on two std::threads (stereo, Frame.cc:165-170) beside three threads that each construct
    an ORBmatcher on the stack per search (Tracking / LocalMapping / LoopClosing), all at once: every result equals the oracle's, every
    repetition equals the first, and the stack-constructed matchers reuse the pool's handles (3 threads x 6 constructions -> at most a few
    handles ever created, not 18)."""
    left, right = synth.frame(21, 3), synth.frame(22, 7)
    k0, d0, k1, d1, sf = _two_frames(ob, synth)
    rng = np.random.default_rng(9)
    searches = []
    for t in range(3):
        ur = np.where(rng.random(len(k1)) < 0.7, k1["x"] - rng.uniform(5, 40, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
        view, keep = hb.frame_view(k1, d1, ur)
        q = np.zeros(len(k0), hb.PROJ_QUERY)
        q["u"] = k0["x"] - 2 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
        q["v"] = k0["y"] - 1 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
        q["invz"] = rng.uniform(0.2, 2.0, len(k0)).astype(np.float32)
        q["octave"], q["angle"], q["desc"] = k0["octave"], k0["angle"], d0
        q["has_obs"] = rng.random(len(k0)) < 0.6
        searches.append((view, q, np.full(len(k1), -1, np.int32), sf, 40.0, (7.0, 15.0, 11.0)[t], (0, 1, 0)[t], (0, 0, 1)[t], keep))
    before = hb.host_run_threads([], [s[:8] for s in searches[:1]], iters=1)[2]   # (whatever earlier tests left in the pool)
    ext, res, pool = hb.host_run_threads([left, right], [s[:8] for s in searches], iters=6)
    for img, (kps, desc) in zip((left, right), ext):
        ko, do = ob.Oracle().extract(img)
        _same(kps, ko, "keypoints")
        _same(desc, do, "descriptors")
    for (view, q, m0, sf_, mbf, th, fwd, bwd, _), (r, m) in zip(searches, res):
        ro, mo = hb.search_frame("oracle", view, q, m0, sf_, mbf, th, fwd, bwd)
        assert r == ro and np.array_equal(m, mo) and r > 50
    assert pool - before <= 3, (before, pool)   # three concurrent matchers at most: 18 constructions, <= 3 new handles


def test_frame_latency_harness_runs_the_reference_call_sequence(hb, ob, synth):
    """amos_host_frame_latency (bench.py's drop_in_latency.cxx_* figures) without the mask network: 3-arg operator() -> MovingKeyPoints ->
    ProcessDesp -> stack ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) on consecutive synthetic frames: the keypoint count is the
    oracle's for the last frame (an all-zero mask gates nothing) and most features find their match."""
    frames = np.stack([synth.frame(31, k) for k in range(4)])
    out = hb.host_frame_latency(frames, warm=2, iters=6)   # 8 calls: the last frame processed is frame 3
    ko, _ = ob.Oracle().extract(frames[3])
    assert out["keypoints_last_frame"] == len(ko) and out["matches_last_frame"] > 300, out
    assert out["eval_image_ms"] < 1.0 and out["eval_image_false"] == 0   # no network in this run
    assert 0 < out["detect_ms"] and 0 < out["process_desp_ms"] and 0 < out["search_by_projection_ms"] and out["frame_ms"] < 50
