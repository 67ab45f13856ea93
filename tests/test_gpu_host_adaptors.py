"""GPU: the eight reference-signature ORBmatcher searches that round 2 only compiled (amos-slam_amd/host/ORBmatcher_adaptors.h:
SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), SearchByProjection(KeyFrame*, Scw, ...), both SearchByBoW,
SearchForInitialization, SearchForTriangulation, SearchBySim3, both Fuse) EXECUTED on a stand-in map -- Frame / KeyFrame / MapPoint
objects of tests/host/ref_standins.h built from arrays (host_capi.cc) -- and held to the oracle: the pre-filter lines of
src/ORBmatcher.cc are restated in tests/adaptor_prefilter.py, the queries they produce go through the oracle's orc_search_*
restatements, and match vectors, replaced points, observation counts and return values must be identical."""
import numpy as np
import pytest

import adaptor_prefilter as pf

pytestmark = pytest.mark.gpu
f32 = np.float32
FX, FY, CX, CY, MB, MBF = 535.4, 539.2, 320.1, 247.6, 0.08, 40.0


@pytest.fixture(scope="module")
def hb(gpu_lib):
    import host_binding
    host_binding.host()
    return host_binding


def _pose(rx, ry, rz, t):
    cx_, sx, cy_, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(f32)


class Scene:
    """Two extracted frames of one synthetic stream as keyframes 0 and 1, and a table of map points: point i (i < n0) is the
    back-projection of keyframe 0's feature i, point n0 + j that of keyframe 1's feature j; normals, distance ranges, observation
    counts and bad flags as a map would hold them, with a share of points failing each pre-filter test."""

    def __init__(self, hb, ob, synth, stream, seed):
        orc = ob.Oracle()
        self.k0, self.d0 = orc.extract(synth.frame(stream, 20))
        self.k1, self.d1 = orc.extract(synth.frame(stream, 21))
        self.sf = orc.tables()["scale"].astype(f32)
        self.nl = len(self.sf)
        rng = self.rng = np.random.default_rng(seed)
        k0, k1 = self.k0, self.k1
        n0, n1 = len(k0), len(k1)
        self.n0, self.n1 = n0, n1
        self.T0 = _pose(0.002, -0.003, 0.001, [0.01, -0.02, 0.005])
        # frame 21 is frame 20 moved by (-2, -1) px: a small rotation of the camera gives that shift to points at every depth
        cands = [(_pose(sx * 1.0 / FY, sy * 2.0 / FX, 0, [0.003, 0.001, 0.002]).astype(np.float64) @ self.T0.astype(np.float64)).astype(f32)
                 for sx in (-1, 1) for sy in (-1, 1)]

        def backproject(k, T, depth):
            ray = np.stack([(k["x"] - CX) / FX, (k["y"] - CY) / FY, np.ones(len(k))], 1) * depth[:, None]
            T64 = T.astype(np.float64)
            return ((ray - T64[:3, 3]) @ T64[:3, :3]).astype(f32)

        w0 = backproject(k0, self.T0, rng.uniform(0.8, 6.0, n0))
        cam = hb.test_camera(np.eye(4), self.sf, FX, FY, CX, CY, MB, MBF)
        shifts = []
        for T in cands:
            u, v, _ = pf.project_kf(cam, pf.transform(*pf.pose(T), w0))
            shifts.append(abs(np.median(u - k0["x"]) + 2) + abs(np.median(v - k0["y"]) + 1))
        self.T1 = cands[int(np.argmin(shifts))]
        assert min(shifts) < 1.0
        w1 = backproject(k1, self.T1, rng.uniform(0.8, 6.0, n1))
        self.world = np.concatenate([w0, w1]).astype(f32)
        npts = n0 + n1
        own_T = [self.T0] * n0 + [self.T1] * n1
        Ow = np.stack([pf.centre(*pf.pose(T)) for T in (self.T0, self.T1)])
        PO = self.world - np.concatenate([np.repeat(Ow[:1], n0, 0), np.repeat(Ow[1:], n1, 0)])
        dist = np.linalg.norm(PO.astype(np.float64), axis=1)
        normal = PO / dist[:, None] + rng.normal(0, 0.15, (npts, 3))
        far = rng.random(npts) < 0.10           # viewing angle beyond 60 degrees
        normal[far] = np.cross(normal[far], [0.3, 1.0, 0.2])
        self.normal = (normal / np.linalg.norm(normal, axis=1)[:, None]).astype(f32)
        level = np.concatenate([k0["octave"], k1["octave"]])
        self.max_dist = (dist * self.sf[level]).astype(f32)       # MapPoint::UpdateNormalAndDepth
        self.min_dist = (self.max_dist / self.sf[-1]).astype(f32)
        out_of_range = rng.random(npts) < 0.06
        self.max_dist[out_of_range] *= f32(0.3)
        self.min_dist[out_of_range] *= f32(0.3)
        self.desc = np.concatenate([self.d0, self.d1])
        self.obs = rng.integers(1, 6, npts).astype(np.int32)
        self.bad = (rng.random(npts) < 0.05).astype(np.uint8)
        self.ur0 = np.where(rng.random(n0) < 0.6, k0["x"] - rng.uniform(5, 30, n0), -1).astype(f32)
        self.ur1 = np.where(rng.random(n1) < 0.6, k1["x"] - rng.uniform(5, 30, n1), -1).astype(f32)
        self.point_of0 = np.where(rng.random(n0) < 0.8, np.arange(n0), -1).astype(np.int32)
        self.point_of1 = np.where(rng.random(n1) < 0.4, n0 + np.arange(n1), -1).astype(np.int32)
        self.cam0 = hb.test_camera(self.T0, self.sf, FX, FY, CX, CY, MB, MBF)
        self.cam1 = hb.test_camera(self.T1, self.sf, FX, FY, CX, CY, MB, MBF)
        self.hb = hb
        self.pts, self._keep_pts = hb.test_points(self.world, self.normal, self.desc, self.obs, self.bad, self.min_dist, self.max_dist)

    def nodes(self, desc, rng=None):
        rng = rng or self.rng
        ids = (desc[:, 0].astype(np.uint32) >> 2) * 3 + 5
        out = {}
        for i in rng.permutation(len(desc)):
            out.setdefault(int(ids[i]), []).append(int(i))
        return out

    def points_table(self, **over):
        a = dict(world=self.world, normal=self.normal, desc=self.desc, obs=self.obs, bad=self.bad, min_dist=self.min_dist, max_dist=self.max_dist)
        a.update(over)
        return self.hb.test_points(**a)


@pytest.fixture(scope="module")
def scene(hb, ob, synth):
    return Scene(hb, ob, synth, stream=31, seed=21)


def test_search_by_projection_relocalisation(hb, scene):
    """int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist),
    src/ORBmatcher.cc:1731 (Tracking::Relocalization): the keyframe's map points projected with CurrentFrame.mTcw."""
    s = scene
    rng = np.random.default_rng(1)
    already = (rng.random(s.pts.n) < 0.1).astype(np.uint8)
    occupants = np.where(rng.random(s.n1) < 0.1, rng.integers(0, s.n0, s.n1), -1).astype(np.int32)   # CurrentFrame.mvpMapPoints on entry
    kf, k0 = hb.test_kf(s.cam0, s.k0, s.d0, s.ur0, s.point_of0)
    cur, k1 = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, occupants)
    has = s.point_of0 >= 0
    p = np.maximum(s.point_of0, 0)
    feat_ok = has & (s.bad[p] == 0) & (already[p] == 0)
    q, q_point = pf.reloc_queries(hb, s.cam1, s.T1, s.world, s.min_dist, s.max_dist, s.desc, s.point_of0, feat_ok, s.k0["angle"], s.sf[1], s.nl)
    assert 300 < len(q) < int(feat_ok.sum())  # some fail the bounds / distance tests
    view, keep = hb.frame_view(s.k1, s.d1, s.ur1)
    m0 = np.where(occupants >= 0, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    total = 0
    for th, orb_dist, ori in ((10.0, 100, True), (3.0, 64, False)):
        want_n, want = hb.search_kf("oracle", view, q, m0, s.sf, th, orb_dist, 0.9, ori)
        got_n, got = hb.ref_search_reloc(cur, kf, s.pts, already, th, orb_dist, 0.9, ori)
        want_pts = np.where(want >= 0, q_point[np.maximum(want, 0)], np.where(want == hb.MATCH_TAKEN, occupants, -1))
        assert got_n == want_n and np.array_equal(got, want_pts)
        total += got_n
    assert total > 100


def test_search_by_projection_keyframe_sim3_pose(hb, scene):
    """int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th), :388
    (LoopClosing::ComputeSim3): Scw's scale divided out, points already in vpMatched skipped, matched features blocked."""
    s = scene
    rng = np.random.default_rng(2)
    kf, keep_kf = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, s.point_of1)
    scale = f32(1.1)
    Scw = s.T1.copy()
    Scw[:3, :] *= scale
    vp = rng.permutation(s.n0)[: s.n0 - 40].astype(np.int32)              # candidate points: those seen from keyframe 0, shuffled
    matched0 = np.full(s.n1, -1, np.int32)
    taken = rng.random(s.n1) < 0.15
    matched0[taken] = rng.choice(vp, int(taken.sum()))                    # vpMatched on entry: some of them among the candidates
    R, t = pf.unscaled(Scw)
    keep = (s.bad[vp] == 0) & ~np.isin(vp, matched0[matched0 >= 0])
    q, pos = pf.keyframe_queries(hb, s.cam1, R, t, pf.centre(R, t), s.world, s.normal, s.min_dist, s.max_dist, s.desc, vp, keep, False, s.sf[1], s.nl)
    assert 300 < len(q) < int(keep.sum())
    view, keepv = hb.frame_view(s.k1, s.d1, s.ur1)
    m0 = np.where(matched0 >= 0, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    for th in (10, 4):
        want_n, want = hb.search_projection_sim("oracle", view, q, m0, s.sf, th)
        got_n, got = hb.ref_search_kf_scw(kf, s.pts, Scw, vp, matched0, th)
        want_pts = np.where(want >= 0, vp[pos[np.maximum(want, 0)]], matched0)
        assert got_n == want_n and np.array_equal(got, want_pts)
    assert got_n > 50


def test_search_by_bow_keyframe_frame(hb, scene):
    """int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), :230: only keyframe features with a good map
    point take part; F's matches are those points."""
    s = scene
    rng = np.random.default_rng(3)
    n0_nodes, n1_nodes = s.nodes(s.d0, rng), s.nodes(s.d1, rng)
    for drop in list(n1_nodes)[::6]:
        del n1_nodes[drop]
    kf, keep0 = hb.test_kf(s.cam0, s.k0, s.d0, s.ur0, s.point_of0, n0_nodes)
    fr, keep1 = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, None, n1_nodes)
    has = ((s.point_of0 >= 0) & (s.bad[np.maximum(s.point_of0, 0)] == 0)).astype(np.uint8)
    vkf, ka = hb.bow_view(s.k0, s.d0, n0_nodes, has)
    vf, kb = hb.bow_view(s.k1, s.d1, n1_nodes)
    for ratio, ori in ((0.7, True), (0.9, False)):
        want_n, want = hb.search_bow("oracle", vkf, vf, ratio, ori)
        got_n, got = hb.ref_search_bow_kf_frame(kf, fr, s.pts, ratio, ori)
        assert got_n == want_n and np.array_equal(got, np.where(want >= 0, s.point_of0[np.maximum(want, 0)], -1))
    assert got_n > 30 and not (s.bad[got[got >= 0]]).any()


def test_search_by_bow_two_keyframes(hb, scene):
    """int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), :656 (loop closing)."""
    s = scene
    rng = np.random.default_rng(4)
    nd0, nd1 = s.nodes(s.d0, rng), s.nodes(s.d1, rng)
    point_of1 = np.where(rng.random(s.n1) < 0.85, s.n0 + np.arange(s.n1), -1).astype(np.int32)
    kf1, keep0 = hb.test_kf(s.cam0, s.k0, s.d0, s.ur0, s.point_of0, nd0)
    kf2, keep1 = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, point_of1, nd1)
    has1 = ((s.point_of0 >= 0) & (s.bad[np.maximum(s.point_of0, 0)] == 0)).astype(np.uint8)
    has2 = ((point_of1 >= 0) & (s.bad[np.maximum(point_of1, 0)] == 0)).astype(np.uint8)
    v1, ka = hb.bow_view(s.k0, s.d0, nd0, has1)
    v2, kb = hb.bow_view(s.k1, s.d1, nd1, has2)
    for ratio, ori in ((0.75, True), (0.95, False)):
        want_n, want = hb.search_bow_kf("oracle", v1, v2, ratio, ori)
        got_n, got = hb.ref_search_bow_kf_kf(kf1, kf2, s.pts, ratio, ori)
        assert got_n == want_n and np.array_equal(got, np.where(want >= 0, point_of1[np.maximum(want, 0)], -1))
    assert got_n > 30 and (got[got >= 0] >= s.n0).all()   # the matches are keyframe 2's points


def test_search_for_initialization(hb, scene):
    """int SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize), :515"""
    s = scene
    f1, keep0 = hb.test_kf(s.cam0, s.k0, s.d0)
    f2, keep1 = hb.test_kf(s.cam1, s.k1, s.d1)
    v1, ka = hb.frame_view(s.k0, s.d0)
    v2, kb = hb.frame_view(s.k1, s.d1)
    prev = np.stack([s.k0["x"], s.k0["y"]], axis=1).astype(f32)
    for window, ratio, ori in ((20, 0.9, True), (100, 0.9, True), (100, 0.8, False)):
        want_n, want, want_prev = hb.search_init("oracle", v1, v2, prev, window, ratio, ori)
        got_n, got, got_prev = hb.ref_search_initialization(f1, f2, prev, window, ratio, ori)
        assert got_n == want_n and np.array_equal(got, want) and np.array_equal(got_prev, want_prev)
    assert got_n > 30


@pytest.mark.parametrize("only_stereo", [0, 1])
def test_search_for_triangulation(hb, scene, only_stereo):
    """int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, bool bOnlyStereo),
    :810 (LocalMapping::CreateNewMapPoints): features WITHOUT a map point, the epipole of pKF1's centre in pKF2 from the two poses."""
    s = scene
    rng = np.random.default_rng(5 + only_stereo)
    nd0, nd1 = s.nodes(s.d0, rng), s.nodes(s.d1, rng)
    po0 = np.where(rng.random(s.n0) < 0.3, np.arange(s.n0), -1).astype(np.int32)
    po1 = np.where(rng.random(s.n1) < 0.3, s.n0 + np.arange(s.n1), -1).astype(np.int32)
    # keyframe 2 = keyframe 1's pose moved sideways: a real epipole, far outside the image
    T2 = (_pose(0, 0, 0, [0.3, 0.05, 0.02]).astype(np.float64) @ s.T1.astype(np.float64)).astype(f32)
    cam2 = hb.test_camera(T2, s.sf, FX, FY, CX, CY, MB, MBF)
    kf1, keep0 = hb.test_kf(s.cam0, s.k0, s.d0, s.ur0, po0, nd0)
    kf2, keep1 = hb.test_kf(cam2, s.k1, s.d1, s.ur1, po1, nd1)
    f12 = np.array([[0, 0, 1.0], [0, 0, -2.0], [-1.0, 2.0, 0]], f32)   # the image translation (2, 1) between the two frames
    ex, ey = pf.epipole(cam2, s.T0, T2)
    v1, ka = hb.bow_view(s.k0, s.d0, nd0, (po0 >= 0).astype(np.uint8), s.ur0)
    v2, kb = hb.bow_view(s.k1, s.d1, nd1, (po1 >= 0).astype(np.uint8), s.ur1)
    want_n, want = hb.search_triangulation("oracle", v1, v2, f12, float(ex), float(ey), s.sf, (s.sf * s.sf).astype(f32), only_stereo)
    got_n, got = hb.ref_search_triangulation(kf1, kf2, s.pts, f12, only_stereo)
    assert got_n == want_n and np.array_equal(got, want)
    assert got_n > 10 and (po0[got[:, 0]] < 0).all() and (po1[got[:, 1]] < 0).all()


def test_search_by_sim3(hb, scene):
    """int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
    const cv::Mat &t12, const float th), :1314: both projection directions, the already-matched bookkeeping, agreement of the two."""
    s = scene
    rng = np.random.default_rng(6)
    point_of2 = np.where(rng.random(s.n1) < 0.85, s.n0 + np.arange(s.n1), -1).astype(np.int32)
    kf1, keep0 = hb.test_kf(s.cam0, s.k0, s.d0, s.ur0, s.point_of0)
    kf2, keep1 = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, point_of2)
    # camera 2 -> camera 1: X1 = s12 R12 X2 + t12 with the true relative pose and a scale slightly off 1
    T1, T2 = s.T0.astype(np.float64), s.T1.astype(np.float64)
    R12 = (T1[:3, :3] @ T2[:3, :3].T).astype(f32)
    t12 = (T1[:3, 3] - R12.astype(np.float64) @ T2[:3, 3]).astype(f32)
    s12 = f32(1.01)
    matches0 = np.full(s.n0, -1, np.int32)
    pre = np.nonzero((s.point_of0 >= 0) & (rng.random(s.n0) < 0.1))[0]
    feats2 = np.nonzero(point_of2 >= 0)[0]
    matches0[pre] = point_of2[rng.choice(feats2, len(pre), replace=False)]    # already matched on entry: points of keyframe 2
    c12, c21 = pf.sim3_hops(s12, R12, t12)
    done1 = matches0 >= 0
    done2 = np.zeros(s.n1, bool)
    done2[matches0[done1] - s.n0] = True                                    # GetIndexInKeyFrame(pKF2) of an already matched point
    ok1 = (s.point_of0 >= 0) & ~done1 & (s.bad[np.maximum(s.point_of0, 0)] == 0)
    ok2 = (point_of2 >= 0) & ~done2 & (s.bad[np.maximum(point_of2, 0)] == 0)
    q12 = pf.sim3_direction(hb, s.cam0, s.cam1, s.T0, c21, s.world, s.min_dist, s.max_dist, s.desc, s.point_of0, ok1, s.sf[1], s.nl)
    q21 = pf.sim3_direction(hb, s.cam0, s.cam0, s.T1, c12, s.world, s.min_dist, s.max_dist, s.desc, point_of2, ok2, s.sf[1], s.nl)
    assert len(q12) > 300 and len(q21) > 300
    v1, ka = hb.frame_view(s.k0, s.d0, s.ur0)
    v2, kb = hb.frame_view(s.k1, s.d1, s.ur1)
    for th in (7.5, 3.0):
        want_n, want = hb.search_sim3("oracle", v1, v2, q12, q21, s.sf, s.sf, th)
        got_n, got = hb.ref_search_sim3(kf1, kf2, s.pts, matches0, s12, R12, t12, th)
        assert got_n == want_n and np.array_equal(got, np.where(want >= 0, point_of2[np.maximum(want, 0)], matches0))
    assert got_n > 50


def _simulate_fuse(s, vp, q, pos, best, kf_points, ur, chi2):
    """The write-back of Fuse(pKF, vpMapPoints, th), src/ORBmatcher.cc:1038-1172, point by point in call order on a small model of the
    map (MapPoint::Replace / AddObservation as MapPoint.cc:244-310 and the stand-ins have them).  `best[k]`: the keyframe feature the
    search of query k ended on (or -1)."""
    kf_points = kf_points.copy()
    bad = s.bad.astype(bool).copy()
    obs = s.obs.copy()
    replaced = np.full(len(bad), -1, np.int32)
    in_kf = {int(p): int(i) for i, p in enumerate(kf_points) if p >= 0}   # mObservations restricted to this keyframe
    n_fused = 0
    best_of = {int(pos[k]): int(best[k]) for k in range(len(q))}

    def add_observation(p, idx):
        if p in in_kf:
            return
        in_kf[p] = idx
        obs[p] += 2 if ur[idx] >= 0 else 1

    def replace(a, b):  # a->Replace(b)
        if a == b:
            return
        bad[a] = True
        replaced[a] = b
        if a in in_kf:
            idx = in_kf.pop(a)
            if b not in in_kf:
                kf_points[idx] = b
                add_observation(b, idx)
            else:
                kf_points[idx] = -1

    for j, p in enumerate(vp):
        p = int(p)
        if p < 0 or bad[p] or p in in_kf:      # :1042-1053, evaluated when the point's turn comes
            continue
        b = best_of.get(j, -1)
        if b < 0:
            continue
        in_feat = int(kf_points[b])
        if in_feat >= 0:
            if not bad[in_feat]:
                if obs[in_feat] > obs[p]:
                    replace(p, in_feat)
                else:
                    replace(in_feat, p)
        else:
            add_observation(p, b)
            kf_points[b] = p
        n_fused += 1
    return n_fused, kf_points, replaced, obs, bad.astype(np.uint8)


def test_fuse_into_keyframe(hb, scene):
    """int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), :1020 (LocalMapping::SearchInNeighbors): NULL and
    duplicated entries, points already in the keyframe, the chi-square gate with the stereo coordinate, Replace in both directions,
    AddObservation / AddMapPoint."""
    s = scene
    rng = np.random.default_rng(7)
    kf, keep_kf = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, s.point_of1)
    vp = rng.permutation(s.n0)[: s.n0 - 30].astype(np.int32)
    vp = np.concatenate([vp[:200], [-1], vp[50:80], vp[200:], s.point_of1[s.point_of1 >= 0][:25]]).astype(np.int32)  # a NULL, 30 duplicates, 25 points of the keyframe itself
    R, t = pf.pose(s.T1)
    in_kf0 = set(int(p) for p in s.point_of1 if p >= 0)
    keep = np.array([p >= 0 and s.bad[p] == 0 and int(p) not in in_kf0 for p in vp])
    q, pos = pf.keyframe_queries(hb, s.cam1, R, t, pf.centre(R, t), s.world, s.normal, s.min_dist, s.max_dist, s.desc, np.maximum(vp, 0), keep, True, s.sf[1], s.nl)
    view, keepv = hb.frame_view(s.k1, s.d1, s.ur1)
    inv_s2 = (f32(1.0) / (s.sf * s.sf).astype(f32)).astype(f32)
    for th in (3.0, 6.0):
        _, best = hb.fuse("oracle", view, q, s.sf, th, inv_s2)
        want = _simulate_fuse(s, vp, q, pos, best, s.point_of1, s.ur1, True)
        got = hb.ref_fuse(kf, s.pts, vp, th)
        assert got[0] == want[0]
        for g, w, what in zip(got[1:], want[1:], ("keyframe map points", "mpReplaced", "Observations()", "isBad()")):
            assert np.array_equal(g, w), what
    assert got[0] > 100 and (got[2] >= 0).sum() > 10          # fused points, several of them replacements
    assert (np.bincount(got[1][got[1] >= 0]) <= 1).all()       # no point sits on two features of the keyframe


def test_fuse_with_sim3_pose(hb, scene):
    """int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint), :1179
    (LoopClosing::SearchAndFuse): no chi-square gate; a feature with a point reports it in vpReplacePoint, a free feature takes the
    point; the keyframe's own points are skipped."""
    s = scene
    rng = np.random.default_rng(8)
    kf, keep_kf = hb.test_kf(s.cam1, s.k1, s.d1, s.ur1, s.point_of1)
    Scw = s.T1.copy()
    Scw[:3, :] *= f32(0.93)
    vp = rng.permutation(s.n0)[: s.n0 - 30].astype(np.int32)
    vp = np.concatenate([vp, vp[10:30], s.point_of1[s.point_of1 >= 0][:25]]).astype(np.int32)   # duplicates and points of the keyframe itself
    R, t = pf.unscaled(Scw)
    own = set(int(p) for p in s.point_of1 if p >= 0 and not s.bad[p])      # pKF->GetMapPoints(): good points only
    keep = np.array([s.bad[p] == 0 and int(p) not in own for p in vp])
    q, pos = pf.keyframe_queries(hb, s.cam1, R, t, pf.centre(R, t), s.world, s.normal, s.min_dist, s.max_dist, s.desc, vp, keep, False, s.sf[1], s.nl)
    view, keepv = hb.frame_view(s.k1, s.d1, s.ur1)
    for th in (4.0, 8.0):
        _, best = hb.fuse("oracle", view, q, s.sf, th, None)
        # :1291-1308 in call order on the live keyframe
        kf_points, obs = s.point_of1.copy(), s.obs.copy()
        replace = np.full(len(vp), -1, np.int32)
        observing = {int(p) for p in kf_points if p >= 0}
        n_fused = 0
        for k in range(len(q)):
            b = int(best[k])
            if b < 0:
                continue
            p = int(vp[pos[k]])
            if kf_points[b] >= 0:
                if not s.bad[kf_points[b]]:
                    replace[pos[k]] = kf_points[b]
            else:
                if p not in observing:
                    observing.add(p)
                    obs[p] += 2 if s.ur1[b] >= 0 else 1
                kf_points[b] = p
            n_fused += 1
        got_n, got_replace, got_kf, got_obs = hb.ref_fuse_scw(kf, s.pts, Scw, vp, th, np.full(len(vp), -1, np.int32))
        assert got_n == n_fused and np.array_equal(got_replace, replace) and np.array_equal(got_kf, kf_points) and np.array_equal(got_obs, obs)
    assert got_n > 100 and (got_replace >= 0).sum() > 10 and (got_kf != s.point_of1).sum() > 10
