"""CPU: the oracle's gated searches (restatements of ORBmatcher::SearchByProjection x2,
SearchForInitialization and Frame::GetFeaturesInArea) against pure-Python restatements written
from the reference's control flow, on descriptors of two consecutive synthetic frames."""
import math

import numpy as np
import pytest

import host_binding as hb

F32 = np.float32
COLS, ROWS = 64, 48


def _round_half_away(v):
    return int(math.floor(abs(float(v)) + 0.5)) * (1 if v >= 0 else -1)


class PyGrid:
    def __init__(self, kps, bounds=(0.0, 640.0, 0.0, 480.0)):
        self.kps = kps
        self.minx, self.maxx, self.miny, self.maxy = (F32(b) for b in bounds)
        self.winv = F32(COLS) / F32(self.maxx - self.minx)
        self.hinv = F32(ROWS) / F32(self.maxy - self.miny)
        self.grid = [[[] for _ in range(ROWS)] for _ in range(COLS)]
        for i, kp in enumerate(kps):
            px = _round_half_away(F32(F32(kp["x"]) - self.minx) * self.winv)
            py = _round_half_away(F32(F32(kp["y"]) - self.miny) * self.hinv)
            if 0 <= px < COLS and 0 <= py < ROWS:
                self.grid[px][py].append(i)

    def area(self, x, y, r, lo=-1, hi=-1):
        x, y, r = F32(x), F32(y), F32(r)
        out = []
        c0 = max(0, int(math.floor(F32(F32(x - self.minx) - r) * self.winv)))
        if c0 >= COLS:
            return out
        c1 = min(COLS - 1, int(math.ceil(F32(F32(x - self.minx) + r) * self.winv)))
        if c1 < 0:
            return out
        r0 = max(0, int(math.floor(F32(F32(y - self.miny) - r) * self.hinv)))
        if r0 >= ROWS:
            return out
        r1 = min(ROWS - 1, int(math.ceil(F32(F32(y - self.miny) + r) * self.hinv)))
        if r1 < 0:
            return out
        check = lo > 0 or hi >= 0
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for i in self.grid[ix][iy]:
                    kp = self.kps[i]
                    if check:
                        if kp["octave"] < lo:
                            continue
                        if hi >= 0 and kp["octave"] > hi:
                            continue
                    if abs(F32(kp["x"]) - x) < r and abs(F32(kp["y"]) - y) < r:
                        out.append(i)
        return out


def _dist(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _three_maxima(sizes):
    m1 = m2 = m3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > m1:
            m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
        elif s > m2:
            m3, m2, i3, i2 = m2, s, i2, i
        elif s > m3:
            m3, i3 = s, i
    if m2 < F32(0.1) * F32(m1):
        i2 = i3 = -1
    elif m3 < F32(0.1) * F32(m1):
        i3 = -1
    return i1, i2, i3


def py_search_frame(kps, desc, ur, q, match, sf, mbf, th, fwd, bwd):
    grid = PyGrid(kps)
    match = match.copy()
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)
    for i, p in enumerate(q):
        radius = F32(th) * F32(sf[p["octave"]])
        o = int(p["octave"])
        cand = grid.area(p["u"], p["v"], radius, *((o, -1) if fwd else (0, o) if bwd else (o - 1, o + 1)))
        best, bi = 256, -1
        for i2 in cand:
            if match[i2] >= 0 and q[match[i2]]["has_obs"]:
                continue
            if ur is not None and ur[i2] > 0:
                u_r = F32(p["u"]) - F32(F32(mbf) * F32(p["invz"]))
                if abs(F32(u_r - F32(ur[i2]))) > radius:
                    continue
            d = _dist(p["desc"], desc[i2])
            if d < best:
                best, bi = d, i2
        if cand and best <= 100:
            match[bi] = i
            n += 1
            rot = F32(p["angle"]) - F32(kps[bi]["angle"])
            if rot < 0:
                rot = F32(rot + F32(360.0))
            b = _round_half_away(F32(rot * factor))
            hist[0 if b == 30 else b].append(bi)
    i1, i2, i3 = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (i1, i2, i3):
            for j in hist[b]:
                match[j] = -1
                n -= 1
    return n, match


@pytest.fixture(scope="module")
def frames(ob, synth):
    orc = ob.Oracle(n_features=400, n_levels=4)
    k0, d0 = orc.extract(synth.frame(30, 5))
    k1, d1 = orc.extract(synth.frame(30, 6))
    return k0, d0, k1, d1, orc.tables()["scale"]


def test_features_in_area_vs_python(frames):
    k0, d0, _, _, _ = frames
    view, keep = hb.frame_view(k0, d0)
    grid = PyGrid(k0)
    rng = np.random.default_rng(0)
    for _ in range(150):
        x, y, r = float(rng.uniform(-20, 660)), float(rng.uniform(-20, 500)), float(rng.uniform(2, 50))
        lo, hi = int(rng.integers(-1, 4)), int(rng.integers(-1, 5))
        assert hb.oracle_features_in_area(view, x, y, r, lo, hi).tolist() == grid.area(x, y, r, lo, hi)


@pytest.mark.parametrize("fwd,bwd", [(0, 0), (1, 0), (0, 1)])
def test_search_by_projection_frame_vs_python(frames, fwd, bwd):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(3)
    ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(5, 30, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur)
    q = np.zeros(len(k0), hb.PROJ_QUERY)
    q["u"] = k0["x"] - 2 + rng.normal(0, 1.0, len(k0)).astype(np.float32)
    q["v"] = k0["y"] - 1 + rng.normal(0, 1.0, len(k0)).astype(np.float32)
    q["invz"] = rng.uniform(0.2, 1.5, len(k0)).astype(np.float32)
    q["octave"], q["angle"], q["desc"] = k0["octave"], k0["angle"], d0
    q["has_obs"] = rng.random(len(k0)) < 0.5
    m0 = np.full(len(k1), -1, np.int32)
    n_o, m_o = hb.search_frame("oracle", view, q, m0, sf, 40.0, 12.0, fwd, bwd)
    n_p, m_p = py_search_frame(k1, d1, ur, q, m0, sf, 40.0, 12.0, fwd, bwd)
    assert n_o == n_p and np.array_equal(m_o, m_p)
    assert n_o > 20


def py_search_kf(kps, desc, q, match, sf, th, orb_dist):
    """Independent restatement of ORBmatcher.cc:1796-1860."""
    grid = PyGrid(kps)
    match = match.copy()
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)
    for i, p in enumerate(q):
        lvl = int(p["level"])
        cand = grid.area(p["u"], p["v"], F32(th) * F32(sf[lvl]), lvl - 1, lvl + 1)
        best, bi = 256, -1
        for i2 in cand:
            if match[i2] != -1:
                continue
            d = _dist(p["desc"], desc[i2])
            if d < best:
                best, bi = d, i2
        if cand and best <= orb_dist:
            match[bi] = i
            n += 1
            rot = F32(p["angle"]) - F32(kps[bi]["angle"])
            if rot < 0:
                rot = F32(rot + F32(360.0))
            b = _round_half_away(F32(rot * factor))
            hist[0 if b == 30 else b].append(bi)
    i1, i2, i3 = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (i1, i2, i3):
            for j in hist[b]:
                match[j] = -1
                n -= 1
    return n, match


def test_search_by_projection_kf_vs_python(frames):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(8)
    view, keep = hb.frame_view(k1, d1)
    q = np.zeros(len(k0), hb.KF_QUERY)
    q["u"] = k0["x"] - 2 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["v"] = k0["y"] - 1 + rng.normal(0, 1.5, len(k0)).astype(np.float32)
    q["level"], q["angle"], q["desc"] = k0["octave"], k0["angle"], d0
    m0 = np.where(rng.random(len(k1)) < 0.15, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    for orb_dist in (64, 100):
        n_o, m_o = hb.search_kf("oracle", view, q, m0, sf, 10.0, orb_dist)
        n_p, m_p = py_search_kf(k1, d1, q, m0, sf, 10.0, orb_dist)
        assert n_o == n_p and np.array_equal(m_o, m_p)
        assert n_o > 20 and (m_o[m0 == hb.MATCH_TAKEN] == hb.MATCH_TAKEN).all()


def _fake_vocabulary(desc, bits=7):
    """A stand-in for DBoW2 node ids: the first `bits` descriptor bits (close descriptors often share a node)."""
    return (desc[:, 0].astype(np.uint32) >> (8 - bits)) * 3 + 5


def _nodes(desc, order_rng):
    ids = _fake_vocabulary(desc)
    nodes = {}
    for i in order_rng.permutation(len(desc)):      # FeatureVector lists are in insertion order, not sorted
        nodes.setdefault(int(ids[i]), []).append(int(i))
    return nodes


def py_search_bow(kkf, dkf, nkf, has_point, kf_, df, nf, ratio):
    """Independent restatement of ORBmatcher.cc:230-382 over dict-shaped feature vectors."""
    match = np.full(len(kf_), -1, np.int32)
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)
    for nid in sorted(set(nkf) & set(nf)):
        for ikf in nkf[nid]:
            if not has_point[ikf]:
                continue
            b1, b2, bi = 256, 256, -1
            for i_f in nf[nid]:
                if match[i_f] >= 0:
                    continue
                d = _dist(dkf[ikf], df[i_f])
                if d < b1:
                    b2, b1, bi = b1, d, i_f
                elif d < b2:
                    b2 = d
            if b1 <= 50 and F32(b1) < F32(ratio) * F32(b2):
                match[bi] = ikf
                rot = F32(kkf[ikf]["angle"]) - F32(kf_[bi]["angle"])
                if rot < 0:
                    rot = F32(rot + F32(360.0))
                b = _round_half_away(F32(rot * factor))
                hist[0 if b == 30 else b].append(bi)
                n += 1
    i1, i2, i3 = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (i1, i2, i3):
            for j in hist[b]:
                match[j] = -1
                n -= 1
    return n, match


def test_search_by_bow_vs_python(frames):
    k0, d0, k1, d1, _ = frames
    rng = np.random.default_rng(9)
    has = (rng.random(len(k0)) < 0.8).astype(np.uint8)
    n0, n1 = _nodes(d0, rng), _nodes(d1, rng)
    for drop in list(n1)[::7]:                       # nodes present on one side only exercise the lower_bound branches
        del n1[drop]
    for drop in list(n0)[3::9]:
        del n0[drop]
    vkf, keep0 = hb.bow_view(k0, d0, n0, has)
    vf, keep1 = hb.bow_view(k1, d1, n1)
    for ratio in (0.7, 0.9):
        n_o, m_o = hb.search_bow("oracle", vkf, vf, ratio)
        n_p, m_p = py_search_bow(k0, d0, n0, has, k1, d1, n1, ratio)
        assert n_o == n_p and np.array_equal(m_o, m_p)
    assert n_o > 10


def py_search_bow_kf(k1, d1, n1, has1, k2, d2, n2, has2, ratio):
    """Independent restatement of ORBmatcher.cc:656-808."""
    m12 = np.full(len(k1), -1, np.int32)
    matched2 = np.zeros(len(k2), bool)
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)
    for nid in sorted(set(n1) & set(n2)):
        for i1 in n1[nid]:
            if not has1[i1]:
                continue
            b1, b2, bi = 256, 256, -1
            for i2 in n2[nid]:
                if matched2[i2] or not has2[i2]:
                    continue
                d = _dist(d1[i1], d2[i2])
                if d < b1:
                    b2, b1, bi = b1, d, i2
                elif d < b2:
                    b2 = d
            if b1 < 50 and F32(b1) < F32(ratio) * F32(b2):
                m12[i1] = bi
                matched2[bi] = True
                rot = F32(k1[i1]["angle"]) - F32(k2[bi]["angle"])
                if rot < 0:
                    rot = F32(rot + F32(360.0))
                b = _round_half_away(F32(rot * factor))
                hist[0 if b == 30 else b].append(i1)
                n += 1
    i1_, i2_, i3_ = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (i1_, i2_, i3_):
            for j in hist[b]:
                m12[j] = -1
                n -= 1
    return n, m12


def test_search_by_bow_keyframes_vs_python(frames):
    k0, d0, k1, d1, _ = frames
    rng = np.random.default_rng(10)
    has0, has1 = (rng.random(len(k0)) < 0.8).astype(np.uint8), (rng.random(len(k1)) < 0.8).astype(np.uint8)
    n0, n1 = _nodes(d0, rng), _nodes(d1, rng)
    for drop in list(n1)[::5]:
        del n1[drop]
    v0, keep0 = hb.bow_view(k0, d0, n0, has0)
    v1, keep1 = hb.bow_view(k1, d1, n1, has1)
    for ratio in (0.75, 0.95):
        n_o, m_o = hb.search_bow_kf("oracle", v0, v1, ratio)
        n_p, m_p = py_search_bow_kf(k0, d0, n0, has0, k1, d1, n1, has1, ratio)
        assert n_o == n_p and np.array_equal(m_o, m_p)
    assert n_o > 10


def py_search_triangulation(k1, d1, n1, has1, ur1, k2, d2, n2, has2, ur2, f12, ex, ey, sf2, sg2, only_stereo):
    """Independent restatement of ORBmatcher.cc:810-1018 + :188-215 in float32 steps."""
    F = [F32(v) for v in np.asarray(f12, np.float32).reshape(9)]
    m12 = np.full(len(k1), -1, np.int32)
    matched2 = np.zeros(len(k2), bool)
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)

    def epi(kp1, kp2):
        x1, y1, x2, y2 = F32(kp1["x"]), F32(kp1["y"]), F32(kp2["x"]), F32(kp2["y"])
        a = F32(F32(F32(x1 * F[0]) + F32(y1 * F[3])) + F[6])
        b = F32(F32(F32(x1 * F[1]) + F32(y1 * F[4])) + F[7])
        c = F32(F32(F32(x1 * F[2]) + F32(y1 * F[5])) + F[8])
        num = F32(F32(F32(a * x2) + F32(b * y2)) + c)
        den = F32(F32(a * a) + F32(b * b))
        if den == 0:
            return False
        dsqr = F32(F32(num * num) / den)
        return float(dsqr) < 3.84 * float(F32(sg2[int(kp2["octave"])]))
    for nid in sorted(set(n1) & set(n2)):
        for i1 in n1[nid]:
            if has1[i1]:
                continue
            st1 = ur1[i1] >= 0
            if only_stereo and not st1:
                continue
            best, bi = 50, -1
            for i2 in n2[nid]:
                if matched2[i2] or has2[i2]:
                    continue
                st2 = ur2[i2] >= 0
                if only_stereo and not st2:
                    continue
                d = _dist(d1[i1], d2[i2])
                if d > 50 or d > best:
                    continue
                if not st1 and not st2:
                    dx, dy = F32(F32(ex) - F32(k2[i2]["x"])), F32(F32(ey) - F32(k2[i2]["y"]))
                    if F32(F32(dx * dx) + F32(dy * dy)) < F32(F32(100) * F32(sf2[int(k2[i2]["octave"])])):
                        continue
                if epi(k1[i1], k2[i2]):
                    bi, best = i2, d
            if bi >= 0:
                m12[i1] = bi
                matched2[bi] = True
                n += 1
                rot = F32(k1[i1]["angle"]) - F32(k2[bi]["angle"])
                if rot < 0:
                    rot = F32(rot + F32(360.0))
                b = _round_half_away(F32(rot * factor))
                hist[0 if b == 30 else b].append(i1)
    a1, a2, a3 = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (a1, a2, a3):
            for j in hist[b]:
                m12[j] = -1
                n -= 1
    return n, np.array([(i, m12[i]) for i in range(len(k1)) if m12[i] >= 0], np.int32).reshape(-1, 2)


def translation_fundamental(tx=2.0, ty=1.0):
    """F12 of a pure image translation (x2 = x1 - tx, y2 = y1 - ty): epipolar lines through (x1,y1)-(tx,ty) along (tx,ty)."""
    # line through p2 = p1 - t with direction t: normal n = (-ty, tx); a = n.x, b = n.y, c = -(n . (p1 - t))
    # [a b c] = [x1 y1 1] * F12  with  F12 = [[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]]
    return np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32)


@pytest.mark.parametrize("only_stereo", [0, 1])
def test_search_for_triangulation_vs_python(frames, only_stereo):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(12 + only_stereo)
    has0, has1 = (rng.random(len(k0)) < 0.3).astype(np.uint8), (rng.random(len(k1)) < 0.3).astype(np.uint8)
    ur0 = np.where(rng.random(len(k0)) < 0.5, k0["x"] - 10, -1).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 10, -1).astype(np.float32)
    n0, n1 = _nodes(d0, rng), _nodes(d1, rng)
    for drop in list(n0)[::6]:
        del n0[drop]
    sg = (sf * sf).astype(np.float32)
    f12 = translation_fundamental()
    ex, ey = 320.0, 200.0
    v0, keep0 = hb.bow_view(k0, d0, n0, has0, ur0)
    v1, keep1 = hb.bow_view(k1, d1, n1, has1, ur1)
    n_o, p_o = hb.search_triangulation("oracle", v0, v1, f12, ex, ey, sf, sg, only_stereo)
    n_p, p_p = py_search_triangulation(k0, d0, n0, has0, ur0, k1, d1, n1, has1, ur1, f12, ex, ey, sf, sg, only_stereo)
    assert n_o == n_p and np.array_equal(p_o, p_p)
    assert n_o > 5


def py_window_search(kps, desc, ur_kf, q, sf, th, max_dist, inv_sigma2=None, occupied=None):
    """Independent restatement of the candidate loop of Fuse / SearchByProjection(KF,Scw) / SearchBySim3."""
    grid = PyGrid(kps)
    occ = None if occupied is None else occupied.copy()
    best = np.full(len(q), -1, np.int32)
    n = 0
    for i, p in enumerate(q):
        lvl = int(p["level"])
        u, v, ur = F32(p["u"]), F32(p["v"]), F32(p["ur"])
        b, bi = 256, -1
        for idx in grid.area(u, v, F32(th) * F32(sf[lvl]), -1, -1):
            if occ is not None and occ[idx] != -1:
                continue
            kl = int(kps[idx]["octave"])
            if kl < lvl - 1 or kl > lvl:
                continue
            if inv_sigma2 is not None:
                ex, ey = F32(u - F32(kps[idx]["x"])), F32(v - F32(kps[idx]["y"]))
                if ur_kf is not None and ur_kf[idx] >= 0:
                    er = F32(ur - F32(ur_kf[idx]))
                    e2 = F32(F32(F32(ex * ex) + F32(ey * ey)) + F32(er * er))
                    if float(F32(e2 * F32(inv_sigma2[kl]))) > 7.8:
                        continue
                else:
                    e2 = F32(F32(ex * ex) + F32(ey * ey))
                    if float(F32(e2 * F32(inv_sigma2[kl]))) > 5.99:
                        continue
            d = _dist(p["desc"], desc[idx])
            if d < b:
                b, bi = d, idx
        if b <= max_dist:
            best[i] = bi
            n += 1
            if occ is not None:
                occ[bi] = i
    return n, best, occ


def _window_queries(k_src, d_src, rng, jitter=1.5):
    q = np.zeros(len(k_src), hb.WINDOW_QUERY)
    q["u"] = k_src["x"] - 2 + rng.normal(0, jitter, len(k_src)).astype(np.float32)
    q["v"] = k_src["y"] - 1 + rng.normal(0, jitter, len(k_src)).astype(np.float32)
    q["ur"] = q["u"] - rng.uniform(5, 30, len(k_src)).astype(np.float32)
    q["level"] = np.minimum(k_src["octave"] + (rng.random(len(k_src)) < 0.3), 3)
    q["src"], q["desc"] = np.arange(len(k_src)), d_src
    return q


def test_fuse_searches_vs_python(frames):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(14)
    ur1 = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(5, 30, len(k1)), -1).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur1)
    q = _window_queries(k0, d0, rng)
    inv_s2 = (1.0 / (sf * sf)).astype(np.float32)
    n_o, b_o = hb.fuse("oracle", view, q, sf, 3.0, inv_s2)          # Fuse(pKF, vpMapPoints, th = 3)
    n_p, b_p, _ = py_window_search(k1, d1, ur1, q, sf, 3.0, 50, inv_s2)
    assert n_o == n_p and np.array_equal(b_o, b_p) and n_o > 20
    n_o2, b_o2 = hb.fuse("oracle", view, q, sf, 4.0)                # Fuse(pKF, Scw, vpPoints, th = 4)
    n_p2, b_p2, _ = py_window_search(k1, d1, ur1, q, sf, 4.0, 50)
    assert n_o2 == n_p2 and np.array_equal(b_o2, b_p2) and n_o2 >= n_o


def test_search_by_projection_sim_vs_python(frames):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(15)
    view, keep = hb.frame_view(k1, d1)
    q = _window_queries(k0, d0, rng)
    m0 = np.where(rng.random(len(k1)) < 0.2, hb.MATCH_TAKEN, hb.MATCH_FREE).astype(np.int32)
    n_o, m_o = hb.search_projection_sim("oracle", view, q, m0, sf, 10)
    n_p, _, m_p = py_window_search(k1, d1, None, q, sf, 10.0, 50, occupied=m0)
    assert n_o == n_p and np.array_equal(m_o, m_p) and n_o > 20


def test_search_by_sim3_vs_python(frames):
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(16)
    v0, keep0 = hb.frame_view(k0, d0)
    v1, keep1 = hb.frame_view(k1, d1)
    q12 = _window_queries(k0, d0, rng)[rng.random(len(k0)) < 0.8]
    q21 = _window_queries(k1, d1, rng)
    q21["u"] += 4
    q21["v"] += 2                                                     # frame 1 -> frame 0 is the opposite shift
    q21 = q21[rng.random(len(k1)) < 0.8]
    n_o, m_o = hb.search_sim3("oracle", v0, v1, q12, q21, sf, sf, 7.5)
    _, b12, _ = py_window_search(k1, d1, None, q12, sf, 7.5, 100)
    _, b21, _ = py_window_search(k0, d0, None, q21, sf, 7.5, 100)
    m1, m2 = np.full(len(k0), -1), np.full(len(k1), -1)
    m1[q12["src"][b12 >= 0]] = b12[b12 >= 0]
    m2[q21["src"][b21 >= 0]] = b21[b21 >= 0]
    want = np.array([m1[i] if m1[i] >= 0 and m2[m1[i]] == i else -1 for i in range(len(k0))], np.int32)
    assert np.array_equal(m_o, want) and n_o == (want >= 0).sum() and n_o > 20


def py_search_points(kps, desc, ur, q, match, obs, sf, th, ratio):
    """Independent restatement of ORBmatcher::SearchByProjection(F, vpMapPoints, th), ORBmatcher.cc:70-175."""
    grid = PyGrid(kps)
    match, obs = match.copy(), obs.copy()
    n = 0
    for i, p in enumerate(q):
        lvl = int(p["level"])
        r = F32(2.5) if F32(p["view_cos"]) > 0.998 else F32(4.0)
        if th != 1.0:
            r = F32(r * F32(th))
        radius = F32(r * F32(sf[lvl]))
        cand = grid.area(p["proj_x"], p["proj_y"], radius, lvl - 1, lvl)
        b1 = b2 = 256
        l1 = l2 = bi = -1
        for idx in cand:
            if match[idx] >= 0 and obs[idx]:
                continue
            if ur is not None and ur[idx] > 0:
                if abs(F32(F32(p["proj_xr"]) - F32(ur[idx]))) > radius:
                    continue
            d = _dist(p["desc"], desc[idx])
            if d < b1:
                b2, b1, l2, l1, bi = b1, d, l1, int(kps[idx]["octave"]), idx
            elif d < b2:
                l2, b2 = int(kps[idx]["octave"]), d
        if cand and b1 <= 100:
            if l1 == l2 and F32(b1) > F32(F32(ratio) * F32(b2)):
                continue
            match[bi] = i
            obs[bi] = 1 if p["has_obs"] else 0
            n += 1
    return n, match, obs


def test_search_by_projection_points_vs_python(frames):
    """Same-level ratio rule (:163-167), occupied-with-observations skip (:116-118), right-coordinate gate (:121-127)."""
    k0, d0, k1, d1, sf = frames
    rng = np.random.default_rng(21)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 20, -1).astype(np.float32)
    view, keep = hb.frame_view(k1, d1, ur)
    q = np.zeros(len(k0), hb.MAP_QUERY)
    q["proj_x"], q["proj_y"] = k0["x"] - 2, k0["y"] - 1
    q["proj_xr"] = q["proj_x"] - 20 + rng.normal(0, 3, len(k0)).astype(np.float32)
    q["view_cos"] = rng.choice([0.9, 0.999], len(k0)).astype(np.float32)
    q["level"] = np.maximum(k0["octave"], 0)
    q["has_obs"] = rng.random(len(k0)) < 0.5
    q["desc"] = d0
    # pre-existing matches: some with observations (never replaced), some without (may be replaced)
    match0 = np.where(rng.random(len(k1)) < 0.2, 0, -1).astype(np.int32)
    obs0 = ((match0 >= 0) & (rng.random(len(k1)) < 0.5)).astype(np.uint8)
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.95)):
        n_o, m_o, o_o = hb.search_points("oracle", view, q, match0, obs0, sf, th, ratio)
        n_p, m_p, o_p = py_search_points(k1, d1, ur, q, match0, obs0, sf, th, ratio)
        assert n_o == n_p and np.array_equal(m_o, m_p) and np.array_equal(o_o, o_p)
    assert n_o > 20


def py_search_init(k1, d1, k2, d2, prev, window, ratio):
    """Independent restatement of ORBmatcher::SearchForInitialization, ORBmatcher.cc:515-643."""
    grid = PyGrid(k2)
    m12 = np.full(len(k1), -1, np.int32)
    m21 = np.full(len(k2), -1, np.int32)
    mdist = np.full(len(k2), 2 ** 31 - 1, np.int64)
    prev = prev.copy()
    n = 0
    hist = [[] for _ in range(30)]
    factor = F32(30) / F32(360.0)
    for i1 in range(len(k1)):
        if k1[i1]["octave"] > 0:
            continue
        cand = grid.area(prev[i1, 0], prev[i1, 1], F32(window), 0, 0)
        b1 = b2 = 2 ** 31 - 1
        bi = -1
        for i2 in cand:
            d = _dist(d1[i1], d2[i2])
            if mdist[i2] <= d:
                continue
            if d < b1:
                b2, b1, bi = b1, d, i2
            elif d < b2:
                b2 = d
        if cand and b1 <= 50 and F32(b1) < F32(F32(b2) * F32(ratio)):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1
                n -= 1
            m12[i1], m21[bi], mdist[bi] = bi, i1, b1
            n += 1
            rot = F32(k1[i1]["angle"]) - F32(k2[bi]["angle"])
            if rot < 0:
                rot = F32(rot + F32(360.0))
            b = _round_half_away(F32(rot * factor))
            hist[0 if b == 30 else b].append(i1)
    a1, a2, a3 = _three_maxima([len(h) for h in hist])
    for b in range(30):
        if b not in (a1, a2, a3):
            for i1 in hist[b]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    n -= 1
    for i1 in range(len(k1)):
        if m12[i1] >= 0:
            prev[i1] = (k2[m12[i1]]["x"], k2[m12[i1]]["y"])
    return n, m12, prev


def test_search_for_initialization_vs_python(frames):
    """Level-0 only, the vMatchedDistance back-check (:554-555), displaced earlier matches (:577-581)."""
    k0, d0, k1, d1, _ = frames
    v0, keep0 = hb.frame_view(k0, d0)
    v1, keep1 = hb.frame_view(k1, d1)
    prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
    for window, ratio in ((100, 0.9), (30, 0.7)):
        n_o, m_o, p_o = hb.search_init("oracle", v0, v1, prev, window, ratio)
        n_p, m_p, p_p = py_search_init(k0, d0, k1, d1, prev, window, ratio)
        assert n_o == n_p and np.array_equal(m_o, m_p) and np.array_equal(p_o.reshape(-1, 2), p_p)
    assert n_o > 10
