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
