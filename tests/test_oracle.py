"""CPU: the oracle against (a) hand-computable cases, (b) independent numpy / pure-Python
restatements of the same published algorithms written separately from the C code, (c) the host
libm for sincosf, and (d) the committed golden fixtures.

PARITY STATUS (see oracle/orb_oracle.h): the reference holds no golden vectors for this path and
OpenCV is not available, so stages that restate OpenCV primitives are "parity unpinned"; the checks
below pin the oracle to the algorithm descriptions in SURVEY.md Appendix A, not to OpenCV outputs.
"""
import ctypes
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# ------------------------------------------------------------------------------------ resize


def _np_resize(src, dw, dh):
    """cv::resize INTER_LINEAR 8UC1, written independently with numpy."""
    sh, sw = src.shape

    def taps(sn, dn, clamp_fraction):
        scale = 1.0 / (np.float64(dn) / np.float64(sn))
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_fraction:
            lo = s < 0
            f[lo], s[lo] = 0, 0
            hi = s >= sn - 1
            f[hi], s[hi] = 0, sn - 1
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        s0 = np.clip(s, 0, sn - 1)
        s1 = np.clip(s + 1, 0, sn - 1)
        return s0, s1, a0, a1

    x0, x1, a0, a1 = taps(sw, dw, True)
    y0, y1, b0, b1 = taps(sh, dh, False)
    S = src.astype(np.int64)
    H0 = S[y0][:, x0] * a0 + S[y0][:, x1] * a1
    H1 = S[y1][:, x0] * a0 + S[y1][:, x1] * a1
    out = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 533, 400), (533, 400, 444, 333), (97, 61, 81, 51), (50, 40, 50, 40), (40, 30, 64, 48)])
def test_resize_matches_numpy_restatement(ob, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    got = ob.resize_linear_u8(src, dw, dh)
    assert np.array_equal(got, _np_resize(src, dw, dh))


def test_resize_properties(ob):
    const = np.full((48, 64), 137, np.uint8)
    assert (ob.resize_linear_u8(const, 53, 40) == 137).all()  # weights sum to 2048 exactly
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(ob.resize_linear_u8(src, 64, 48), src)  # same size: fx = fy = 0
    ramp = np.tile(np.arange(0, 240, 4, dtype=np.uint8), (10, 1))  # linear ramp stays monotone
    out = ob.resize_linear_u8(ramp, 50, 8)
    assert (np.diff(out.astype(int), axis=1) >= 0).all()


def test_pyramid_border_is_reflect101(ob, synth):
    img = synth.frame(0, 0)
    orc = ob.Oracle()
    orc.detect(img)
    for l in (0, 3, 7):
        inner = orc.level_image(l)
        padded = orc.level_image(l, padded=True)
        assert np.array_equal(padded, np.pad(inner, 19, mode="reflect"))  # numpy 'reflect' == REFLECT_101
    assert np.array_equal(orc.level_image(0), img)


# ------------------------------------------------------------------------------------ FAST

_CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
           (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _py_fast(img, t):
    """FAST-9/16 + score + 3x3 strict NMS straight from the definition."""
    h, w = img.shape
    score = np.zeros((h, w), np.int64)
    I = img.astype(np.int64)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = I[y, x]
            d = [v - I[y + dy, x + dx] for dx, dy in _CIRCLE]
            best = None
            for sign in (1, -1):
                for k in range(16):
                    m = min(sign * d[(k + j) % 16] for j in range(9))
                    if m > t:
                        best = m if best is None else max(best, m)
            if best is not None:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0:
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 20), (3, 40)])
def test_fast_matches_definition(ob, seed, t):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (10, 12)).astype(np.float32)
    img = np.kron(base, np.ones((4, 4), np.float32))[:37, :43]
    img = np.clip(img + rng.integers(-8, 9, img.shape), 0, 255).astype(np.uint8)
    got = ob.fast9_16(img, t)
    want = _py_fast(img, t)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == want
    assert len(want) > 0
    assert (got["size"] == 7).all() and (got["angle"] == -1).all() and (got["class_id"] == -1).all()


def test_fast_small_and_flat(ob):
    assert len(ob.fast9_16(np.full((30, 30), 9, np.uint8), 20)) == 0
    assert len(ob.fast9_16(np.zeros((6, 40), np.uint8), 20)) == 0  # fewer than 7 rows: nothing tested
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255  # isolated bright dot: all 16 circle pixels darker => corner, score 254
    got = ob.fast9_16(img, 20)
    assert len(got) == 1 and (got[0]["x"], got[0]["y"], got[0]["response"]) == (10, 10, 254)


def test_cell_fallback_threshold(ob):
    """ORBextractor.cc:1126-1139: a cell retries with minThFAST only when it came back empty."""
    img = np.full((480, 640), 100, np.uint8)
    img[40:440:8, 300:340:8] = 112  # isolated weak dots: corners at t=7 (diff 12), none at t=20
    orc = ob.Oracle()
    orc.detect(img)
    cand = orc.level_candidates(0)
    assert len(cand) > 0 and cand["response"].max() < 20


# ------------------------------------------------------------------------------------ blur / angle / sincos


def test_blur_matches_numpy(ob):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (45, 57), dtype=np.uint8)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    assert k.sum() == 256
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    hor = sum(k[j] * p[:, j:j + 57] for j in range(7))
    acc = sum(k[i] * hor[i:i + 45, :] for i in range(7))
    want = ((acc + 32768) >> 16).astype(np.uint8)
    assert np.array_equal(ob.gaussian_blur7(img), want)
    assert (ob.gaussian_blur7(np.full((20, 20), 201, np.uint8)) == 201).all()


def test_fast_atan2(ob):
    assert ob.fast_atan2(0.0, 0.0) == 0.0
    assert ob.fast_atan2(0.0, 5.0) == 0.0
    assert abs(ob.fast_atan2(5.0, 0.0) - 90.0) < 1e-4
    assert abs(ob.fast_atan2(0.0, -5.0) - 180.0) < 1e-4
    assert abs(ob.fast_atan2(-5.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(5)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-100000, 100000, 2))
        if x == 0 and y == 0:
            continue
        a = ob.fast_atan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        assert 0.0 <= a <= 360.0
        assert min(abs(a - ref), 360 - abs(a - ref)) < 0.02  # polynomial's documented accuracy


def _libm_sincosf():
    libm = ctypes.CDLL("libm.so.6")
    libm.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    libm.sincosf.restype = None

    def f(x):
        s, c = ctypes.c_float(), ctypes.c_float()
        libm.sincosf(ctypes.c_float(x), ctypes.byref(s), ctypes.byref(c))
        return s.value, c.value
    return f


def test_sincosf_equals_host_libm_on_samples(ob):
    """The oracle's (and the GPU's) sincosf restates glibc's algorithm; on this host they agree bit
    for bit.  The exhaustive sweep over all 1.09e9 floats in [0, 6.2832] is tools/sincosf_sweep.c
    (0 mismatches, recorded in DESIGN.md)."""
    libm = _libm_sincosf()
    rng = np.random.default_rng(6)
    xs = np.concatenate([rng.uniform(0, 6.2832, 20000), np.linspace(0, 6.2832, 5000),
                         np.arange(0, 360, 0.25) * (math.pi / 180.0), [0.0, 1e-5, 2.0 ** -13, math.pi / 4, math.pi / 2]])
    for x in xs.astype(np.float32):
        assert ob.sincosf(float(x)) == libm(float(x)), float(x)


# ------------------------------------------------------------------------------------ quad-tree


def _py_octree(pts, minX, maxX, minY, maxY, N):
    """DistributeOctTree (ORBextractor.cc:706-1049) with Python lists, written from the reference's
    control flow; (size, address) ties use creation order."""
    f32 = np.float32
    nIni = int(np.round(f32(maxX - minX) / f32(maxY - minY)))  # std::round: half away from zero
    nIni = int(math.floor(float(f32(maxX - minX) / f32(maxY - minY)) + 0.5))
    hX = f32(maxX - minX) / f32(nIni)
    seq = [0]

    def node(x0, y0, x1, y1):
        seq[0] += 1
        return {"b": (x0, y0, x1, y1), "k": [], "nm": False, "seq": seq[0]}

    nodes = [node(int(hX * f32(i)), 0, int(hX * f32(i + 1)), maxY - minY) for i in range(nIni)]
    for i, p in enumerate(pts):
        nodes[int(f32(p["x"]) / hX)]["k"].append(i)
    lst = []
    for n in nodes:
        if len(n["k"]) == 1:
            n["nm"] = True
        if n["k"]:
            lst.append(n)

    def divide(n):
        x0, y0, x1, y1 = n["b"]
        hx = int(math.ceil(float(f32(x1 - x0) / f32(2))))
        hy = int(math.ceil(float(f32(y1 - y0) / f32(2))))
        ch = [node(x0, y0, x0 + hx, y0 + hy), node(x0 + hx, y0, x1, y0 + hy), node(x0, y0 + hy, x0 + hx, y1),
              node(x0 + hx, y0 + hy, x1, y1)]
        for i in n["k"]:
            px, py = pts[i]["x"], pts[i]["y"]
            if px < x0 + hx:
                ch[0 if py < y0 + hy else 2]["k"].append(i)
            else:
                ch[1 if py < y0 + hy else 3]["k"].append(i)
        for c in ch:
            c["nm"] = len(c["k"]) == 1
        return ch

    finish = False
    while not finish:
        prev = len(lst)
        n_expand = 0
        vec = []
        front = []
        keep = []
        for n in lst:
            if n["nm"]:
                keep.append(n)
                continue
            for c in divide(n):
                if c["k"]:
                    front.insert(0, c)
                    if len(c["k"]) > 1:
                        n_expand += 1
                        vec.append(c)
        lst = front + keep
        if len(lst) >= N or len(lst) == prev:
            finish = True
        elif len(lst) + n_expand * 3 > N:
            while not finish:
                prev = len(lst)
                pv = sorted(vec, key=lambda n: (len(n["k"]), n["seq"]))
                vec = []
                for n in reversed(pv):
                    for c in divide(n):
                        if c["k"]:
                            lst.insert(0, c)
                            if len(c["k"]) > 1:
                                vec.append(c)
                    lst.remove(n)
                    if len(lst) >= N:
                        break
                if len(lst) >= N or len(lst) == prev:
                    finish = True
    out = []
    for n in lst:
        best = n["k"][0]
        for i in n["k"][1:]:
            if pts[i]["response"] > pts[best]["response"]:
                best = i
        out.append(best)
    return out


@pytest.mark.parametrize("seed,n,N,w,h", [(0, 400, 60, 300, 200), (1, 1500, 217, 608, 448), (2, 50, 100, 200, 150),
                                         (3, 900, 100, 400, 100), (4, 3, 10, 100, 80), (5, 700, 73, 182, 129)])
def test_octree_matches_python_restatement(ob, seed, n, N, w, h):
    rng = np.random.default_rng(seed)
    cells = rng.choice(w * h, size=n, replace=False)  # distinct integer positions, like FAST output
    cells.sort()
    pts = np.zeros(n, ob.KP_DTYPE)
    pts["x"], pts["y"] = cells % w, cells // w
    pts["response"] = rng.integers(6, 60, n)  # many ties: first-wins rule matters
    got = ob.distribute_octree(pts, 16, 16 + w, 16, 16 + h, N)
    want = _py_octree(pts, 16, 16 + w, 16, 16 + h, N)
    assert [(k["x"], k["y"], k["response"]) for k in got] == [(pts[i]["x"], pts[i]["y"], pts[i]["response"]) for i in want]
    assert len(got) <= max(N, 4 * int(round(w / h))) + 3


def test_octree_edge_cases(ob):
    assert len(ob.distribute_octree(np.zeros(0, ob.KP_DTYPE), 16, 316, 16, 216, 50)) == 0
    one = np.zeros(1, ob.KP_DTYPE)
    one["x"], one["y"], one["response"] = 7, 9, 33
    got = ob.distribute_octree(one, 16, 316, 16, 216, 50)
    assert len(got) == 1 and got[0]["response"] == 33


# ------------------------------------------------------------------------------------ descriptors


def test_descriptor_definition(ob, synth):
    """rBRIEF bit k of byte i compares pattern points 16i+2k and 16i+2k+1 of the blurred level,
    rotated by the keypoint angle (ORBextractor.cc:173-227)."""
    import re
    img = synth.frame(9, 0)
    orc = ob.Oracle()
    kps, desc = orc.extract(img)
    hdr = open(os.path.join(ROOT, "include", "amos_orb_pattern.h")).read()
    body = hdr[hdr.index("amos_orb_pattern[256 * 4] = {"):]
    pat = np.array([int(v) for v in re.findall(r"-?\d+", body[body.index("{") + 1:body.index("};")])]).reshape(256, 4)
    libm = _libm_sincosf()
    f32 = np.float32
    offset = 0
    for level in range(8):
        lk = orc.level_keypoints(level)
        blur = orc.blurred_image(level)
        for i in range(0, len(lk), 17):
            kp = lk[i]
            ang = f32(kp["angle"]) * f32(math.pi / f32(180.0))
            b, a = (f32(v) for v in libm(float(ang)))
            cy, cx = int(np.rint(kp["y"])), int(np.rint(kp["x"]))
            bits = []
            for x0, y0, x1, y1 in pat:
                vals = []
                for px, py in ((x0, y0), (x1, y1)):
                    # fma(px, b, py*a): exact product in float64, one rounding of the sum to float32
                    r = int(np.rint(f32(np.float64(f32(px)) * np.float64(b) + np.float64(f32(py) * a))))
                    c = int(np.rint(f32(np.float64(f32(px)) * np.float64(a) - np.float64(f32(py) * b))))
                    vals.append(int(blur[cy + r, cx + c]))
                bits.append(vals[0] < vals[1])
            want = np.packbits(np.array(bits, np.uint8), bitorder="little")
            assert np.array_equal(desc[offset + i], want), (level, i)
        offset += len(lk)
    assert offset == len(kps)
    sc = orc.tables()["scale"]
    lk3 = orc.level_keypoints(3)
    n_before = sum(len(orc.level_keypoints(l)) for l in range(3))
    assert kps[n_before]["x"] == f32(lk3[0]["x"]) * sc[3]  # pt *= scale for level != 0


def test_orientation_definition(ob, synth):
    img = synth.frame(10, 0)
    orc = ob.Oracle()
    orc.detect(img)
    umax = orc.tables()["umax"]
    for level in (0, 4):
        lv = orc.level_image(level).astype(np.int64)
        for kp in orc.level_keypoints(level)[::23]:
            cx, cy = int(kp["x"]), int(kp["y"])
            m10 = m01 = 0
            for v in range(-15, 16):
                d = int(umax[abs(v)])
                row = lv[cy + v, cx - d:cx + d + 1]
                m10 += int((np.arange(-d, d + 1) * row).sum())
                m01 += v * int(row.sum())
            assert kp["angle"] == np.float32(ob.fast_atan2(float(m01), float(m10)))
            assert kp["octave"] == level and kp["class_id"] == -1
            assert kp["size"] == np.float32(int(np.float32(31) * orc.tables()["scale"][level]))


# ------------------------------------------------------------------------------------ morphology / gate


def test_closing_definition(ob):
    rng = np.random.default_rng(8)
    mask = np.zeros((70, 90), np.uint8)
    mask[20:45, 30:60] = 255
    mask[30:34, 40:44] = 0  # hole: closed by the 31x31 element
    mask[rng.integers(0, 70, 5), rng.integers(0, 90, 5)] = 255
    r = 15
    dx = [int(np.rint(15 * math.sqrt((r * r - (i - r) ** 2) / (r * r)))) for i in range(31)]
    assert dx[0] == 0 and dx[15] == 15 and dx[1] == 5

    def morph(src, fn, neutral):
        out = np.zeros_like(src)
        h, w = src.shape
        for y in range(h):
            for x in range(w):
                acc = neutral
                for i in range(31):
                    yy = y + i - 15
                    if 0 <= yy < h:
                        seg = src[yy, max(x - dx[i], 0):min(x + dx[i] + 1, w)]
                        if seg.size:
                            acc = fn(acc, int(fn(seg)))
                out[y, x] = acc
        return out

    want = morph(morph(mask, max, 0), min, 255)
    got = ob.close_ellipse31(mask)
    assert np.array_equal(got, want)
    assert got[31, 41] == 255  # the hole is gone
    assert (got >= mask).all()  # closing is extensive


def test_gate_semantics(ob, synth):
    img = synth.frame(2, 5)
    mask = synth.person_mask(2, 5)
    orc = ob.Oracle()
    orc.detect(img)
    before = [orc.level_keypoints(l).copy() for l in range(8)]
    removed = orc.gate(mask)
    closed = orc.closed_mask()
    sc = orc.tables()["scale"]
    exp_removed = []
    for l in range(8):
        s = np.float32(1.0) if l == 0 else sc[l]
        kept = []
        for kp in before[l]:
            x, y = int(np.float32(kp["x"]) * s), int(np.float32(kp["y"]) * s)
            (exp_removed if closed[y, x] else kept).append(kp)
        assert [tuple(k) for k in orc.level_keypoints(l)] == [tuple(k) for k in kept]
    assert [tuple(k) for k in removed] == [tuple(k) for k in exp_removed]
    assert len(removed) > 0


# ------------------------------------------------------------------------------------ matcher


def test_descriptor_distance_known_answers(ob):
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert ob.descriptor_distance(z, o) == 256 and ob.descriptor_distance(o, o) == 0
    for b in (0, 7, 8, 100, 255):
        f = z.copy()
        f[b // 8] = 1 << (b % 8)
        assert ob.descriptor_distance(z, f) == 1
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        want = sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))
        assert ob.descriptor_distance(a, b) == want


def _seq_best2(dists, idxs, init):
    best, bi, second, si = init, -1, init, -1
    for d, i in zip(dists, idxs):
        if d < best:
            second, si, best, bi = best, bi, d, i
        elif d < second:
            second, si = d, i
    return bi, best, si, second


def test_best2_tie_rules(ob):
    """Ties: the first candidate wins, a later equal one becomes second best (ORBmatcher.cc:135-147)."""
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    t = np.repeat(q, 6, axis=0)
    t[0, 0] ^= 0b111  # d=3
    t[1, 0] ^= 0b1  # d=1
    t[2, 1] ^= 0b1  # d=1 (tie with 1)
    t[3, 0] ^= 0b11  # d=2
    t[4] = q  # d=0
    t[5] = q  # d=0 (tie)
    for order in ([0, 1, 2, 3], [2, 1, 0], [3, 2, 1], [0, 4, 5], [5, 4], [1], []):
        off, idx = np.array([0, len(order)], np.int32), np.array(order, np.int32)
        r = ob.list_best2(q, t, off, idx, 256)[0]
        d = [ob.descriptor_distance(q[0], t[i]) for i in order]
        assert tuple(r) == _seq_best2(d, order, 256)
    r = ob.list_best2(q, t, np.array([0, 3], np.int32), np.array([0, 3, 1], np.int32), 2)[0]
    assert tuple(r) == (1, 1, -1, 2)  # only distances < init_dist take part


def test_three_maxima(ob):
    """ComputeThreeMaxima incl. the 10 % rule (ORBmatcher.cc:1899-1907)."""
    h = np.zeros(30, np.int32)
    h[[3, 10, 20]] = [50, 30, 20]
    assert ob.three_maxima(h) == (3, 10, 20)
    h[20] = 4  # < 0.1 * 50
    assert ob.three_maxima(h) == (3, 10, -1)
    h[10] = 4
    assert ob.three_maxima(h) == (3, -1, -1)
    h[:] = 0
    assert ob.three_maxima(h) == (-1, -1, -1)
    h[[1, 2]] = 7  # equal bins: the first stays first
    assert ob.three_maxima(h)[:2] == (1, 2)


# ------------------------------------------------------------------------------------ callers either side (8f)


def test_color_to_gray_definition(ob):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    want = ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(ob.color_to_gray(img, rgb_order=False), want)
    assert np.array_equal(ob.color_to_gray(img[..., ::-1], rgb_order=True), want)
    assert np.array_equal(ob.color_to_gray(np.dstack([img, img[..., :1]]), rgb_order=False), want)  # BGRA: alpha ignored
    grey = np.repeat(rng.integers(0, 256, (5, 7, 1), dtype=np.uint8), 3, axis=2)
    assert np.array_equal(ob.color_to_gray(grey), grey[..., 0])  # weights sum to 2^15


def test_rgbd_glue_definition(ob):
    rng = np.random.default_rng(12)
    n, w, h = 200, 640, 480
    kps = np.zeros(n, ob.KP_DTYPE)
    kps["x"], kps["y"] = rng.uniform(0, w - 0.01, n), rng.uniform(0, h - 0.01, n)
    raw = rng.integers(0, 40000, (h, w)).astype(np.uint16)
    raw[rng.random((h, w)) < 0.2] = 0
    factor = np.float32(1.0) / np.float32(5000.0)
    depth = (raw.astype(np.float32) * factor).astype(np.float32)
    assert np.array_equal(ob.depth_convert(raw[:3, :5], float(factor)), depth[:3, :5])
    ur, dep, cell = ob.rgbd_glue(kps, depth, 40.0, (0.0, 640.0, 0.0, 480.0))
    for i in range(n):
        d = depth[int(kps["y"][i]), int(kps["x"][i])]
        if d > 0:
            assert dep[i] == d and ur[i] == np.float32(kps["x"][i]) - np.float32(40.0) / d
        else:
            assert dep[i] == -1 and ur[i] == -1
        px = int(math.floor(abs(float(np.float32(kps["x"][i]) * np.float32(0.1))) + 0.5))
        py = int(math.floor(abs(float(np.float32(kps["y"][i]) * np.float32(0.1))) + 0.5))
        assert cell[i] == (px * 48 + py if px < 64 and py < 48 else -1)


def test_undistort_points_inverts_the_distortion_model(ob):
    """cv::undistortPoints restatement: distorting the result with the forward Brown model returns the input
    (5 fixed-point iterations converge to < 1e-3 px for TUM1-like coefficients), k1 == 0 is the identity."""
    fx, fy, cx, cy = 517.306408, 516.469215, 318.643040, 255.313989         # TUM1.yaml
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    und = ob.undistort_points(pts, fx, fy, cx, cy, dist)
    k1, k2, p1, p2, k3 = (float(v) for v in dist)
    x, y = (und[:, 0].astype(np.float64) - np.float32(cx)) / np.float32(fx), (und[:, 1].astype(np.float64) - np.float32(cy)) / np.float32(fy)
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([xd * np.float32(fx) + np.float32(cx), yd * np.float32(fy) + np.float32(cy)], 1)
    assert np.abs(back - pts).max() < 0.05
    assert np.abs(und - pts).max() > 1.0                                      # the coefficients do something
    assert np.array_equal(ob.undistort_points(pts, fx, fy, cx, cy, np.zeros(5, np.float32)), pts)
    assert np.array_equal(ob.undistort_points(pts, fx, fy, cx, cy, np.zeros(0, np.float32)), pts)
    b = ob.image_bounds(640, 480, fx, fy, cx, cy, dist)
    c = ob.undistort_points(np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32), fx, fy, cx, cy, dist)
    assert b == (min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1]))
    assert ob.image_bounds(640, 480, fx, fy, cx, cy, np.zeros(5, np.float32)) == (0.0, 640.0, 0.0, 480.0)


def _lab_image(rng, h, w):
    """Blocky colour regions + noise, like a Lab image of an indoor scene; depth in millimetre-like units."""
    base = np.kron(rng.integers(0, 256, ((h + 23) // 24, (w + 23) // 24, 3)), np.ones((24, 24, 1)))[:h, :w]
    lab = np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    depth = (5000 * (1.5 + 0.5 * np.sin(np.arange(w) / 40.0))[None, :] * np.ones((h, 1))).astype(np.uint16)
    depth[rng.random((h, w)) < 0.1] = 0
    return lab, depth


def py_slic(lab, depth, length, m, iterations):
    """Independent restatement of cluster::SLIC (cluster.cc:88-343 after cvtColor): Sobel by array slicing, the
    assignment as a per-pixel minimum over (distance, centre index) instead of the sequential overwrite."""
    h, w = depth.shape
    L = lab.astype(np.int64)
    pad = np.pad(L, ((1, 1), (1, 1), (0, 0)), mode="reflect")          # REFLECT_101
    sob_dy = (pad[2:, :-2] + 2 * pad[2:, 1:-1] + pad[2:, 2:]) - (pad[:-2, :-2] + 2 * pad[:-2, 1:-1] + pad[:-2, 2:])
    sob_dx = (pad[:-2, 2:] + 2 * pad[1:-1, 2:] + pad[2:, 2:]) - (pad[:-2, :-2] + 2 * pad[1:-1, :-2] + pad[2:, :-2])
    grad = sob_dy * 0.5 + sob_dx * 0.5
    g2 = (grad ** 2)[..., 0] + (grad ** 2)[..., 1] + (grad ** 2)[..., 2]
    cents = []
    for i in range(0, h, length):
        cy = i + length // 2
        if cy >= h:
            continue
        for j in range(0, w, length):
            cx = j + length // 2
            if cx >= w:
                continue
            cents.append([cx, cy, int(L[cy, cx, 0]), int(L[cy, cx, 1]), int(L[cy, cx, 2]), int(depth[cy, cx]), len(cents) + 1])
    for c in cents:
        cx, cy = c[0], c[1]
        if cx - 1 < 0 or cx + 1 >= w or cy - 1 < 0 or cy + 1 >= h:
            continue
        win = g2[cy - 1:cy + 2, cx - 1:cx + 2]
        k = int(np.argmin(win))                                         # first minimum in row-major (m outer, n inner) order
        c[0], c[1] = cx + k % 3 - 1, cy + k // 3 - 1
        c[2:5] = [int(v) for v in L[c[1], c[0]]]
    labels = np.zeros((h, w))
    for _ in range(iterations):
        best = np.full((h, w), 999999.0)
        owner = np.full((h, w), -1)
        for ck, (cx, cy, cL, cA, cB, cD, lb) in enumerate(cents):
            y0, y1, x0, x1 = max(cy - length, 0), min(cy + length, h), max(cx - length, 0), min(cx + length, w)
            if y0 >= y1 or x0 >= x1:
                continue
            blk = L[y0:y1, x0:x1]
            disc = np.sqrt(((blk[..., 0] - cL) ** 2 + (blk[..., 1] - cA) ** 2 + (blk[..., 2] - cB) ** 2).astype(np.float64))
            yy, xx = np.mgrid[y0:y1, x0:x1]
            diss = np.sqrt(((xx - cx) ** 2 + (yy - cy) ** 2).astype(np.float64))
            dis = np.sqrt(disc * disc + m * (diss * diss))
            take = dis < best[y0:y1, x0:x1]                              # strictly smaller: the earlier centre keeps ties
            best[y0:y1, x0:x1][take] = dis[take]
            owner[y0:y1, x0:x1][take] = ck
        cov = owner >= 0
        labels[cov] = np.array([c[6] for c in cents])[owner[cov]]
        for c in cents:
            cx, cy, lb = c[0], c[1], c[6]
            y0, y1, x0, x1 = max(cy - length, 0), min(cy + length, h), max(cx - length, 0), min(cx + length, w)
            sel = labels[y0:y1, x0:x1] == lb if (y0 < y1 and x0 < x1) else np.zeros((0, 0), bool)
            num = float(sel.sum()) if sel.size else 0.0
            if num == 0:
                num = 0.000000001
            yy, xx = np.mgrid[y0:y1, x0:x1] if (y0 < y1 and x0 < x1) else (np.zeros((0, 0)), np.zeros((0, 0)))
            blk = L[y0:y1, x0:x1]
            c[0] = int(float(xx[sel].sum()) / num)
            c[1] = int(float(yy[sel].sum()) / num)
            c[2] = int(float(blk[..., 0][sel].sum()) / num)
            c[3] = int(float(blk[..., 1][sel].sum()) / num)
            c[4] = int(float(blk[..., 2][sel].sum()) / num)
            c[5] = int(float(depth[y0:y1, x0:x1].astype(np.int64)[sel].sum()) / num)
    return labels, np.array(cents, np.int64)


@pytest.mark.parametrize("h,w,length", [(60, 83, 5), (47, 64, 4), (33, 40, 7)])
def test_slic_matches_python_restatement(ob, h, w, length):
    rng = np.random.default_rng(h * w)
    lab, depth = _lab_image(rng, h, w)
    for iters in (0, 1, 5):
        lo, co = ob.slic(lab, depth, length, 10, iters)
        lp, cp = py_slic(lab, depth, length, 10, iters)
        assert np.array_equal(lo, lp), f"label map, {iters} iterations"
        for k, name in enumerate(("x", "y", "L", "A", "B", "D", "label")):
            assert np.array_equal(co[name], cp[:, k]), (name, iters)
    assert len(np.unique(lo)) > len(co) // 2                               # most centres own pixels


# ------------------------------------------------------------------------------------ golden fixtures


@pytest.mark.parametrize("name", ["c1_640x480_s0k0", "c1_640x480_s3k17", "small_320x240_s1k2"])
def test_golden_fixtures(ob, synth, name):
    """Regression fixtures produced by tools/gen_golden.py FROM THE ORACLE (the reference cannot be
    run here, so these pin the oracle against accidental change -- they are not reference outputs)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    w, h, nf, nl, stream, k = (int(v) for v in g["meta"])
    img = synth.frame(stream, k, h, w)
    import hashlib
    assert hashlib.sha256(img.tobytes()).digest() == g["image_sha256"].tobytes()
    orc = ob.Oracle(n_features=nf, n_levels=nl)
    kps, desc = orc.extract(img)
    assert kps.tobytes() == g["keypoints"].tobytes()
    assert np.array_equal(desc, g["descriptors"])
    assert [len(orc.level_candidates(l)) for l in range(nl)] == list(g["candidates_per_level"])
