"""ctypes binding of oracle/liborb_oracle.so -- the CPU checker.  Used by tests/, smoke() and
bench.py's cpu_baseline leg only; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
BEST2_DTYPE = np.dtype([("best_idx", "<i4"), ("best_dist", "<i4"), ("second_idx", "<i4"),
                        ("second_dist", "<i4")])


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("orb_oracle.c", "lk_oracle.c", "corner_oracle.c", "orb_oracle.h")]
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        if os.path.exists(srcs[0]):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_create.argtypes = [C.POINTER(OrbParams)]
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """Mirror of the C-ABI's extractor entry points on the CPU oracle."""

    def __init__(self, n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.params = OrbParams(n_features, scale_factor, n_levels, ini_th, min_th)
        self.h = C.c_void_p(self.L.orc_create(C.byref(self.params)))
        if not self.h:
            raise ValueError("orc_create failed")
        self.n_levels = n_levels
        self.n_features = n_features
        self.shape = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.n_levels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(fpl), _p(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=umax)

    def level_sizes(self, width, height):
        lw, lh = np.zeros(self.n_levels, np.int32), np.zeros(self.n_levels, np.int32)
        self.L.orc_level_sizes(self.h, C.c_int(width), C.c_int(height), _p(lw), _p(lh))
        return lw, lh

    def detect(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        self.shape = (h, w)
        rc = self.L.orc_detect(self.h, _p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h))
        if rc != 0:
            raise RuntimeError(f"orc_detect rc={rc}")

    def level_keypoints(self, level):
        n = self.L.orc_level_count(self.h, C.c_int(level))
        out = np.zeros(max(n, 1), KP_DTYPE)
        rc = self.L.orc_level_keypoints(self.h, C.c_int(level), _p(out), C.c_int(len(out)))
        assert rc == n, rc
        return out[:n]

    def set_level_keypoints(self, level, kps):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        rc = self.L.orc_set_level_keypoints(self.h, C.c_int(level), _p(kps), C.c_int(len(kps)))
        assert rc == 0, rc

    def level_candidates(self, level, cap=200000):
        out = np.zeros(cap, KP_DTYPE)
        n = self.L.orc_level_candidates(self.h, C.c_int(level), _p(out), C.c_int(cap))
        assert n >= 0, n
        return out[:n]

    def level_image(self, level, padded=False):
        lw, lh = self.level_sizes(self.shape[1], self.shape[0])
        w, h = int(lw[level]), int(lh[level])
        if padded:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        rc = self.L.orc_level_image(self.h, C.c_int(level), _p(out), C.c_size_t(w), C.c_int(int(padded)))
        assert rc == 0, rc
        return out

    def blurred_image(self, level):
        lw, lh = self.level_sizes(self.shape[1], self.shape[0])
        out = np.zeros((int(lh[level]), int(lw[level])), np.uint8)
        rc = self.L.orc_blurred_image(self.h, C.c_int(level), _p(out), C.c_size_t(out.shape[1]))
        assert rc == 0, rc
        return out

    def gate(self, mask, labels=None, center_ids=None, rm_vector=None, cap=100000):
        mask = np.ascontiguousarray(mask, np.uint8)
        removed = np.zeros(cap, KP_DTYPE)
        nrem = C.c_int(0)
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.float64)
            center_ids = np.ascontiguousarray(center_ids, np.int32)
            rm_vector = np.ascontiguousarray(rm_vector, np.int32)
            rc = self.L.orc_gate(self.h, _p(mask), C.c_size_t(mask.strides[0]), _p(labels),
                                 C.c_size_t(labels.shape[1]), _p(center_ids), C.c_int(len(center_ids)),
                                 _p(rm_vector), C.c_int(len(rm_vector)), _p(removed), C.c_int(cap),
                                 C.byref(nrem))
        else:
            rc = self.L.orc_gate(self.h, _p(mask), C.c_size_t(mask.strides[0]), None, C.c_size_t(0), None,
                                 C.c_int(0), None, C.c_int(0), _p(removed), C.c_int(cap), C.byref(nrem))
        if rc != 0:
            raise RuntimeError(f"orc_gate rc={rc}")
        return removed[:nrem.value]

    def closed_mask(self):
        out = np.zeros(self.shape, np.uint8)
        rc = self.L.orc_closed_mask(self.h, _p(out), C.c_size_t(out.shape[1]))
        assert rc == 0, rc
        return out

    def describe(self, cap=None):
        cap = cap or (self.n_features * 4 + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = self.L.orc_describe(self.h, _p(kps), _p(desc), C.c_int(cap), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"orc_describe rc={rc}")
        return kps[:n.value], desc[:n.value]

    def extract(self, gray, cap=None):
        self.detect(gray)
        return self.describe(cap)


# ---- primitives ---------------------------------------------------------------------------

def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), C.c_int(src.shape[1]), C.c_int(src.shape[0]),
                               C.c_size_t(src.strides[0]), _p(dst), C.c_int(dw), C.c_int(dh), C.c_size_t(dw))
    return dst


def fast9_16(img, threshold, cap=100000):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_fast9_16(_p(img), C.c_size_t(img.strides[0]), C.c_int(img.shape[1]), C.c_int(img.shape[0]),
                           C.c_int(threshold), _p(out), C.c_int(cap))
    return out[:n]


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_gaussian_blur7(_p(img), C.c_size_t(img.strides[0]), C.c_int(img.shape[1]), C.c_int(img.shape[0]),
                             _p(out), C.c_size_t(out.strides[0]))
    return out


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def sincosf(x):
    s, c = C.c_float(0), C.c_float(0)
    lib().orc_sincosf(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def distribute_octree(pts, minX, maxX, minY, maxY, N):
    pts = np.ascontiguousarray(pts, KP_DTYPE)
    out = np.zeros(N + 16 + len(pts), KP_DTYPE)
    n = lib().orc_distribute_octree(_p(pts), C.c_int(len(pts)), C.c_int(minX), C.c_int(maxX), C.c_int(minY),
                                    C.c_int(maxY), C.c_int(N), _p(out), C.c_int(len(out)))
    assert n >= 0, n
    return out[:n]


def close_ellipse31(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros_like(mask)
    lib().orc_close_ellipse31(_p(mask), C.c_size_t(mask.strides[0]), C.c_int(mask.shape[1]),
                              C.c_int(mask.shape[0]), _p(out), C.c_size_t(out.strides[0]))
    return out


def descriptor_distance(a, b):
    a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_descriptor_distance(_p(a), _p(b)))


def distances(q, t):
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    out = np.zeros((len(q), len(t)), np.uint16)
    lib().orc_distances(_p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(out))
    return out


def list_distances(q, t, cand_off, cand_idx):
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    cand_off, cand_idx = np.ascontiguousarray(cand_off, np.int32), np.ascontiguousarray(cand_idx, np.int32)
    out = np.zeros(len(cand_idx), np.uint16)
    lib().orc_list_distances(_p(q), C.c_int(len(q)), _p(t), _p(cand_off), _p(cand_idx), _p(out))
    return out


def list_best2(q, t, cand_off, cand_idx, init_dist=256):
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    cand_off, cand_idx = np.ascontiguousarray(cand_off, np.int32), np.ascontiguousarray(cand_idx, np.int32)
    out = np.zeros(len(q), BEST2_DTYPE)
    lib().orc_list_best2(_p(q), C.c_int(len(q)), _p(t), _p(cand_off), _p(cand_idx), C.c_int(init_dist), _p(out))
    return out


def bruteforce_best2(q, t, init_dist=256):
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    out = np.zeros(len(q), BEST2_DTYPE)
    lib().orc_bruteforce_best2(_p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), C.c_int(init_dist), _p(out))
    return out


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orc_three_maxima(_p(sizes), C.c_int(len(sizes)), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def color_to_gray(img, rgb_order=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_color_to_gray(_p(img), C.c_size_t(img.strides[0]), C.c_int(w), C.c_int(h), C.c_int(c), C.c_int(int(rgb_order)), _p(out),
                            C.c_size_t(w))
    return out


def depth_convert(raw_u16, factor):
    f = lib().orc_depth_convert
    f.restype = C.c_float
    f.argtypes = [C.c_uint16, C.c_float]
    return np.array([f(int(v), factor) for v in np.asarray(raw_u16).ravel()], np.float32).reshape(np.shape(raw_u16))


def rgbd_glue(kps, depth_f32, mbf, bounds, kps_un=None):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    kun = None if kps_un is None else np.ascontiguousarray(kps_un, KP_DTYPE)
    depth_f32 = np.ascontiguousarray(depth_f32, np.float32)
    n = len(kps)
    ur, dep, cell = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib().orc_rgbd_glue(_p(kps), _p(kun), C.c_int(n), _p(depth_f32), C.c_size_t(depth_f32.shape[1]), C.c_int(depth_f32.shape[1]),
                        C.c_int(depth_f32.shape[0]), C.c_float(mbf), C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]),
                        C.c_float(bounds[3]), _p(ur), _p(dep), _p(cell))
    return ur, dep, cell


def undistort_points(xy, fx, fy, cx, cy, dist):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(xy)
    lib().orc_undistort_points(_p(xy), C.c_int(len(xy)), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(dist),
                               C.c_int(len(dist)), _p(out))
    return out


def image_bounds(width, height, fx, fy, cx, cy, dist):
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    lib().orc_image_bounds(C.c_int(width), C.c_int(height), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(dist),
                           C.c_int(len(dist)), _p(out))
    return tuple(float(v) for v in out)


SLIC_CENTER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("L", "<i4"), ("A", "<i4"), ("B", "<i4"), ("D", "<i4"), ("label", "<i4"), ("id", "<i4")])


def slic(lab, depth, length=5, m=10, iterations=5):
    """orc_slic: cluster::SLIC from the Lab image on.  Returns (label map float64, centres)."""
    lab = np.ascontiguousarray(lab, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    h, w = depth.shape
    cap = ((h + length - 1) // length) * ((w + length - 1) // length) + 1
    labels = np.zeros((h, w), np.float64)
    centers = np.zeros(cap, SLIC_CENTER_DTYPE)
    n = lib().orc_slic(_p(lab), _p(depth), C.c_int(w), C.c_int(h), C.c_int(length), C.c_int(m), C.c_int(iterations), _p(labels), _p(centers),
                       C.c_int(cap))
    assert 0 < n <= cap
    return labels, centers[:n]


def kmeans(centers, k=15, seed=1, max_iter=1000):
    """orc_kmeans: cluster::randCent + kmeans (cluster.cc:353-460) with the seeded generator.  Returns (centres with id, passes)."""
    c = np.ascontiguousarray(centers, SLIC_CENTER_DTYPE).copy()
    passes = lib().orc_kmeans(_p(c), C.c_int(len(c)), C.c_int(k), C.c_uint32(seed), C.c_int(max_iter))
    return c, passes


def lk_track(prev, nxt, pts, win=22, max_level=5, max_count=20, epsilon=0.01, min_eig=1e-4):
    """orc_lk_track: cv::calcOpticalFlowPyrLK as Tracking::GetSceneFlowObj calls it (Tracking.cc:896).  Returns (next_pts, status, err, top level)."""
    prev, nxt = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    h, w = prev.shape
    out = np.zeros_like(pts)
    status = np.zeros(len(pts), np.uint8)
    err = np.zeros(len(pts), np.float32)
    f = lib().orc_lk_track
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float,
                  C.c_void_p, C.c_void_p, C.c_void_p]
    top = f(_p(prev), prev.strides[0], _p(nxt), nxt.strides[0], w, h, _p(pts), len(pts), win, max_level, max_count, epsilon, min_eig, _p(out), _p(status), _p(err))
    assert top >= 0
    return out, status, err, top


def lk_pyramid_level(gray, level, win=22, max_level=5):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    img = np.zeros((h, w), np.uint8)
    deriv = np.zeros((h, w, 2), np.int16)
    lw, lh = C.c_int(0), C.c_int(0)
    top = lib().orc_lk_pyramid_level(_p(gray), C.c_size_t(gray.strides[0]), w, h, win, max_level, level, _p(img), _p(deriv), C.byref(lw), C.byref(lh))
    assert top >= 0
    return img.reshape(-1)[:lw.value * lh.value].reshape(lh.value, lw.value), deriv.reshape(-1)[:2 * lw.value * lh.value].reshape(lh.value, lw.value, 2), top


def bgr_to_lab(img, rgb_order=False):
    """orc_bgr_to_lab: cv::cvtColor(COLOR_BGR2Lab / RGB2Lab) on 8-bit pixels (OpenCV 4.5's fixed-point path, restated)."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_bgr_to_lab(_p(img), C.c_size_t(img.size // 3), C.c_int(2 if rgb_order else 0), _p(out))
    return out


# ---- oracle/corner_oracle.c: goodFeaturesToTrack (Harris) + cornerSubPix, Tracking.cc:894-895

def corner_harris(gray, k=0.04):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_corner_harris(_p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h), C.c_double(k), _p(out))
    return out


def good_features_to_track(gray, max_corners=1000, quality=0.01, min_distance=8.0, k=0.04, cap=20000, with_response=False):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    xy = np.zeros((cap, 2), np.float32)
    resp = np.zeros((h, w), np.float32) if with_response else None
    n = lib().orc_good_features_to_track(_p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h), C.c_int(max_corners), C.c_double(quality),
                                         C.c_double(min_distance), C.c_double(k), _p(xy), C.c_int(cap), _p(resp))
    return (xy[:n].copy(), resp) if with_response else xy[:n].copy()


def corner_subpix(gray, xy, win=10, max_count=20, epsilon=0.03):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.ascontiguousarray(xy, np.float32).copy()
    rc = lib().orc_corner_subpix(_p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h), _p(out), C.c_int(len(out)), C.c_int(win), C.c_int(max_count),
                                 C.c_double(epsilon))
    if rc != 0:
        raise RuntimeError("orc_corner_subpix rc=%d" % rc)
    return out


def corner_subpix_mask(win=10):
    m = np.zeros((2 * win + 1, 2 * win + 1), np.float32)
    lib().orc_corner_subpix_mask(C.c_int(win), _p(m))
    return m
