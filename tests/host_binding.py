"""ctypes binding of tests/host/libamos_host_test.so (the harness over the C++ drop-in classes of amos-slam_amd/host/libamos_host.so) and
of the oracle's gated-search restatements."""
import ctypes as C
import os

import numpy as np

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "tests", "host", "libamos_host_test.so")

KP = ob.KP_DTYPE
PROJ_QUERY = np.dtype([("u", "<f4"), ("v", "<f4"), ("invz", "<f4"), ("octave", "<i4"), ("angle", "<f4"), ("has_obs", "<i4"),
                       ("desc", "u1", (32,))])
MAP_QUERY = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                      ("has_obs", "<i4"), ("desc", "u1", (32,))])


KF_QUERY = np.dtype([("u", "<f4"), ("v", "<f4"), ("level", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))])
WINDOW_QUERY = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("level", "<i4"), ("src", "<i4"), ("desc", "u1", (32,))])
MATCH_FREE, MATCH_TAKEN = -1, -2


class BowView(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys", C.c_void_p), ("descriptors", C.c_void_p), ("has_point", C.c_void_p), ("u_right", C.c_void_p),
                ("n_nodes", C.c_int32), ("node_ids", C.c_void_p), ("node_off", C.c_void_p), ("node_idx", C.c_void_p)]


def bow_view(kps, desc, nodes, has_point=None, u_right=None):
    """nodes: dict node id -> list of feature indices (a DBoW2::FeatureVector).  Returns (view, keepalive)."""
    kps = np.ascontiguousarray(kps, KP)
    desc = np.ascontiguousarray(desc, np.uint8)
    ids = np.array(sorted(nodes), np.uint32)
    off = np.zeros(len(ids) + 1, np.int32)
    idx = []
    for k, nid in enumerate(ids):
        idx.extend(nodes[int(nid)])
        off[k + 1] = len(idx)
    idx = np.array(idx if idx else [0], np.int32)
    hp = None if has_point is None else np.ascontiguousarray(has_point, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    v = BowView(len(kps), kps.ctypes.data, desc.ctypes.data, hp.ctypes.data if hp is not None else None,
                ur.ctypes.data if ur is not None else None, len(ids), ids.ctypes.data, off.ctypes.data, idx.ctypes.data)
    return v, (kps, desc, ids, off, idx, hp, ur)


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("descriptors", C.c_void_p), ("u_right", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


def frame_view(kps, desc, u_right=None, bounds=(0.0, 640.0, 0.0, 480.0)):
    """Returns (view, keepalive) for numpy arrays."""
    kps = np.ascontiguousarray(kps, KP)
    desc = np.ascontiguousarray(desc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    v = FrameView(len(kps), kps.ctypes.data, desc.ctypes.data, ur.ctypes.data if ur is not None else None, *bounds)
    return v, (kps, desc, ur)


def window_best2(train_view, qk, qdesc, scale_factors, th, mode=0, init_dist=256, query_uv=None, query_invz=None, mbf=0.0):
    """oracle: GetFeaturesInArea + best/second loop without the greedy skip (orc_window_best2)."""
    qk, qdesc = np.ascontiguousarray(qk, KP), np.ascontiguousarray(qdesc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    uv = None if query_uv is None else np.ascontiguousarray(query_uv, np.float32)
    iz = None if query_invz is None else np.ascontiguousarray(query_invz, np.float32)
    out = np.zeros(len(qk), ob.BEST2_DTYPE)
    ob.lib().orc_window_best2(C.byref(train_view), _p(qk), _p(qdesc), C.c_int(len(qk)), _p(uv), _p(iz), _p(sf), C.c_float(th), C.c_float(mbf),
                              C.c_int(mode), C.c_int(init_dist), _p(out))
    return out


_host = None


def host():
    global _host
    if _host is None:
        import __graft_entry__ as entry
        entry.load_package().lib()  # loads torch's HIP runtime first, then libamos_frontend.so
        _host = C.CDLL(HOST_SO)
        _host.amos_host_last_error.restype = C.c_char_p
    return _host


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc):
    if rc < 0:
        raise RuntimeError(f"host call failed rc={rc}: {host().amos_host_last_error().decode()}")
    return rc


def host_extract(gray, nf=1000, sf=1.2, nl=8, ini=20, mn=7, pyr_level=None):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = nf * 2 + 64 * nl
    kps, desc, n = np.zeros(cap, KP), np.zeros((cap, 32), np.uint8), C.c_int(0)
    pyr = None
    if pyr_level is not None:
        lw, lh = ob.Oracle(nf, sf, nl).level_sizes(w, h)
        pyr = np.zeros((int(lh[pyr_level]) + 38, int(lw[pyr_level]) + 38), np.uint8)
    _chk(host().amos_host_extract(_p(gray), C.c_int(w), C.c_int(h), C.c_int(nf), C.c_float(sf), C.c_int(nl), C.c_int(ini), C.c_int(mn),
                                  _p(kps), _p(desc), C.c_int(cap), C.byref(n), C.c_int(pyr_level or 0), _p(pyr)))
    return kps[:n.value].copy(), desc[:n.value].copy(), pyr


def host_amos_flow(gray, mask, labels=None, center_ids=None, rm=None, nf=1000, sf=1.2, nl=8, ini=20, mn=7):
    gray, mask = np.ascontiguousarray(gray, np.uint8), np.ascontiguousarray(mask, np.uint8)
    h, w = gray.shape
    cap = nf * 2 + 64 * nl
    kps, desc, n = np.zeros(cap, KP), np.zeros((cap, 32), np.uint8), C.c_int(0)
    removed, nrem = np.zeros(cap, KP), C.c_int(0)
    lists, counts = np.zeros(cap, KP), np.zeros(nl, np.int32)
    if labels is not None:
        labels = np.ascontiguousarray(labels, np.float64)
        center_ids = np.ascontiguousarray(center_ids, np.int32)
        rm = np.ascontiguousarray(rm, np.int32)
    _chk(host().amos_host_amos_flow(_p(gray), C.c_int(w), C.c_int(h), C.c_int(nf), C.c_float(sf), C.c_int(nl), C.c_int(ini), C.c_int(mn),
                                    _p(mask), _p(labels), _p(center_ids), C.c_int(0 if center_ids is None else len(center_ids)), _p(rm),
                                    C.c_int(0 if rm is None else len(rm)), _p(removed), C.byref(nrem), _p(kps), _p(desc), C.c_int(cap),
                                    C.byref(n), _p(lists), _p(counts)))
    return removed[:nrem.value].copy(), kps[:n.value].copy(), desc[:n.value].copy(), lists[:counts.sum()].copy(), counts


def host_pyramid_on_demand(gray, nl=8, pyr_level=2):
    """3-arg operator() under the default pyramid mode, then DownloadPyramid(): (rows, cols per level before any pixel copy, padded plane of
    pyr_level afterwards, the extractor's device)."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    lw, lh = ob.Oracle(1000, 1.2, nl).level_sizes(w, h)
    rc = np.zeros((nl, 2), np.int32)
    pyr = np.zeros((int(lh[pyr_level]) + 38, int(lw[pyr_level]) + 38), np.uint8)
    dev = C.c_int32(-7)
    _chk(host().amos_host_pyramid_on_demand(_p(gray), C.c_int(w), C.c_int(h), C.c_int(nl), _p(rc), C.c_int(pyr_level), _p(pyr), C.byref(dev)))
    return rc, pyr, dev.value


class ThreadExtractJob(C.Structure):
    _fields_ = [("gray", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p), ("cap", C.c_int32),
                ("n", C.c_int32), ("rc", C.c_int32)]


class ThreadMatchJob(C.Structure):
    _fields_ = [("cur", C.c_void_p), ("q", C.c_void_p), ("nq", C.c_int32), ("cur_match_in", C.c_void_p), ("cur_match_out", C.c_void_p),
                ("scale_factors", C.c_void_p), ("nsf", C.c_int32), ("mbf", C.c_float), ("th", C.c_float), ("forward", C.c_int32),
                ("backward", C.c_int32), ("result", C.c_int32), ("rc", C.c_int32)]


def host_run_threads(frames, searches, iters=6):
    """frames: gray images, one ORBextractor + std::thread each (4-arg operator(), `iters` times); searches: (frame view, PROJ_QUERY array,
    cur_match, scale factors, mbf, th, forward, backward), one std::thread each constructing a stack ORBmatcher per iteration.  Returns
    ([(kps, desc)], [(result, cur_match)], handles the matcher pool has created so far)."""
    keep = []
    ej = (ThreadExtractJob * len(frames))()
    for j, g in zip(ej, frames):
        g = np.ascontiguousarray(g, np.uint8)
        cap = 2000 + 64 * 8
        kps, desc = np.zeros(cap, KP), np.zeros((cap, 32), np.uint8)
        keep.append((g, kps, desc))
        j.gray, j.w, j.h, j.kps, j.desc, j.cap, j.n, j.rc = g.ctypes.data, g.shape[1], g.shape[0], kps.ctypes.data, desc.ctypes.data, cap, 0, -9
    mj = (ThreadMatchJob * len(searches))()
    outs = []
    for j, (view, q, match0, sf, mbf, th, fwd, bwd) in zip(mj, searches):
        q = np.ascontiguousarray(q, PROJ_QUERY)
        m_in = np.ascontiguousarray(match0, np.int32)
        m_out = np.zeros_like(m_in)
        sf = np.ascontiguousarray(sf, np.float32)
        keep.append((view, q, m_in, sf))
        outs.append(m_out)
        j.cur, j.q, j.nq, j.cur_match_in, j.cur_match_out = C.addressof(view), q.ctypes.data, len(q), m_in.ctypes.data, m_out.ctypes.data
        j.scale_factors, j.nsf, j.mbf, j.th, j.forward, j.backward, j.result, j.rc = sf.ctypes.data, len(sf), mbf, th, fwd, bwd, -9, -9
    pool = C.c_int(0)
    _chk(host().amos_host_run_threads(ej, C.c_int(len(frames)), mj, C.c_int(len(searches)), C.c_int(iters), C.byref(pool)))
    ext = []
    for j, (_, kps, desc) in zip(ej, keep[:len(frames)]):
        assert j.rc == 0, f"extractor thread: rc {j.rc}"
        ext.append((kps[:j.n].copy(), desc[:j.n].copy()))
    res = []
    for j, m in zip(mj, outs):
        assert j.rc == 0, f"matcher thread: rc {j.rc}"
        res.append((j.result, m))
    return ext, res, pool.value


def host_frame_latency(gray_frames, bgr_frames=None, py_file=None, weights="", warm=5, iters=100, pyramid_mode=-1):
    """amos_host_frame_latency: the per-frame C++ drop-in path (evalImage -> operator() -> MovingKeyPoints -> ProcessDesp -> stack ORBmatcher
    SearchByProjection), host buffers in and out.  Returns a dict of mean milliseconds per frame and the last frame's counts."""
    gray = np.ascontiguousarray(gray_frames, np.uint8)
    n, h, w = gray.shape
    bgr = None if bgr_frames is None else np.ascontiguousarray(bgr_frames, np.uint8)
    ms, counts = np.zeros(6, np.float64), np.zeros(3, np.int32)
    _chk(host().amos_host_frame_latency((py_file or "").encode(), (weights or "").encode(), _p(bgr), _p(gray), C.c_int(n), C.c_int(w), C.c_int(h),
                                        C.c_int(warm), C.c_int(iters), C.c_int(pyramid_mode), _p(ms), _p(counts)))
    keys = ("eval_image_ms", "detect_ms", "moving_keypoints_ms", "process_desp_ms", "search_by_projection_ms", "frame_ms")
    out = {k: round(float(v), 4) for k, v in zip(keys, ms)}
    out.update(keypoints_last_frame=int(counts[0]), matches_last_frame=int(counts[1]), eval_image_false=int(counts[2]), frames=iters)
    return out


def host_descriptor_distance(a, b):
    a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
    return _chk(host().amos_host_descriptor_distance(_p(a), _p(b)))


def _area(fn, view, x, y, r, lo, hi):
    out = np.zeros(view.n + 1, np.int32)
    n = fn(C.byref(view), C.c_float(x), C.c_float(y), C.c_float(r), C.c_int(lo), C.c_int(hi), _p(out), C.c_int(len(out)))
    assert n >= 0
    return out[:n].copy()


def host_features_in_area(view, x, y, r, lo=-1, hi=-1):
    return _area(host().amos_host_features_in_area, view, x, y, r, lo, hi)


def oracle_features_in_area(view, x, y, r, lo=-1, hi=-1):
    return _area(ob.lib().orc_features_in_area, view, x, y, r, lo, hi)


def search_frame(which, view, queries, cur_match, scale_factors, mbf, th, forward, backward, nnratio=0.9, check_ori=True):
    queries = np.ascontiguousarray(queries, PROJ_QUERY)
    match = np.ascontiguousarray(cur_match, np.int32).copy()
    sf = np.ascontiguousarray(scale_factors, np.float32)
    if which == "host":
        r = _chk(host().amos_host_search_by_projection_frame(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(sf),
                                                             C.c_int(len(sf)), C.c_float(mbf), C.c_float(th), C.c_int(forward),
                                                             C.c_int(backward), C.c_float(nnratio), C.c_int(check_ori)))
    else:
        r = ob.lib().orc_search_by_projection_frame(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(sf), C.c_float(mbf),
                                                    C.c_float(th), C.c_int(forward), C.c_int(backward), C.c_int(check_ori))
    return r, match


def search_points(which, view, queries, cur_match, cur_has_obs, scale_factors, th, nnratio=0.8):
    queries = np.ascontiguousarray(queries, MAP_QUERY)
    match = np.ascontiguousarray(cur_match, np.int32).copy()
    obs = np.ascontiguousarray(cur_has_obs, np.uint8).copy()
    sf = np.ascontiguousarray(scale_factors, np.float32)
    if which == "host":
        r = _chk(host().amos_host_search_by_projection_points(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(obs), _p(sf),
                                                              C.c_int(len(sf)), C.c_float(th), C.c_float(nnratio)))
    else:
        r = ob.lib().orc_search_by_projection_points(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(obs), _p(sf),
                                                     C.c_float(th), C.c_float(nnratio))
    return r, match, obs


def search_kf(which, view, queries, cur_match, scale_factors, th, orb_dist, nnratio=0.9, check_ori=True):
    queries = np.ascontiguousarray(queries, KF_QUERY)
    match = np.ascontiguousarray(cur_match, np.int32).copy()
    sf = np.ascontiguousarray(scale_factors, np.float32)
    if which == "host":
        r = _chk(host().amos_host_search_by_projection_kf(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(sf), C.c_int(len(sf)),
                                                          C.c_float(th), C.c_int(orb_dist), C.c_float(nnratio), C.c_int(check_ori)))
    else:
        r = ob.lib().orc_search_by_projection_kf(C.byref(view), _p(queries), C.c_int(len(queries)), _p(match), _p(sf), C.c_float(th),
                                                 C.c_int(orb_dist), C.c_int(check_ori))
    return r, match


def search_bow(which, kf_view, f_view, nnratio=0.7, check_ori=True):
    m = np.zeros(max(f_view.n, 1), np.int32)
    if which == "host":
        r = _chk(host().amos_host_search_by_bow(C.byref(kf_view), C.byref(f_view), _p(m), C.c_float(nnratio), C.c_int(check_ori)))
    else:
        r = ob.lib().orc_search_by_bow(C.byref(kf_view), C.byref(f_view), _p(m), C.c_float(nnratio), C.c_int(check_ori))
    return r, m[:f_view.n]


def search_bow_kf(which, v1, v2, nnratio=0.75, check_ori=True):
    m = np.zeros(max(v1.n, 1), np.int32)
    if which == "host":
        r = _chk(host().amos_host_search_by_bow_kf(C.byref(v1), C.byref(v2), _p(m), C.c_float(nnratio), C.c_int(check_ori)))
    else:
        r = ob.lib().orc_search_by_bow_kf(C.byref(v1), C.byref(v2), _p(m), C.c_float(nnratio), C.c_int(check_ori))
    return r, m[:v1.n]


def search_triangulation(which, v1, v2, f12, ex, ey, scale_factors2, level_sigma2_2, only_stereo, check_ori=True, nnratio=0.6):
    f12 = np.ascontiguousarray(f12, np.float32).reshape(9)
    sf, sg = np.ascontiguousarray(scale_factors2, np.float32), np.ascontiguousarray(level_sigma2_2, np.float32)
    pairs = np.zeros((max(v1.n, 1), 2), np.int32)
    if which == "host":
        r = _chk(host().amos_host_search_for_triangulation(C.byref(v1), C.byref(v2), _p(f12), C.c_float(ex), C.c_float(ey), _p(sf), _p(sg),
                                                           C.c_int(len(sf)), C.c_int(only_stereo), C.c_float(nnratio), C.c_int(check_ori),
                                                           _p(pairs), C.c_int(len(pairs))))
    else:
        r = ob.lib().orc_search_for_triangulation(C.byref(v1), C.byref(v2), _p(f12), C.c_float(ex), C.c_float(ey), _p(sf), _p(sg),
                                                  C.c_int(only_stereo), C.c_int(check_ori), _p(pairs), C.c_int(len(pairs)))
    return r, pairs[:max(r, 0)].copy()


def fuse(which, view, queries, scale_factors, th, inv_level_sigma2=None):
    """Fuse #1 (inv_level_sigma2 given: chi2 gate) / Fuse #2.  Returns (nFused, best feature per query)."""
    q = np.ascontiguousarray(queries, WINDOW_QUERY)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    is2 = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    best = np.zeros(max(len(q), 1), np.int32)
    if which == "host":
        r = _chk(host().amos_host_fuse(C.byref(view), _p(q), C.c_int(len(q)), _p(sf), _p(is2), C.c_int(len(sf)), C.c_float(th), _p(best)))
    else:
        r = ob.lib().orc_window_search(C.byref(view), _p(q), C.c_int(len(q)), _p(sf), _p(is2), C.c_float(th), C.c_int(50), None, _p(best))
    return r, best[:len(q)]


def search_projection_sim(which, view, queries, matched, scale_factors, th):
    q = np.ascontiguousarray(queries, WINDOW_QUERY)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.ascontiguousarray(matched, np.int32).copy()
    if which == "host":
        r = _chk(host().amos_host_search_by_projection_sim(C.byref(view), _p(q), C.c_int(len(q)), _p(m), _p(sf), C.c_int(len(sf)), C.c_int(th)))
    else:
        best = np.zeros(max(len(q), 1), np.int32)
        r = ob.lib().orc_window_search(C.byref(view), _p(q), C.c_int(len(q)), _p(sf), None, C.c_float(th), C.c_int(50), _p(m), _p(best))
    return r, m


def search_sim3(which, v1, v2, q12, q21, sf1, sf2, th):
    q12, q21 = np.ascontiguousarray(q12, WINDOW_QUERY), np.ascontiguousarray(q21, WINDOW_QUERY)
    sf1, sf2 = np.ascontiguousarray(sf1, np.float32), np.ascontiguousarray(sf2, np.float32)
    m = np.zeros(max(v1.n, 1), np.int32)
    if which == "host":
        r = _chk(host().amos_host_search_by_sim3(C.byref(v1), C.byref(v2), _p(q12), C.c_int(len(q12)), _p(q21), C.c_int(len(q21)), _p(sf1),
                                                 _p(sf2), C.c_int(len(sf1)), C.c_float(th), _p(m)))
    else:
        r = ob.lib().orc_search_by_sim3(C.byref(v1), C.byref(v2), _p(q12), C.c_int(len(q12)), _p(q21), C.c_int(len(q21)), _p(sf1), _p(sf2),
                                        C.c_float(th), _p(m))
    return r, m[:v1.n]


def search_init(which, view1, view2, prev_matched, window=100, nnratio=0.9, check_ori=True):
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(view1.n, np.int32)
    if which == "host":
        r = _chk(host().amos_host_search_for_initialization(C.byref(view1), C.byref(view2), _p(prev), _p(m12), C.c_int(window),
                                                            C.c_float(nnratio), C.c_int(check_ori)))
    else:
        r = ob.lib().orc_search_for_initialization(C.byref(view1), C.byref(view2), _p(prev), _p(m12), C.c_int(window), C.c_float(nnratio),
                                                   C.c_int(check_ori))
    return r, m12, prev


def host_yolact_eval(py_file, weights, bgr):
    """ORB_SLAM2::yolact(py_file, weights, 20).evalImage(bgr) -> mask or raises RuntimeError."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.zeros((h, w), np.uint8)
    mw, mh = C.c_int(0), C.c_int(0)
    _chk(host().amos_host_yolact_eval(py_file.encode(), weights.encode(), _p(bgr), C.c_int(w), C.c_int(h), _p(out), C.byref(mw),
                                      C.byref(mh)))
    assert (mh.value, mw.value) == (h, w)
    return out


# ---- the reference-signature adaptors (amos-slam_amd/host/ORBmatcher_adaptors.h) on stand-in Frame / MapPoint objects

class TestCamera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mb", C.c_float), ("mbf", C.c_float),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("Tcw", C.c_float * 16),
                ("n_levels", C.c_int32), ("scale_factors", C.c_float * 16)]


def test_camera(Tcw, scale_factors, fx=535.4, fy=539.2, cx=320.1, cy=247.6, mb=0.08, mbf=40.0, bounds=(0.0, 640.0, 0.0, 480.0)):
    cam = TestCamera(fx, fy, cx, cy, mb, mbf, *bounds)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    for i in range(16):
        cam.Tcw[i] = float(T[i])
    cam.n_levels = len(scale_factors)
    for i, s in enumerate(scale_factors):
        cam.scale_factors[i] = float(s)
    return cam


def ref_search_last_frame(cur_cam, cur_keys_un, cur_desc, cur_u_right, cur_occupant_obs, last_cam, last_keys, last_keys_un, has_point, outlier,
                          world, mp_desc, mp_obs, th, mono, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), reference signature."""
    ck, cd = np.ascontiguousarray(cur_keys_un, KP), np.ascontiguousarray(cur_desc, np.uint8)
    cur = None if cur_u_right is None else np.ascontiguousarray(cur_u_right, np.float32)
    occ = None if cur_occupant_obs is None else np.ascontiguousarray(cur_occupant_obs, np.int32)
    lk, lku = np.ascontiguousarray(last_keys, KP), np.ascontiguousarray(last_keys_un, KP)
    hp, ol = np.ascontiguousarray(has_point, np.uint8), np.ascontiguousarray(outlier, np.uint8)
    w, md, mo = np.ascontiguousarray(world, np.float32), np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(mp_obs, np.int32)
    match = np.zeros(max(len(ck), 1), np.int32)
    r = _chk(host().amos_host_ref_search_last_frame(C.byref(cur_cam), C.c_int(len(ck)), _p(ck), _p(cd), _p(cur), _p(occ), C.byref(last_cam),
                                                   C.c_int(len(lk)), _p(lk), _p(lku), _p(hp), _p(ol), _p(w), _p(md), _p(mo), C.c_float(th),
                                                   C.c_int(int(mono)), C.c_float(nnratio), C.c_int(int(check_ori)), _p(match)))
    return r, match[:len(ck)]


def ref_search_local_points(cam, keys_un, desc, u_right, points, in_view, bad, th, nnratio=0.8):
    """ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th), reference signature."""
    k, d = np.ascontiguousarray(keys_un, KP), np.ascontiguousarray(desc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    pts = np.ascontiguousarray(points, MAP_QUERY)
    iv, bd = np.ascontiguousarray(in_view, np.uint8), np.ascontiguousarray(bad, np.uint8)
    match = np.zeros(max(len(k), 1), np.int32)
    r = _chk(host().amos_host_ref_search_local_points(C.byref(cam), C.c_int(len(k)), _p(k), _p(d), _p(ur), _p(pts), _p(iv), _p(bd), C.c_int(len(pts)),
                                                      C.c_float(th), C.c_float(nnratio), _p(match)))
    return r, match[:len(k)]


# ---- the stand-in "map" for the other eight reference signatures (host_capi.cc: amos_test_points / amos_test_kf)

class TestPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("world", C.c_void_p), ("normal", C.c_void_p), ("desc", C.c_void_p), ("obs", C.c_void_p), ("bad", C.c_void_p),
                ("min_dist", C.c_void_p), ("max_dist", C.c_void_p)]


class TestKF(C.Structure):
    _fields_ = [("cam", TestCamera), ("n", C.c_int32), ("keys", C.c_void_p), ("keys_un", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p),
                ("point_of", C.c_void_p), ("n_nodes", C.c_int32), ("node_ids", C.c_void_p), ("node_off", C.c_void_p), ("node_idx", C.c_void_p)]


def test_points(world, normal, desc, obs, bad, min_dist, max_dist):
    keep = (np.ascontiguousarray(world, np.float32), np.ascontiguousarray(normal, np.float32), np.ascontiguousarray(desc, np.uint8),
            np.ascontiguousarray(obs, np.int32), np.ascontiguousarray(bad, np.uint8), np.ascontiguousarray(min_dist, np.float32),
            np.ascontiguousarray(max_dist, np.float32))
    return TestPoints(len(keep[0]), *[a.ctypes.data for a in keep]), keep


def test_kf(cam, keys, desc, u_right=None, point_of=None, nodes=None, keys_un=None):
    """A stand-in Frame / KeyFrame: camera (pose inside), features, per-feature map point index, feature vector {node: [features]}."""
    k = np.ascontiguousarray(keys, KP)
    ku = k if keys_un is None else np.ascontiguousarray(keys_un, KP)
    d = np.ascontiguousarray(desc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    po = None if point_of is None else np.ascontiguousarray(point_of, np.int32)
    nodes = nodes or {}
    ids = np.array(sorted(nodes), np.uint32)
    off = np.zeros(len(ids) + 1, np.int32)
    idx = []
    for j, nid in enumerate(ids):
        idx.extend(nodes[int(nid)])
        off[j + 1] = len(idx)
    idx = np.array(idx if idx else [0], np.int32)
    s = TestKF(cam, len(k), k.ctypes.data, ku.ctypes.data, d.ctypes.data, ur.ctypes.data if ur is not None else None,
               po.ctypes.data if po is not None else None, len(ids), ids.ctypes.data, off.ctypes.data, idx.ctypes.data)
    return s, (k, ku, d, ur, po, ids, off, idx)


def standin_predict_scale(max_dist, cur_dist, scale_factor, n_levels):
    """MapPoint::PredictScale of the stand-in for arrays of (mfMaxDistance, current distance); scale_factor = mvScaleFactors[1]."""
    host().amos_host_standin_predict_scale.restype = C.c_int
    return [int(host().amos_host_standin_predict_scale(C.c_float(float(m)), C.c_float(float(c)), C.c_float(float(scale_factor)), C.c_int(n_levels)))
            for m, c in zip(np.atleast_1d(max_dist), np.atleast_1d(cur_dist))]


def ref_search_reloc(cur, kf, pts, already_found, th, orb_dist, nnratio=0.9, check_ori=True):
    out = np.zeros(max(cur.n, 1), np.int32)
    af = np.ascontiguousarray(already_found, np.uint8)
    r = _chk(host().amos_host_ref_search_reloc(C.byref(cur), C.byref(kf), C.byref(pts), _p(af), C.c_float(th), C.c_int(orb_dist), C.c_float(nnratio),
                                              C.c_int(int(check_ori)), _p(out)))
    return r, out[:cur.n]


def ref_search_kf_scw(kf, pts, scw, vp, matched, th, nnratio=0.75):
    m = np.ascontiguousarray(matched, np.int32).copy()
    vp = np.ascontiguousarray(vp, np.int32)
    s = np.ascontiguousarray(scw, np.float32).reshape(16)
    r = _chk(host().amos_host_ref_search_kf_scw(C.byref(kf), C.byref(pts), _p(s), _p(vp), C.c_int(len(vp)), _p(m), C.c_int(th), C.c_float(nnratio)))
    return r, m


def ref_search_bow_kf_frame(kf, frame, pts, nnratio=0.7, check_ori=True):
    m = np.zeros(max(frame.n, 1), np.int32)
    r = _chk(host().amos_host_ref_search_bow_kf_frame(C.byref(kf), C.byref(frame), C.byref(pts), C.c_float(nnratio), C.c_int(int(check_ori)), _p(m)))
    return r, m[:frame.n]


def ref_search_bow_kf_kf(kf1, kf2, pts, nnratio=0.75, check_ori=True):
    m = np.zeros(max(kf1.n, 1), np.int32)
    r = _chk(host().amos_host_ref_search_bow_kf_kf(C.byref(kf1), C.byref(kf2), C.byref(pts), C.c_float(nnratio), C.c_int(int(check_ori)), _p(m)))
    return r, m[:kf1.n]


def ref_search_initialization(f1, f2, prev_matched, window, nnratio=0.9, check_ori=True):
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(f1.n, 1), np.int32)
    r = _chk(host().amos_host_ref_search_initialization(C.byref(f1), C.byref(f2), _p(prev), _p(m12), C.c_int(window), C.c_float(nnratio),
                                                       C.c_int(int(check_ori))))
    return r, m12[:f1.n], prev


def ref_search_triangulation(kf1, kf2, pts, f12, only_stereo, nnratio=0.6, check_ori=True):
    f = np.ascontiguousarray(f12, np.float32).reshape(9)
    pairs = np.zeros((max(kf1.n, 1), 2), np.int32)
    r = _chk(host().amos_host_ref_search_triangulation(C.byref(kf1), C.byref(kf2), C.byref(pts), _p(f), C.c_int(int(only_stereo)), C.c_float(nnratio),
                                                      C.c_int(int(check_ori)), _p(pairs), C.c_int(len(pairs))))
    return r, pairs[:max(r, 0)].copy()


def ref_search_sim3(kf1, kf2, pts, matches12, s12, r12, t12, th, nnratio=0.75):
    m = np.ascontiguousarray(matches12, np.int32).copy()
    R, t = np.ascontiguousarray(r12, np.float32).reshape(9), np.ascontiguousarray(t12, np.float32).reshape(3)
    r = _chk(host().amos_host_ref_search_sim3(C.byref(kf1), C.byref(kf2), C.byref(pts), _p(m), C.c_float(s12), _p(R), _p(t), C.c_float(th),
                                             C.c_float(nnratio)))
    return r, m


def ref_fuse(kf, pts, vp, th, nnratio=0.6):
    vp = np.ascontiguousarray(vp, np.int32)
    kf_points = np.zeros(max(kf.n, 1), np.int32)
    replaced, obs, bad = np.zeros(pts.n, np.int32), np.zeros(pts.n, np.int32), np.zeros(pts.n, np.uint8)
    r = _chk(host().amos_host_ref_fuse(C.byref(kf), C.byref(pts), _p(vp), C.c_int(len(vp)), C.c_float(th), C.c_float(nnratio), _p(kf_points), _p(replaced),
                                      _p(obs), _p(bad)))
    return r, kf_points[:kf.n], replaced, obs, bad


def ref_fuse_scw(kf, pts, scw, vp, th, replace_point, nnratio=0.6):
    vp = np.ascontiguousarray(vp, np.int32)
    rp = np.ascontiguousarray(replace_point, np.int32).copy()
    s = np.ascontiguousarray(scw, np.float32).reshape(16)
    kf_points, obs = np.zeros(max(kf.n, 1), np.int32), np.zeros(pts.n, np.int32)
    r = _chk(host().amos_host_ref_fuse_scw(C.byref(kf), C.byref(pts), _p(s), _p(vp), C.c_int(len(vp)), C.c_float(th), C.c_float(nnratio), _p(rp),
                                          _p(kf_points), _p(obs)))
    return r, rp, kf_points[:kf.n], obs
