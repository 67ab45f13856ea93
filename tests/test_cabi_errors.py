"""Error behaviour of the C ABI (include/amos_frontend.h): every failure is an error code plus a
message, never a crash or an exception across the boundary."""
import ctypes as C

import numpy as np
import pytest


def test_no_gpu_is_an_error_code_not_a_crash(pkg):
    """On a box without a HIP device amos_*_create must fail with AMOS_ERR_DEVICE and a message."""
    L = pkg.lib()
    n = L.amos_device_count()
    if n > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    p = pkg.OrbParams(1000, 1.2, 8, 20, 7)
    rc = L.amos_orb_create(C.byref(p), C.c_int(640), C.c_int(480), C.c_int(1), C.c_int(0), None, C.byref(h))
    assert rc == -2 and not h.value and len(L.amos_last_error()) > 0
    m = C.c_void_p()
    assert L.amos_match_create(C.c_int(0), None, C.byref(m)) == -2
    with pytest.raises(pkg.AmosError):
        pkg.OrbExtractor()


def test_null_and_range_arguments(pkg):
    L = pkg.lib()
    assert L.amos_orb_create(None, C.c_int(640), C.c_int(480), C.c_int(1), C.c_int(0), None, None) == -1
    bad = pkg.OrbParams(1000, 1.2, 99, 20, 7)  # more than AMOS_MAX_LEVELS
    h = C.c_void_p()
    assert L.amos_orb_create(C.byref(bad), C.c_int(640), C.c_int(480), C.c_int(1), C.c_int(0), None, C.byref(h)) == -1
    bad = pkg.OrbParams(1000, 1.0, 8, 20, 7)  # scale factor must be > 1
    assert L.amos_orb_create(C.byref(bad), C.c_int(640), C.c_int(480), C.c_int(1), C.c_int(0), None, C.byref(h)) == -1
    assert L.amos_orb_sync(None) == -1 and L.amos_match_sync(None) == -1
    assert L.amos_orb_stream(None) is None
    L.amos_orb_destroy(None)  # no-ops
    L.amos_match_destroy(None)


def test_new_entry_points_reject_bad_arguments(pkg):
    """The entry points added for the callers either side of the path validate before touching the device."""
    L = pkg.lib()
    f4 = (C.c_float * 4)()
    assert L.amos_frame_image_bounds(C.c_int(0), C.c_int(480), C.c_float(500), C.c_float(500), C.c_float(320), C.c_float(240), None,
                                     C.c_int(0), f4) == -1
    assert L.amos_frame_image_bounds(C.c_int(640), C.c_int(480), C.c_float(0), C.c_float(500), C.c_float(320), C.c_float(240), None,
                                     C.c_int(0), f4) == -1                                  # fx == 0
    assert L.amos_frame_image_bounds(C.c_int(640), C.c_int(480), C.c_float(500), C.c_float(500), C.c_float(320), C.c_float(240), None,
                                     C.c_int(5), f4) == -1                                  # n_dist > 0 without coefficients
    assert L.amos_frame_undistort_batch_device(None, C.c_float(1), C.c_float(1), C.c_float(0), C.c_float(0), None, C.c_int(0), None) == -1
    assert L.amos_frame_rgbd_glue_batch_device(None, None, C.c_int(0), C.c_float(1), C.c_size_t(0), C.c_size_t(0), C.c_float(0), C.c_float(0),
                                               C.c_float(640), C.c_float(0), C.c_float(480), None, None, None, None) == -1
    assert L.amos_frame_grid_build_batch_device(None, None, None, C.c_int(1), C.c_int(1), None, None) == -1
    assert L.amos_match_window_best2_batch_device(None, None, None) == -1
    p = C.c_void_p()
    assert L.amos_mask_pre_create(C.c_int(0), None, C.c_int(1), C.c_int(480), C.c_int(1), C.byref(p)) == -1 and not p.value
    assert L.amos_mask_pre_create(C.c_int(0), None, C.c_int(640), C.c_int(480), C.c_int(0), C.byref(p)) == -1
    assert L.amos_mask_preprocess_batch_device(None, None, C.c_int(1), None) == -1
    L.amos_mask_pre_destroy(None)
    assert len(L.amos_last_error()) > 0


def test_image_bounds_host_routine_matches_the_oracle(pkg, ob):
    """amos_frame_image_bounds runs on the host (four points): bit-exact against the oracle without a GPU."""
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    cam = (517.306408, 516.469215, 318.643040, 255.313989)
    assert pkg.image_bounds(640, 480, *cam, dist) == ob.image_bounds(640, 480, *cam, dist)
    assert pkg.image_bounds(640, 480, *cam, dist[:4]) == ob.image_bounds(640, 480, *cam, dist[:4])
    assert pkg.image_bounds(752, 480, *cam, np.zeros(4, np.float32)) == (0.0, 752.0, 0.0, 480.0)


@pytest.mark.gpu
def test_state_and_capacity_errors(gpu_lib, synth):
    ext = gpu_lib.OrbExtractor(max_width=640, max_height=480, max_batch=2)
    L, h = ext.L, ext.h
    kps = np.zeros(8, gpu_lib.KP_DTYPE)
    desc = np.zeros((8, 32), np.uint8)
    n = C.c_int(0)
    # describe / gate / level access before any detect
    assert L.amos_orb_describe(h, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), C.c_int(8), C.byref(n)) == -4
    assert L.amos_orb_level_count(h, C.c_int(0), C.c_int(0)) == -1
    mask = np.zeros((480, 640), np.uint8)
    with pytest.raises(gpu_lib.AmosError, match="before"):
        ext.gate(mask)
    # frame larger than the handle's allocation
    big = np.zeros((600, 800), np.uint8)
    with pytest.raises(gpu_lib.AmosError, match="exceeds"):
        ext.detect(big)
    # caller buffer too small: count is still reported
    img = synth.frame(1, 1)
    rc = L.amos_orb_extract(h, img.ctypes.data_as(C.c_void_p), C.c_size_t(640), C.c_int(640), C.c_int(480),
                            kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), C.c_int(8), C.byref(n))
    assert rc == -3 and n.value > 900
    # stride smaller than the width
    assert L.amos_orb_detect(h, img.ctypes.data_as(C.c_void_p), C.c_size_t(100), C.c_int(640), C.c_int(480)) == -1
    # batch larger than max_batch
    assert L.amos_orb_extract_batch_device(h, C.c_void_p(256), C.c_size_t(640 * 480), C.c_size_t(640), C.c_int(640), C.c_int(480), C.c_int(3)) == -3
    # level list longer than the level's capacity
    ext.detect(img)
    too_many = np.zeros(5000, gpu_lib.KP_DTYPE)
    with pytest.raises(gpu_lib.AmosError, match="capacity"):
        ext.set_level_keypoints(7, too_many)
    # a keypoint outside the mask is reported, not dereferenced
    k0 = ext.level_keypoints(0).copy()
    k0["x"][0] = 5000.0
    ext.set_level_keypoints(0, k0)
    with pytest.raises(gpu_lib.AmosError, match="outside the mask"):
        ext.gate(mask)


@pytest.mark.gpu
def test_matcher_argument_errors(gpu_lib):
    m = gpu_lib.OrbMatcher()
    q = np.zeros((4, 32), np.uint8)
    t = np.zeros((5, 32), np.uint8)
    off = np.array([0, 2, 2, 3, 4], np.int32)
    with pytest.raises(gpu_lib.AmosError, match="outside"):
        m.list_best2(q, t, off, np.array([0, 1, 9, 2], np.int32))
    with pytest.raises(gpu_lib.AmosError, match="monotonic"):
        m.list_best2(q, t, np.array([0, 2, 1, 3, 4], np.int32), np.array([0, 1, 2, 3], np.int32))
    with pytest.raises(gpu_lib.AmosError, match="65535"):
        m.bruteforce_best2(q, np.zeros((70000, 32), np.uint8))
    # empty candidate lists are fine
    r = m.list_best2(q, t, np.zeros(5, np.int32), np.zeros(0, np.int32))
    assert (r["best_idx"] == -1).all() and (r["best_dist"] == 256).all()
