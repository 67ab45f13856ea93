"""amos-slam_amd: MI355X-native front-end hot path of Amos-SLAM (ORB extract + Hamming match +
mask gate), as a thin Python mirror of the C ABI in include/amos_frontend.h.

The compute is in csrc/libamos_frontend.so (hand-written HIP for gfx950).  There is no CPU
fallback: importing the binding without the built library raises, and every call goes to the GPU.
The directory name has a hyphen (it is the project's name); load it with `load_package()` from
__graft_entry__.py or tests/conftest.py, which registers it as module `amos_slam_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMOS_FRONTEND_LIB", os.path.join(_HERE, "csrc", "libamos_frontend.so"))  # override: kernel experiments

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
BEST2_DTYPE = np.dtype([("best_idx", "<i4"), ("best_dist", "<i4"), ("second_idx", "<i4"),
                        ("second_dist", "<i4")])

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # ORBmatcher.cc:49-51

EXPORTS = [
    "amos_last_error", "amos_device_count", "amos_current_device", "amos_build_variant", "amos_orb_tables_host", "amos_orb_geometry_probe", "amos_orb_create", "amos_orb_destroy", "amos_orb_tables",
    "amos_orb_level_sizes", "amos_orb_detect", "amos_orb_level_count", "amos_orb_level_keypoints",
    "amos_orb_set_level_keypoints", "amos_orb_level_layout", "amos_orb_fetch_levels", "amos_orb_store_levels", "amos_orb_gate", "amos_orb_closed_mask", "amos_orb_describe",
    "amos_orb_extract", "amos_orb_level_image", "amos_orb_pyramid_images", "amos_orb_blurred_image", "amos_orb_level_candidates",
    "amos_orb_extract_batch_device", "amos_orb_detect_batch_device", "amos_orb_gate_batch_device",
    "amos_orb_describe_batch_device", "amos_orb_extract_batch_device_color", "amos_frame_rgbd_glue_batch_device", "amos_frame_undistort_batch_device", "amos_frame_image_bounds", "amos_frame_grid_build_batch_device", "amos_match_window_best2_batch_device",
    "amos_orb_batch_results_device", "amos_orb_batch_fetch", "amos_orb_sync",
    "amos_orb_stream", "amos_orb_timing_enable", "amos_orb_timing_collect", "amos_match_create", "amos_match_destroy", "amos_match_sync", "amos_match_stream",
    "amos_match_distances", "amos_match_list_distances", "amos_match_list_best2", "amos_match_bruteforce_best2",
    "amos_match_bruteforce_best2_batch_device", "amos_match_set_bruteforce_kernel", "amos_slic_center_count", "amos_slic_create", "amos_slic_destroy", "amos_slic_stream",
    "amos_slic_run", "amos_slic_batch_device", "amos_cluster_kmeans_batch_device", "amos_cluster_kmeans", "amos_cluster_bgr2lab_batch_device", "amos_flow_check_device", "amos_flow_epipolar_device", "amos_flow_scene_flow_device", "amos_flow_fundamental_score_device", "amos_flow_pnp_score_device", "amos_lk_create", "amos_lk_destroy", "amos_lk_stream", "amos_lk_levels", "amos_lk_track_device", "amos_mask_pre_create", "amos_mask_pre_destroy", "amos_mask_pre_stream", "amos_mask_preprocess_batch_device", "amos_mask_bias_act_device", "amos_mask_bias_relu_maxpool_device", "amos_mask_stem_weight_floats", "amos_mask_stem_weights_device", "amos_mask_stem_device", "amos_mask_conv1x1_supported", "amos_mask_conv1x1_device", "amos_mask_conv_supported", "amos_mask_conv_device", "amos_mask_conv_workspace_bytes", "amos_mask_conv_ws_device", "amos_mask_conv_tile_mode", "amos_mask_conv_kernel_name", "amos_corners_create", "amos_corners_destroy", "amos_corners_stream", "amos_corners_good_features_device", "amos_corners_candidate_count", "amos_corners_subpix_device", "amos_mask_winograd_supported", "amos_mask_winograd_weight_floats", "amos_mask_winograd_weights_device", "amos_mask_winograd_conv_device", "amos_mask_winograd24_weight_floats", "amos_mask_winograd24_weights_device", "amos_mask_winograd24_conv_device", "amos_mask_winograd24_conv_layout_device", "amos_mask_winograd24_persistent_mode", "amos_mask_winograd24_narrow_mode", "amos_mask_bilinear_nhwc_device", "amos_mask_bilinear_nhwc_act_device", "amos_mask_bilinear_x2_mode", "amos_mask_nms_column_max_device", "amos_mask_class_scores_device", "amos_mask_person_mask_device", "amos_mask_head_outputs_device", "amos_mask_head_outputs_scores_device", "amos_mask_person_masks_scores_device", "amos_mask_topk_rows_device", "amos_mask_topk_rows_sparse_device", "amos_mask_post_workspace_bytes", "amos_mask_person_masks_device", "amos_orb_detect_color_with_mask_pre_batch_device",
]


class AmosError(RuntimeError):
    pass


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


_lib = None


def lib():
    """The C-ABI library.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmosError(f"{LIB_PATH} is missing: build it with `make -C amos-slam_amd/csrc` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 under the same sonames as
            # /opt/rocm.  One process must hold ONE HIP runtime: when torch is used beside this library
            # (device memory, streams, torch.distributed), load torch's copy first so both share it.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.amos_last_error.restype = C.c_char_p
        L.amos_build_variant.restype = C.c_char_p
        variant = L.amos_build_variant().decode()
        if variant != "default" and os.environ.get("AMOS_ALLOW_EXPERIMENT_BUILD") != "1":
            # timing-experiment builds (tools/*_variants.sh) compute WRONG results by design: never load one by accident (a stale
            # AMOS_FRONTEND_LIB export); the experiment scripts set AMOS_ALLOW_EXPERIMENT_BUILD=1 themselves
            raise AmosError(f"{LIB_PATH} is an experiment build ({variant}): its results are wrong; unset AMOS_FRONTEND_LIB "
                            "or set AMOS_ALLOW_EXPERIMENT_BUILD=1 for a timing run")
        L.amos_orb_stream.restype = C.c_void_p
        L.amos_match_stream.restype = C.c_void_p
        L.amos_orb_destroy.restype = None
        L.amos_match_destroy.restype = None
        L.amos_orb_destroy.argtypes = [C.c_void_p]
        L.amos_match_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise AmosError(f"{what} failed (rc={rc}): {lib().amos_last_error().decode()}")
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def geometry_probe(max_width, max_height, width, height, n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7):
    """Host-only: (rc, need[6], cap[6]) of amos_orb_geometry_probe -- no device is touched."""
    params = OrbParams(n_features, scale_factor, n_levels, ini_th, min_th)
    need, cap = np.zeros(6, np.int32), np.zeros(6, np.int32)
    rc = lib().amos_orb_geometry_probe(C.byref(params), C.c_int(max_width), C.c_int(max_height), C.c_int(width), C.c_int(height),
                                       _p(need), _p(cap))
    return rc, need, cap


def device_count():
    return _check(lib().amos_device_count(), "amos_device_count")


class OrbExtractor:
    """Mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:93-168) over the C ABI.

    detect()            = 3-arg operator()      (ORBextractor.cc:1672)
    gate()              = MovingKeyPoints       (ORBextractor.cc:1688)
    describe()          = ProcessDesp           (ORBextractor.cc:1747)
    extract()           = 4-arg operator()      (ORBextractor.cc:1544)
    """

    def __init__(self, n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7,
                 max_width=640, max_height=480, max_batch=1, device=0, stream=None):
        self.L = lib()
        self.params = OrbParams(n_features, scale_factor, n_levels, ini_th, min_th)
        self.n_levels, self.n_features = n_levels, n_features
        self.max_batch = max_batch
        h = C.c_void_p()
        _check(self.L.amos_orb_create(C.byref(self.params), C.c_int(max_width), C.c_int(max_height),
                                      C.c_int(max_batch), C.c_int(device), C.c_void_p(stream), C.byref(h)),
               "amos_orb_create")
        self.h = h
        self.shape = None
        cap = C.c_int(0)
        self.L.amos_orb_batch_results_device(self.h, None, None, None, C.byref(cap))
        self.capacity = cap.value

    def close(self):
        if getattr(self, "h", None):
            self.L.amos_orb_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # -- a1
    def tables(self):
        n = self.n_levels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl, umax = np.zeros(n, np.int32), np.zeros(16, np.int32)
        _check(self.L.amos_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(fpl), _p(umax)), "amos_orb_tables")
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=umax)

    def level_sizes(self, width, height):
        lw, lh = np.zeros(self.n_levels, np.int32), np.zeros(self.n_levels, np.int32)
        _check(self.L.amos_orb_level_sizes(self.h, C.c_int(width), C.c_int(height), _p(lw), _p(lh)), "amos_orb_level_sizes")
        return lw, lh

    # -- a7
    def detect(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        self.shape = (h, w)
        _check(self.L.amos_orb_detect(self.h, _p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h)), "amos_orb_detect")

    def level_keypoints(self, level, frame=0):
        n = _check(self.L.amos_orb_level_count(self.h, C.c_int(frame), C.c_int(level)), "amos_orb_level_count")
        out = np.zeros(max(n, 1), KP_DTYPE)
        _check(self.L.amos_orb_level_keypoints(self.h, C.c_int(frame), C.c_int(level), _p(out), C.c_int(len(out))),
               "amos_orb_level_keypoints")
        return out[:n]

    def set_level_keypoints(self, level, kps, frame=0):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        _check(self.L.amos_orb_set_level_keypoints(self.h, C.c_int(frame), C.c_int(level), _p(kps), C.c_int(len(kps))),
               "amos_orb_set_level_keypoints")

    def level_candidates(self, level, frame=0, cap=1 << 20):
        out = np.zeros(cap, KP_DTYPE)
        n = _check(self.L.amos_orb_level_candidates(self.h, C.c_int(frame), C.c_int(level), _p(out), C.c_int(cap)),
                   "amos_orb_level_candidates")
        return out[:n].copy()

    def level_image(self, level, padded=False, frame=0):
        lw, lh = self.level_sizes(self.shape[1], self.shape[0])
        w, h = int(lw[level]), int(lh[level])
        if padded:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        _check(self.L.amos_orb_level_image(self.h, C.c_int(frame), C.c_int(level), _p(out), C.c_size_t(w),
                                           C.c_int(int(padded))), "amos_orb_level_image")
        return out

    def blurred_image(self, level, frame=0):
        lw, lh = self.level_sizes(self.shape[1], self.shape[0])
        out = np.zeros((int(lh[level]), int(lw[level])), np.uint8)
        _check(self.L.amos_orb_blurred_image(self.h, C.c_int(frame), C.c_int(level), _p(out), C.c_size_t(out.shape[1])),
               "amos_orb_blurred_image")
        return out

    # -- a8
    def gate(self, mask, labels=None, center_ids=None, rm_vector=None):
        mask = np.ascontiguousarray(mask, np.uint8)
        removed = np.zeros(self.capacity + 1, KP_DTYPE)
        nrem = C.c_int(0)
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.float64)
            center_ids = np.ascontiguousarray(center_ids, np.int32)
            rm_vector = np.ascontiguousarray(rm_vector, np.int32)
            rc = self.L.amos_orb_gate(self.h, _p(mask), C.c_size_t(mask.strides[0]), _p(labels),
                                      C.c_size_t(labels.shape[1]), _p(center_ids), C.c_int(len(center_ids)),
                                      _p(rm_vector), C.c_int(len(rm_vector)), _p(removed), C.c_int(len(removed)),
                                      C.byref(nrem))
        else:
            rc = self.L.amos_orb_gate(self.h, _p(mask), C.c_size_t(mask.strides[0]), None, C.c_size_t(0), None,
                                      C.c_int(0), None, C.c_int(0), _p(removed), C.c_int(len(removed)), C.byref(nrem))
        _check(rc, "amos_orb_gate")
        return removed[:nrem.value].copy()

    def closed_mask(self):
        out = np.zeros(self.shape, np.uint8)
        _check(self.L.amos_orb_closed_mask(self.h, _p(out), C.c_size_t(out.shape[1])), "amos_orb_closed_mask")
        return out

    # -- a9 / a11
    def describe(self):
        kps = np.zeros(self.capacity + 1, KP_DTYPE)
        desc = np.zeros((self.capacity + 1, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.amos_orb_describe(self.h, _p(kps), _p(desc), C.c_int(len(kps)), C.byref(n)), "amos_orb_describe")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        self.shape = (h, w)
        kps = np.zeros(self.capacity + 1, KP_DTYPE)
        desc = np.zeros((self.capacity + 1, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.amos_orb_extract(self.h, _p(gray), C.c_size_t(gray.strides[0]), C.c_int(w), C.c_int(h), _p(kps),
                                       _p(desc), C.c_int(len(kps)), C.byref(n)), "amos_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    # -- batched, device resident
    def extract_batch_device(self, d_ptr, frame_stride, row_stride, width, height, n_frames):
        """d_ptr: integer device address of n_frames gray frames.  Asynchronous on the handle's stream."""
        self.shape = (height, width)
        _check(self.L.amos_orb_extract_batch_device(self.h, C.c_void_p(d_ptr), C.c_size_t(frame_stride),
                                                    C.c_size_t(row_stride), C.c_int(width), C.c_int(height),
                                                    C.c_int(n_frames)), "amos_orb_extract_batch_device")

    def detect_batch_device(self, d_ptr, frame_stride, row_stride, width, height, n_frames):
        self.shape = (height, width)
        _check(self.L.amos_orb_detect_batch_device(self.h, C.c_void_p(d_ptr), C.c_size_t(frame_stride), C.c_size_t(row_stride),
                                                   C.c_int(width), C.c_int(height), C.c_int(n_frames)), "amos_orb_detect_batch_device")

    def gate_batch_device(self, d_masks, mask_frame_stride, mask_row_stride):
        _check(self.L.amos_orb_gate_batch_device(self.h, C.c_void_p(d_masks), C.c_size_t(mask_frame_stride),
                                                 C.c_size_t(mask_row_stride)), "amos_orb_gate_batch_device")

    def describe_batch_device(self):
        _check(self.L.amos_orb_describe_batch_device(self.h), "amos_orb_describe_batch_device")

    def extract_batch_device_color(self, d_ptr, frame_stride, row_stride, width, height, n_frames, channels=3, rgb_order=False):
        """cvtColor(BGR/RGB[A] -> gray) fused into the level-0 import (Tracking.cc:308-321)."""
        self.shape = (height, width)
        _check(self.L.amos_orb_extract_batch_device_color(self.h, C.c_void_p(d_ptr), C.c_size_t(frame_stride), C.c_size_t(row_stride),
                                                          C.c_int(width), C.c_int(height), C.c_int(n_frames), C.c_int(channels),
                                                          C.c_int(int(rgb_order))), "amos_orb_extract_batch_device_color")

    def detect_color_with_mask_pre_batch_device(self, pre, d_ptr, frame_stride, row_stride, width, height, n_frames, d_net_input, channels=3,
                                                rgb_order=False):
        """8f-4: one read of the colour frames -> padded gray level 0 (+ the rest of detect) and the mask network's input tensor."""
        self.shape = (height, width)
        _check(self.L.amos_orb_detect_color_with_mask_pre_batch_device(self.h, pre.p, C.c_void_p(d_ptr), C.c_size_t(frame_stride), C.c_size_t(row_stride),
                                                                       C.c_int(width), C.c_int(height), C.c_int(n_frames), C.c_int(channels),
                                                                       C.c_int(int(rgb_order)), C.c_void_p(d_net_input)),
               "amos_orb_detect_color_with_mask_pre_batch_device")

    def rgbd_glue_batch_device(self, d_depth, depth_is_u16, depth_map_factor, depth_frame_stride, depth_row_stride, mbf, bounds,
                               d_u_right, d_depth_out, d_grid_cell, d_kps_un=None):
        """ComputeStereoFromRGBD + grid cell of every keypoint of the last batch (Frame.cc:1576-1615, 1007-1030)."""
        _check(self.L.amos_frame_rgbd_glue_batch_device(self.h, C.c_void_p(d_depth), C.c_int(int(depth_is_u16)), C.c_float(depth_map_factor),
                                                        C.c_size_t(depth_frame_stride), C.c_size_t(depth_row_stride), C.c_float(mbf),
                                                        C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]),
                                                        C.c_void_p(d_kps_un), C.c_void_p(d_u_right), C.c_void_p(d_depth_out),
                                                        C.c_void_p(d_grid_cell)),
               "amos_frame_rgbd_glue_batch_device")

    def undistort_batch_device(self, fx, fy, cx, cy, dist_coef, d_kps_un):
        """Frame::UndistortKeyPoints for every keypoint of the last batch (Frame.cc:1052-1118)."""
        dc = np.ascontiguousarray(dist_coef, np.float32)
        _check(self.L.amos_frame_undistort_batch_device(self.h, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(dc),
                                                        C.c_int(len(dc)), C.c_void_p(d_kps_un)), "amos_frame_undistort_batch_device")

    def batch_results_device(self):
        kps, desc, cnt, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        _check(self.L.amos_orb_batch_results_device(self.h, C.byref(kps), C.byref(desc), C.byref(cnt), C.byref(cap)),
               "amos_orb_batch_results_device")
        return kps.value, desc.value, cnt.value, cap.value

    def batch_fetch(self, frame):
        kps = np.zeros(self.capacity + 1, KP_DTYPE)
        desc = np.zeros((self.capacity + 1, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.amos_orb_batch_fetch(self.h, C.c_int(frame), _p(kps), _p(desc), C.c_int(len(kps)), C.byref(n)),
               "amos_orb_batch_fetch")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def sync(self):
        _check(self.L.amos_orb_sync(self.h), "amos_orb_sync")

    def pyramid_launches(self):
        """launches of the resize kernel(s) per pass (the "pyramid" stage of timing_collect)"""
        return self.n_levels - 1

    STAGES = ("import", "pyramid", "fast", "octree", "orient", "blur", "describe")

    def timing_enable(self, max_records):
        _check(self.L.amos_orb_timing_enable(self.h, C.c_int(max_records)), "amos_orb_timing_enable")

    def timing_collect(self):
        """Average milliseconds per stage over the passes recorded since the last collect."""
        ms = np.zeros(len(self.STAGES), np.float32)
        n = C.c_int(0)
        _check(self.L.amos_orb_timing_collect(self.h, _p(ms), C.byref(n)), "amos_orb_timing_collect")
        return dict(zip(self.STAGES, ms.tolist())), n.value

    @property
    def stream(self):
        return self.L.amos_orb_stream(self.h)


SLIC_CENTER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("L", "<i4"), ("A", "<i4"), ("B", "<i4"), ("D", "<i4"), ("label", "<i4"), ("id", "<i4")])


class Slic:
    """cluster::SLIC of the reference from the Lab image on (src/cluster.cc:300-343): superpixel label map + centres."""

    def __init__(self, max_width=640, max_height=480, max_batch=1, device=0, stream=None):
        self.L = lib()
        s = C.c_void_p()
        _check(self.L.amos_slic_create(C.c_int(device), C.c_void_p(stream), C.c_int(max_width), C.c_int(max_height), C.c_int(max_batch),
                                       C.byref(s)), "amos_slic_create")
        self.s = s
        self.L.amos_slic_stream.restype = C.c_void_p

    def close(self):
        if getattr(self, "s", None):
            self.L.amos_slic_destroy(self.s)
            self.s = None

    def __del__(self):
        self.close()

    @staticmethod
    def center_count(width, height, length=5):
        nx, ny = C.c_int(0), C.c_int(0)
        n = lib().amos_slic_center_count(C.c_int(width), C.c_int(height), C.c_int(length), C.byref(nx), C.byref(ny))
        return n, nx.value, ny.value

    def run(self, lab, depth, length=5, m=10, iterations=5):
        lab = np.ascontiguousarray(lab, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        h, w = depth.shape
        assert lab.shape == (h, w, 3)
        n = self.center_count(w, h, length)[0]
        labels = np.zeros((h, w), np.float64)
        centers = np.zeros(max(n, 1), SLIC_CENTER_DTYPE)
        nc = C.c_int(0)
        _check(self.L.amos_slic_run(self.s, _p(lab), _p(depth), C.c_int(w), C.c_int(h), C.c_int(length), C.c_int(m), C.c_int(iterations),
                                    _p(labels), _p(centers), C.byref(nc)), "amos_slic_run")
        return labels, centers[:nc.value]

    def run_batch_device(self, d_lab, d_depth, width, height, n_frames, d_labels, d_centers, length=5, m=10, iterations=5):
        _check(self.L.amos_slic_batch_device(self.s, C.c_void_p(d_lab), C.c_void_p(d_depth), C.c_int(width), C.c_int(height), C.c_int(n_frames),
                                             C.c_int(length), C.c_int(m), C.c_int(iterations), C.c_void_p(d_labels), C.c_void_p(d_centers)),
               "amos_slic_batch_device")

    def bgr2lab_batch_device(self, d_bgr, n_pixels, d_lab, rgb_order=False):
        """cv::cvtColor(COLOR_BGR2Lab), 8-bit (cluster.cc:310)."""
        _check(self.L.amos_cluster_bgr2lab_batch_device(self.s, C.c_void_p(d_bgr), C.c_size_t(n_pixels), C.c_int(int(rgb_order)), C.c_void_p(d_lab)),
               "amos_cluster_bgr2lab_batch_device")

    def kmeans(self, centers, k=15, seed=1, max_iter=1000):
        """cluster::randCent + kmeans on the SLIC centres (cluster.cc:353-460, seeded): returns (centres with .id set, passes)."""
        c = np.ascontiguousarray(centers, SLIC_CENTER_DTYPE).copy()
        passes = C.c_int(0)
        _check(self.L.amos_cluster_kmeans(self.s, _p(c), C.c_int(len(c)), C.c_int(k), C.c_uint32(seed), C.c_int(max_iter), C.byref(passes)),
               "amos_cluster_kmeans")
        return c, passes.value

    def kmeans_batch_device(self, d_centers, n_centers, n_frames, k=15, seed=1, max_iter=1000, d_passes=None):
        _check(self.L.amos_cluster_kmeans_batch_device(self.s, C.c_void_p(d_centers), C.c_int(n_centers), C.c_int(n_frames), C.c_int(k),
                                                       C.c_uint32(seed), C.c_int(max_iter), C.c_void_p(d_passes)), "amos_cluster_kmeans_batch_device")

    def sync(self):
        import torch  # noqa: F401  (the HIP runtime is torch's)
        st = self.L.amos_slic_stream(self.s)
        torch.cuda.ExternalStream(st).synchronize()


class MaskPreprocessor:
    """amos_mask_pre_*: BGR frames -> the mask network's [n, 3, 550, 550] input tensor in three HIP kernels
    (yolact.cc:220, 385-451; yolact_interface.py:862-866; utils/augmentations.py:616-657)."""

    def __init__(self, width=640, height=480, max_batch=16, device=0, stream=None):
        self.L = lib()
        p = C.c_void_p()
        _check(self.L.amos_mask_pre_create(C.c_int(device), C.c_void_p(stream), C.c_int(width), C.c_int(height), C.c_int(max_batch),
                                           C.byref(p)), "amos_mask_pre_create")
        self.p, self.max_batch, self.shape = p, max_batch, (height, width)
        self.L.amos_mask_pre_stream.restype = C.c_void_p
        self.stream_ptr = self.L.amos_mask_pre_stream(self.p)

    def close(self):
        if getattr(self, "p", None):
            self.L.amos_mask_pre_destroy(self.p)
            self.p = None

    def __del__(self):
        self.close()

    def run(self, d_bgr, n_frames, d_out):
        _check(self.L.amos_mask_preprocess_batch_device(self.p, C.c_void_p(d_bgr), C.c_int(n_frames), C.c_void_p(d_out)),
               "amos_mask_preprocess_batch_device")


def mask_bias_act(stream_ptr, y_ptr, bias_ptr, residual_ptr, n, channels, relu):
    """amos_mask_bias_act_device: y = act((y + bias[c]) + residual) in place on an NHWC float32 tensor (device pointers)."""
    _check(lib().amos_mask_bias_act_device(C.c_void_p(stream_ptr), C.c_void_p(y_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr), C.c_size_t(n),
                                           C.c_int(channels), C.c_int(int(relu))), "amos_mask_bias_act_device")


class SceneFlowCamera(C.Structure):
    """amos_scene_flow_camera (include/amos_frontend.h)."""
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("invfx", C.c_float), ("invfy", C.c_float), ("Tlw", C.c_float * 12), ("Rwc", C.c_float * 9),
                ("Ow", C.c_float * 3)]


def flow_check(stream, d_last, last_stride, d_cur, cur_stride, cols, rows, d_pre, d_next, d_state_in, n, d_state_out):
    _check(lib().amos_flow_check_device(C.c_void_p(stream), C.c_void_p(d_last), C.c_size_t(last_stride), C.c_void_p(d_cur), C.c_size_t(cur_stride),
                                        C.c_int(cols), C.c_int(rows), C.c_void_p(d_pre), C.c_void_p(d_next), C.c_void_p(d_state_in), C.c_int(n),
                                        C.c_void_p(d_state_out)), "amos_flow_check_device")


def flow_epipolar(stream, d_F, d_pre, d_next, d_state, n, d_dd):
    _check(lib().amos_flow_epipolar_device(C.c_void_p(stream), C.c_void_p(d_F), C.c_void_p(d_pre), C.c_void_p(d_next), C.c_void_p(d_state), C.c_int(n),
                                           C.c_void_p(d_dd)), "amos_flow_epipolar_device")


def flow_scene_flow(stream, d_depth_last, last_stride, d_depth_cur, cur_stride, d_match_pre, d_match_cur, n, cam, d_out):
    _check(lib().amos_flow_scene_flow_device(C.c_void_p(stream), C.c_void_p(d_depth_last), C.c_size_t(last_stride), C.c_void_p(d_depth_cur),
                                             C.c_size_t(cur_stride), C.c_void_p(d_match_pre), C.c_void_p(d_match_cur), C.c_int(n), C.byref(cam),
                                             C.c_void_p(d_out)), "amos_flow_scene_flow_device")


def flow_fundamental_score(stream, d_F, n_hyp, d_p1, d_p2, n, threshold, d_err, d_inliers, d_mask):
    """amos_flow_fundamental_score_device: error / inlier test / inlier count of n correspondences under n_hyp fundamental matrices (device pointers)."""
    _check(lib().amos_flow_fundamental_score_device(C.c_void_p(stream), C.c_void_p(d_F), C.c_int(n_hyp), C.c_void_p(d_p1), C.c_void_p(d_p2), C.c_int(n),
                                                    C.c_double(threshold), C.c_void_p(d_err), C.c_void_p(d_inliers), C.c_void_p(d_mask)),
           "amos_flow_fundamental_score_device")


def flow_pnp_score(stream, d_Rt, n_hyp, d_obj, d_img, n, fx, fy, cx, cy, reprojection_error, d_err, d_inliers, d_mask):
    """amos_flow_pnp_score_device: reprojection error / inlier test / inlier count of n 3-D -> 2-D correspondences under n_hyp poses (device pointers)."""
    _check(lib().amos_flow_pnp_score_device(C.c_void_p(stream), C.c_void_p(d_Rt), C.c_int(n_hyp), C.c_void_p(d_obj), C.c_void_p(d_img), C.c_int(n), C.c_double(fx),
                                            C.c_double(fy), C.c_double(cx), C.c_double(cy), C.c_double(reprojection_error), C.c_void_p(d_err), C.c_void_p(d_inliers),
                                            C.c_void_p(d_mask)), "amos_flow_pnp_score_device")


def mask_bias_relu_maxpool(stream_ptr, x_ptr, bias_ptr, y_ptr, n, in_h, in_w, channels):
    """amos_mask_bias_relu_maxpool_device: max_pool2d(relu(x + bias), 3, 2, 1) of an NHWC float32 tensor in one pass (device pointers)."""
    _check(lib().amos_mask_bias_relu_maxpool_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(bias_ptr), C.c_void_p(y_ptr), C.c_int(n), C.c_int(in_h),
                                                    C.c_int(in_w), C.c_int(channels)), "amos_mask_bias_relu_maxpool_device")


def mask_stem_weight_floats():
    return int(lib().amos_mask_stem_weight_floats())


def mask_stem_weights(stream_ptr, w_ptr, w_strides, packed_ptr):
    """amos_mask_stem_weights_device: the [64][3][7][7] stem weight (element strides: out channel, in channel, row, column) -> the kernel's layout."""
    sn, sc, sy, sx = (int(v) for v in w_strides)
    _check(lib().amos_mask_stem_weights_device(C.c_void_p(stream_ptr), C.c_void_p(w_ptr), C.c_longlong(sn), C.c_longlong(sc), C.c_longlong(sy), C.c_longlong(sx),
                                               C.c_void_p(packed_ptr)), "amos_mask_stem_weights_device")


def mask_stem(stream_ptr, x_ptr, x_strides, packed_ptr, bias_ptr, y_ptr, batch, height, width):
    """amos_mask_stem_device: conv 7 x 7 / 2 (3 -> 64) + bias + ReLU + max-pool 3 x 3 / 2 in one kernel; x float32 [batch][3][height][width]
    through its element strides, y channels-last [batch][ph][pw][64] (device pointers)."""
    sb, sc, sy, sx = (int(v) for v in x_strides)
    _check(lib().amos_mask_stem_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_longlong(sb), C.c_longlong(sc), C.c_longlong(sy), C.c_longlong(sx),
                                       C.c_void_p(packed_ptr), C.c_void_p(bias_ptr), C.c_void_p(y_ptr), C.c_int(batch), C.c_int(height), C.c_int(width)),
           "amos_mask_stem_device")


def mask_bilinear_x2_mode(mode=-1):
    """amos_mask_bilinear_x2_mode: 1 = exact x 2 enlargements take the 2 x 2-outputs-per-thread kernel (default), 0 = never; returns the previous mode."""
    return int(lib().amos_mask_bilinear_x2_mode(C.c_int(mode)))


def mask_conv1x1_supported(cin, cout, stride):
    return lib().amos_mask_conv1x1_supported(C.c_int(cin), C.c_int(cout), C.c_int(stride)) == 0


def mask_conv1x1(stream_ptr, x_ptr, w_ptr, bias_ptr, residual_ptr, y_ptr, batch, in_h, in_w, cin, cout, stride, relu):
    """amos_mask_conv1x1_device: a 1 x 1 convolution + bias (+ residual) (+ ReLU) on NHWC float32 tensors (device pointers)."""
    _check(lib().amos_mask_conv1x1_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr),
                                          C.c_void_p(y_ptr), C.c_int(batch), C.c_int(in_h), C.c_int(in_w), C.c_int(cin), C.c_int(cout), C.c_int(stride),
                                          C.c_int(int(relu))), "amos_mask_conv1x1_device")


def mask_conv_supported(cin, cout, kh, kw, stride, pad):
    return lib().amos_mask_conv_supported(C.c_int(cin), C.c_int(cout), C.c_int(kh), C.c_int(kw), C.c_int(stride), C.c_int(pad)) == 0


def mask_winograd_supported(cin, cout):
    return lib().amos_mask_winograd_supported(C.c_int(cin), C.c_int(cout)) == 0


def mask_winograd_weights(stream_ptr, w_ptr, u_ptr, cin, cout):
    """amos_mask_winograd_weights_device: weight [cout][3][3][cin] -> transformed weight (16 * cin * cout floats at u_ptr)."""
    _check(lib().amos_mask_winograd_weights_device(C.c_void_p(stream_ptr), C.c_void_p(w_ptr), C.c_void_p(u_ptr), C.c_int(cin), C.c_int(cout)),
           "amos_mask_winograd_weights_device")


def mask_winograd_conv(stream_ptr, x_ptr, u_ptr, bias_ptr, residual_ptr, y_ptr, batch, h, w, cin, cout, relu):
    """amos_mask_winograd_conv_device: 3 x 3 / stride 1 / pad 1 convolution + bias (+ residual) (+ ReLU) on NHWC float32 tensors."""
    _check(lib().amos_mask_winograd_conv_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(u_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr),
                                                C.c_void_p(y_ptr), C.c_int(batch), C.c_int(h), C.c_int(w), C.c_int(cin), C.c_int(cout), C.c_int(int(relu))),
           "amos_mask_winograd_conv_device")


def mask_winograd24_weights(stream_ptr, w_ptr, u_ptr, cin, cout):
    """amos_mask_winograd24_weights_device: weight [cout][3][3][cin] -> G2 g G4^T (24 * cin * cout floats at u_ptr), F(2 x 4, 3 x 3)."""
    _check(lib().amos_mask_winograd24_weights_device(C.c_void_p(stream_ptr), C.c_void_p(w_ptr), C.c_void_p(u_ptr), C.c_int(cin), C.c_int(cout)),
           "amos_mask_winograd24_weights_device")


def mask_winograd24_conv(stream_ptr, x_ptr, u_ptr, bias_ptr, residual_ptr, y_ptr, batch, h, w, cin, cout, relu):
    """amos_mask_winograd24_conv_device: the same convolution as mask_winograd_conv as Winograd F(2 x 4, 3 x 3) (its own weight layout)."""
    _check(lib().amos_mask_winograd24_conv_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(u_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr),
                                                  C.c_void_p(y_ptr), C.c_int(batch), C.c_int(h), C.c_int(w), C.c_int(cin), C.c_int(cout), C.c_int(int(relu))),
           "amos_mask_winograd24_conv_device")


def mask_winograd24_conv_layout(stream_ptr, x_ptr, u_ptr, bias_ptr, residual_ptr, y_ptr, batch, h, w, cin, cout, relu, in_blocked, out_blocked):
    """amos_mask_winograd24_conv_layout_device: mask_winograd24_conv with the channel-blocked layout [batch][c / 8][h][w][8] on the input
    (in_blocked) and / or on the output and residual (out_blocked); False = channels-last."""
    _check(lib().amos_mask_winograd24_conv_layout_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(u_ptr), C.c_void_p(bias_ptr),
                                                         C.c_void_p(residual_ptr), C.c_void_p(y_ptr), C.c_int(batch), C.c_int(h), C.c_int(w), C.c_int(cin),
                                                         C.c_int(cout), C.c_int(int(relu)), C.c_int(int(in_blocked)), C.c_int(int(out_blocked))),
           "amos_mask_winograd24_conv_layout_device")


def mask_winograd24_persistent_mode(mode=-2):
    """amos_mask_winograd24_persistent_mode: -1 by launch size, 0 one work-group per id, 1 persistent; returns the previous mode (-2: query only)."""
    return int(lib().amos_mask_winograd24_persistent_mode(C.c_int(mode)))


def mask_winograd24_narrow_mode(mode=-2):
    """amos_mask_winograd24_narrow_mode: -1 by launch size, 0 64 output channels per work-group, 1 32; returns the previous mode (-2: query only)."""
    return int(lib().amos_mask_winograd24_narrow_mode(C.c_int(mode)))


def mask_conv_tile_mode(mode=-2):
    """amos_mask_conv_tile_mode: -1 automatic, 0 wide (128 x 128), 1 narrow (128 x 64); returns the previous mode (-2: query only)."""
    return int(lib().amos_mask_conv_tile_mode(C.c_int(mode)))


def mask_conv_kernel_name(batch, in_h, in_w, cin, cout, kh, kw, stride, pad):
    buf = C.create_string_buffer(96)
    _check(lib().amos_mask_conv_kernel_name(C.c_int(batch), C.c_int(in_h), C.c_int(in_w), C.c_int(cin), C.c_int(cout), C.c_int(kh), C.c_int(kw),
                                            C.c_int(stride), C.c_int(pad), buf, C.c_int(96)), "amos_mask_conv_kernel_name")
    return buf.value.decode()


def mask_conv(stream_ptr, x_ptr, w_ptr, bias_ptr, residual_ptr, y_ptr, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, relu):
    """amos_mask_conv_device: convolution + bias (+ residual) (+ ReLU) on NHWC float32 tensors, weight [cout][kh][kw][cin] (device pointers)."""
    _check(lib().amos_mask_conv_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr),
                                       C.c_void_p(y_ptr), C.c_int(batch), C.c_int(in_h), C.c_int(in_w), C.c_int(cin), C.c_int(cout), C.c_int(kh), C.c_int(kw),
                                       C.c_int(stride), C.c_int(pad), C.c_int(int(relu))), "amos_mask_conv_device")


def mask_conv_workspace_bytes(batch, in_h, in_w, cin, cout, kh, kw, stride, pad):
    """amos_mask_conv_workspace_bytes: scratch mask_conv_ws wants for its split-K plan of this shape; 0 = the plan is an ordinary launch."""
    fn = lib().amos_mask_conv_workspace_bytes
    fn.restype = C.c_size_t
    return int(fn(C.c_int(batch), C.c_int(in_h), C.c_int(in_w), C.c_int(cin), C.c_int(cout), C.c_int(kh), C.c_int(kw), C.c_int(stride), C.c_int(pad)))


def mask_conv_ws(stream_ptr, x_ptr, w_ptr, bias_ptr, residual_ptr, y_ptr, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, relu, workspace_ptr, workspace_bytes):
    """amos_mask_conv_ws_device: mask_conv with split-K for small launches (workspace: mask_conv_workspace_bytes; its first 16 KB zero before the
    first use and left zero; launches sharing a workspace ordered on one stream)."""
    _check(lib().amos_mask_conv_ws_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), C.c_void_p(residual_ptr),
                                          C.c_void_p(y_ptr), C.c_int(batch), C.c_int(in_h), C.c_int(in_w), C.c_int(cin), C.c_int(cout), C.c_int(kh), C.c_int(kw),
                                          C.c_int(stride), C.c_int(pad), C.c_int(int(relu)), C.c_void_p(workspace_ptr), C.c_size_t(workspace_bytes)),
           "amos_mask_conv_ws_device")


def mask_bilinear_nhwc(stream_ptr, x_ptr, y_ptr, n, in_h, in_w, out_h, out_w, channels, scale_h, scale_w, relu=False):
    _check(lib().amos_mask_bilinear_nhwc_act_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_int(n), C.c_int(in_h), C.c_int(in_w),
                                                    C.c_int(out_h), C.c_int(out_w), C.c_int(channels), C.c_float(scale_h), C.c_float(scale_w), C.c_int(int(relu))),
           "amos_mask_bilinear_nhwc_act_device")


def mask_class_scores(stream_ptr, conf_ptr, scores_ptr, batch, n_priors, n_classes_with_background, threshold):
    _check(lib().amos_mask_class_scores_device(C.c_void_p(stream_ptr), C.c_void_p(conf_ptr), C.c_void_p(scores_ptr), C.c_int(batch), C.c_int(n_priors),
                                               C.c_int(n_classes_with_background), C.c_float(threshold)), "amos_mask_class_scores_device")


def mask_person_mask(stream_ptr, masks_ptr, flags_ptr, out_ptr, batch, n_det, mask_h, mask_w, out_h, out_w):
    _check(lib().amos_mask_person_mask_device(C.c_void_p(stream_ptr), C.c_void_p(masks_ptr), C.c_void_p(flags_ptr), C.c_void_p(out_ptr), C.c_int(batch),
                                              C.c_int(n_det), C.c_int(mask_h), C.c_int(mask_w), C.c_int(out_h), C.c_int(out_w)), "amos_mask_person_mask_device")


def mask_head_outputs(stream_ptr, raw_ptr, bias_ptr, loc_ptr, conf_ptr, coef_ptr, batch, cells, channels_padded, anchors, n_classes_with_background, mask_dim,
                      n_priors_total, prior_offset):
    _check(lib().amos_mask_head_outputs_device(C.c_void_p(stream_ptr), C.c_void_p(raw_ptr), C.c_void_p(bias_ptr), C.c_void_p(loc_ptr), C.c_void_p(conf_ptr),
                                               C.c_void_p(coef_ptr), C.c_int(batch), C.c_int(cells), C.c_int(channels_padded), C.c_int(anchors),
                                               C.c_int(n_classes_with_background), C.c_int(mask_dim), C.c_int(n_priors_total), C.c_int(prior_offset)),
           "amos_mask_head_outputs_device")


def mask_head_outputs_scores(stream_ptr, raw_ptr, bias_ptr, loc_ptr, conf_ptr, coef_ptr, scores_ptr, threshold, batch, cells, channels_padded, anchors,
                             n_classes_with_background, mask_dim, n_priors_total, prior_offset):
    """amos_mask_head_outputs_scores_device: mask_head_outputs + Detect's class scores [batch][classes][n_priors_total] from the same kernel; conf_ptr may be
    None (the softmax tensor is then not written)."""
    _check(lib().amos_mask_head_outputs_scores_device(C.c_void_p(stream_ptr), C.c_void_p(raw_ptr), C.c_void_p(bias_ptr), C.c_void_p(loc_ptr), C.c_void_p(conf_ptr),
                                                      C.c_void_p(coef_ptr), C.c_void_p(scores_ptr), C.c_float(threshold), C.c_int(batch), C.c_int(cells),
                                                      C.c_int(channels_padded), C.c_int(anchors), C.c_int(n_classes_with_background), C.c_int(mask_dim),
                                                      C.c_int(n_priors_total), C.c_int(prior_offset)), "amos_mask_head_outputs_scores_device")


def mask_person_masks_scores(stream_ptr, loc_ptr, scores_ptr, coef_ptr, priors_ptr, proto_ptr, batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w,
                             out_h, out_w, workspace_ptr, workspace_bytes, masks_ptr, found_ptr):
    """amos_mask_person_masks_scores_device: mask_person_masks from the class scores mask_head_outputs_scores wrote instead of the softmax tensor."""
    _check(lib().amos_mask_person_masks_scores_device(C.c_void_p(stream_ptr), C.c_void_p(loc_ptr), C.c_void_p(scores_ptr), C.c_void_p(coef_ptr),
                                                      C.c_void_p(priors_ptr), C.c_void_p(proto_ptr), C.c_int(batch), C.c_int(n_priors),
                                                      C.c_int(n_classes_with_background), C.c_int(mask_dim), C.c_int(proto_h), C.c_int(proto_w), C.c_int(out_h),
                                                      C.c_int(out_w), C.c_void_p(workspace_ptr), C.c_size_t(workspace_bytes), C.c_void_p(masks_ptr),
                                                      C.c_void_p(found_ptr)), "amos_mask_person_masks_scores_device")


def mask_topk_rows(stream_ptr, x_ptr, values_ptr, indices_ptr, rows, n, k):
    _check(lib().amos_mask_topk_rows_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(values_ptr), C.c_void_p(indices_ptr), C.c_int(rows), C.c_int(n),
                                            C.c_int(k)), "amos_mask_topk_rows_device")


def mask_topk_rows_sparse(stream_ptr, x_ptr, values_ptr, indices_ptr, rows, n, k, fill):
    """amos_mask_topk_rows_sparse_device: mask_topk_rows for rows that are mostly `fill` (one scan instead of five; the same result)."""
    _check(lib().amos_mask_topk_rows_sparse_device(C.c_void_p(stream_ptr), C.c_void_p(x_ptr), C.c_void_p(values_ptr), C.c_void_p(indices_ptr), C.c_int(rows),
                                                   C.c_int(n), C.c_int(k), C.c_float(fill)), "amos_mask_topk_rows_sparse_device")


def mask_post_workspace_bytes(batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w):
    """amos_mask_post_workspace_bytes: device scratch mask_person_masks needs for these sizes."""
    fn = lib().amos_mask_post_workspace_bytes
    fn.restype = C.c_size_t
    return int(fn(C.c_int(batch), C.c_int(n_priors), C.c_int(n_classes_with_background), C.c_int(mask_dim), C.c_int(proto_h), C.c_int(proto_w)))


def mask_person_masks(stream_ptr, loc_ptr, conf_ptr, coef_ptr, priors_ptr, proto_ptr, batch, n_priors, n_classes_with_background, mask_dim, proto_h, proto_w,
                      out_h, out_w, workspace_ptr, workspace_bytes, masks_ptr, found_ptr):
    """amos_mask_person_masks_device: Detect + postprocess + prep_display of the static-shape batch path in seven launches (device pointers)."""
    _check(lib().amos_mask_person_masks_device(C.c_void_p(stream_ptr), C.c_void_p(loc_ptr), C.c_void_p(conf_ptr), C.c_void_p(coef_ptr), C.c_void_p(priors_ptr),
                                               C.c_void_p(proto_ptr), C.c_int(batch), C.c_int(n_priors), C.c_int(n_classes_with_background), C.c_int(mask_dim),
                                               C.c_int(proto_h), C.c_int(proto_w), C.c_int(out_h), C.c_int(out_w), C.c_void_p(workspace_ptr),
                                               C.c_size_t(workspace_bytes), C.c_void_p(masks_ptr), C.c_void_p(found_ptr)), "amos_mask_person_masks_device")


def mask_nms_column_max(stream_ptr, boxes_ptr, out_ptr, n_lists, k):
    """amos_mask_nms_column_max_device: out[list][j] = max_{i < j} IoU(box i, box j) for score-sorted box lists (device pointers)."""
    _check(lib().amos_mask_nms_column_max_device(C.c_void_p(stream_ptr), C.c_void_p(boxes_ptr), C.c_void_p(out_ptr), C.c_int(n_lists), C.c_int(k)),
           "amos_mask_nms_column_max_device")


class LkTracker:
    """cv::calcOpticalFlowPyrLK on given points (Tracking.cc:896: 22 x 22 window, 5 levels, 20 iterations / 0.01), device resident."""

    def __init__(self, width=640, height=480, win_size=22, max_level=5, device=0, stream=None):
        self.L = lib()
        k = C.c_void_p()
        _check(self.L.amos_lk_create(C.c_int(device), C.c_void_p(stream), C.c_int(width), C.c_int(height), C.c_int(win_size), C.c_int(max_level), C.byref(k)),
               "amos_lk_create")
        self.k = k
        self.L.amos_lk_stream.restype = C.c_void_p
        self.L.amos_lk_stream.argtypes = [C.c_void_p]
        self.L.amos_lk_destroy.argtypes = [C.c_void_p]
        self.L.amos_lk_destroy.restype = None
        self.levels = self.L.amos_lk_levels(self.k)
        self.stream = self.L.amos_lk_stream(self.k)

    def close(self):
        if getattr(self, "k", None):
            self.L.amos_lk_destroy(self.k)
            self.k = None

    def __del__(self):
        self.close()

    def track_device(self, d_prev, prev_stride, d_next, next_stride, d_prev_xy, n, d_next_xy, d_status, d_err=None, max_count=20, epsilon=0.01,
                     min_eig=1e-4):
        _check(self.L.amos_lk_track_device(self.k, C.c_void_p(d_prev), C.c_size_t(prev_stride), C.c_void_p(d_next), C.c_size_t(next_stride),
                                           C.c_void_p(d_prev_xy), C.c_int(n), C.c_int(max_count), C.c_double(epsilon), C.c_float(min_eig),
                                           C.c_void_p(d_next_xy), C.c_void_p(d_status), C.c_void_p(d_err)), "amos_lk_track_device")


def image_bounds(width, height, fx, fy, cx, cy, dist_coef):
    """Frame::ComputeImageBounds (Frame.cc:1121-1170): (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    dc = np.ascontiguousarray(dist_coef, np.float32)
    out = np.zeros(4, np.float32)
    _check(lib().amos_frame_image_bounds(C.c_int(width), C.c_int(height), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(dc),
                                         C.c_int(len(dc)), _p(out)), "amos_frame_image_bounds")
    return tuple(float(v) for v in out)


class WindowSearch(C.Structure):
    """amos_window_search (include/amos_frontend.h)."""
    _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("d_counts", C.c_void_p), ("d_cell_start", C.c_void_p),
                ("d_items", C.c_void_p), ("d_query_uv", C.c_void_p), ("d_query_invz", C.c_void_p), ("d_u_right", C.c_void_p),
                ("d_pairs_q", C.c_void_p), ("d_pairs_t", C.c_void_p), ("scale_factors", C.c_void_p), ("n_pairs", C.c_int32),
                ("capacity", C.c_int32), ("n_levels", C.c_int32), ("mode", C.c_int32), ("init_dist", C.c_int32), ("th", C.c_float),
                ("mbf", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


class OrbMatcher:
    """The distance / best-two primitives every ORBmatcher::Search* inner loop reduces to
    (ORBmatcher.cc:1913-1933 and the candidate loops at :127-148, :278-304, :560-580, :1644-1690)."""

    def __init__(self, device=0, stream=None):
        self.L = lib()
        m = C.c_void_p()
        _check(self.L.amos_match_create(C.c_int(device), C.c_void_p(stream), C.byref(m)), "amos_match_create")
        self.m = m

    def close(self):
        if getattr(self, "m", None):
            self.L.amos_match_destroy(self.m)
            self.m = None

    def __del__(self):
        self.close()

    @staticmethod
    def _sets(q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        return q, t

    def distances(self, q, t):
        q, t = self._sets(q, t)
        out = np.zeros((len(q), len(t)), np.uint16)
        _check(self.L.amos_match_distances(self.m, _p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(out)),
               "amos_match_distances")
        return out

    def list_distances(self, q, t, cand_off, cand_idx):
        q, t = self._sets(q, t)
        cand_off, cand_idx = np.ascontiguousarray(cand_off, np.int32), np.ascontiguousarray(cand_idx, np.int32)
        out = np.zeros(len(cand_idx), np.uint16)
        _check(self.L.amos_match_list_distances(self.m, _p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(cand_off),
                                                _p(cand_idx), _p(out)), "amos_match_list_distances")
        return out

    def list_best2(self, q, t, cand_off, cand_idx, init_dist=256):
        q, t = self._sets(q, t)
        cand_off, cand_idx = np.ascontiguousarray(cand_off, np.int32), np.ascontiguousarray(cand_idx, np.int32)
        out = np.zeros(len(q), BEST2_DTYPE)
        _check(self.L.amos_match_list_best2(self.m, _p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(cand_off),
                                            _p(cand_idx), C.c_int(init_dist), _p(out)), "amos_match_list_best2")
        return out

    def bruteforce_best2(self, q, t, init_dist=256):
        q, t = self._sets(q, t)
        out = np.zeros(len(q), BEST2_DTYPE)
        _check(self.L.amos_match_bruteforce_best2(self.m, _p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)),
                                                  C.c_int(init_dist), _p(out)), "amos_match_bruteforce_best2")
        return out

    def set_bruteforce_kernel(self, mode):
        """0 = by size, 1 = xor + popcount kernel, 2 = i8 MFMA kernel (identical results)."""
        _check(self.L.amos_match_set_bruteforce_kernel(self.m, C.c_int({"auto": 0, "popcount": 1, "mfma": 2}.get(mode, mode))),
               "amos_match_set_bruteforce_kernel")

    def bruteforce_best2_batch_device(self, d_desc, frame_stride_bytes, d_counts, d_pairs_q, d_pairs_t, n_pairs,
                                      capacity, init_dist, d_out):
        _check(self.L.amos_match_bruteforce_best2_batch_device(
            self.m, C.c_void_p(d_desc), C.c_size_t(frame_stride_bytes), C.c_void_p(d_counts), C.c_void_p(d_pairs_q),
            C.c_void_p(d_pairs_t), C.c_int(n_pairs), C.c_int(capacity), C.c_int(init_dist), C.c_void_p(d_out)),
            "amos_match_bruteforce_best2_batch_device")

    def grid_build_batch_device(self, d_grid_cell, d_counts, n_frames, capacity, d_cell_start, d_items):
        """Frame::AssignFeaturesToGrid for a resident batch (Frame.cc:431-461)."""
        _check(self.L.amos_frame_grid_build_batch_device(self.m, C.c_void_p(d_grid_cell), C.c_void_p(d_counts), C.c_int(n_frames),
                                                         C.c_int(capacity), C.c_void_p(d_cell_start), C.c_void_p(d_items)),
               "amos_frame_grid_build_batch_device")

    def window_best2_batch_device(self, d_kps, d_desc, d_counts, d_cell_start, d_items, d_pairs_q, d_pairs_t, n_pairs, capacity,
                                  scale_factors, th, d_out, mode=0, init_dist=256, bounds=(0.0, 640.0, 0.0, 480.0), d_query_uv=None,
                                  d_query_invz=None, d_u_right=None, mbf=0.0):
        """GetFeaturesInArea + best/second loop of SearchByProjection(F, LastF) (Frame.cc:894-1003, ORBmatcher.cc:1629-1690)."""
        sf = np.ascontiguousarray(scale_factors, np.float32)
        w = WindowSearch(d_kps, d_desc, d_counts, d_cell_start, d_items, d_query_uv, d_query_invz, d_u_right, d_pairs_q, d_pairs_t,
                         sf.ctypes.data, n_pairs, capacity, len(sf), mode, init_dist, th, mbf, *bounds)
        _check(self.L.amos_match_window_best2_batch_device(self.m, C.byref(w), C.c_void_p(d_out)),
               "amos_match_window_best2_batch_device")

    def sync(self):
        _check(self.L.amos_match_sync(self.m), "amos_match_sync")

    @property
    def stream(self):
        return self.L.amos_match_stream(self.m)


class CornerDetector:
    """amos_corners_*: cv::goodFeaturesToTrack (Harris) + cv::cornerSubPix of Tracking::GetSceneFlowObj (Tracking.cc:894-895) on
    device-resident gray frames; the corners stay on the device (feed LkTracker.track_device)."""

    def __init__(self, max_width=640, max_height=480, device=0, stream=None):
        self.L = lib()
        self.L.amos_corners_stream.restype = C.c_void_p
        self.L.amos_corners_destroy.restype = None
        h = C.c_void_p()
        _check(self.L.amos_corners_create(C.c_int(device), C.c_void_p(stream), C.c_int(max_width), C.c_int(max_height), C.byref(h)), "amos_corners_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.amos_corners_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def stream(self):
        return self.L.amos_corners_stream(self.h)

    def good_features_device(self, gray_ptr, stride, width, height, xy_ptr, xy_capacity, count_ptr, max_corners=1000, quality=0.01, min_distance=8.0,
                             harris_k=0.04, response_ptr=None):
        _check(self.L.amos_corners_good_features_device(self.h, C.c_void_p(gray_ptr), C.c_size_t(stride), C.c_int(width), C.c_int(height), C.c_int(max_corners),
                                                        C.c_double(quality), C.c_double(min_distance), C.c_double(harris_k), C.c_void_p(xy_ptr),
                                                        C.c_int(xy_capacity), C.c_void_p(count_ptr), C.c_void_p(response_ptr)),
               "amos_corners_good_features_device")

    def candidate_count(self):
        n = C.c_int(0)
        _check(self.L.amos_corners_candidate_count(self.h, C.byref(n)), "amos_corners_candidate_count")
        return n.value

    def subpix_device(self, gray_ptr, stride, width, height, xy_ptr, count_ptr=None, n=0, win=10, max_count=20, epsilon=0.03):
        _check(self.L.amos_corners_subpix_device(self.h, C.c_void_p(gray_ptr), C.c_size_t(stride), C.c_int(width), C.c_int(height), C.c_void_p(xy_ptr),
                                                 C.c_void_p(count_ptr), C.c_int(n), C.c_int(win), C.c_int(max_count), C.c_double(epsilon)),
               "amos_corners_subpix_device")
