/*
 * amos_frontend.h -- C ABI of the MI355X-native Amos-SLAM front-end hot path.
 *
 * This is the drop-in boundary.  Every entry point below replaces one member of the
 * reference's C++ front-end API (file:line under /root/reference):
 *
 *   ORBextractor::ORBextractor        src/ORBextractor.cc:492-609   -> amos_orb_create / amos_orb_tables
 *   ORBextractor::operator() (3-arg)  src/ORBextractor.cc:1672-1686 -> amos_orb_detect (+ amos_orb_level_keypoints)
 *   ORBextractor::MovingKeyPoints     src/ORBextractor.cc:1688-1745 -> amos_orb_gate
 *   ORBextractor::ProcessDesp         src/ORBextractor.cc:1747-1820 -> amos_orb_describe
 *   ORBextractor::operator() (4-arg)  src/ORBextractor.cc:1544-1668 -> amos_orb_extract
 *   ORBextractor::mvImagePyramid      include/ORBextractor.h:168    -> amos_orb_level_image
 *   ORBmatcher::DescriptorDistance    src/ORBmatcher.cc:1913-1933   -> amos_match_distances / amos_match_list_distances
 *   inner best / second-best loops of ORBmatcher::Search* (src/ORBmatcher.cc:70-175, 230-382,
 *     515-643, 1569-1728)                                           -> amos_match_list_best2 / amos_match_bruteforce_best2
 *
 * The host-side C++ classes with the reference's names (ORB_SLAM2::ORBextractor, ORBmatcher)
 * live in amos-slam_amd/host/ and are thin wrappers over these functions; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only, no C++ or torch types.  Every function returns
 * AMOS_OK (0) or a negative error code; amos_last_error() returns a thread-local description.
 * Unless a name ends in _device, pointers are HOST pointers and the call is synchronous.
 * All work of one handle is issued on one HIP stream owned by that handle (or handed in at create).
 * A handle is stateful exactly like an ORBextractor instance (one frame/batch in flight per handle);
 * distinct handles may be used from distinct threads concurrently.
 */
#ifndef AMOS_FRONTEND_H
#define AMOS_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMOS_OK 0
#define AMOS_ERR_INVALID (-1)   /* bad argument (null, size, range)                              */
#define AMOS_ERR_DEVICE (-2)    /* HIP runtime error; text in amos_last_error()                  */
#define AMOS_ERR_CAPACITY (-3)  /* caller buffer too small, or frame larger than the handle's max */
#define AMOS_ERR_STATE (-4)     /* call order violated (e.g. describe before detect)             */

#define AMOS_EDGE_THRESHOLD 19  /* border of every pyramid plane, ORBextractor.cc:93             */
#define AMOS_MAX_LEVELS 16
#define AMOS_FRAME_GRID_ROWS 48 /* Frame.h:56 */
#define AMOS_FRAME_GRID_COLS 64 /* Frame.h:61 */
#define AMOS_TH_HIGH 100        /* ORBmatcher.cc:49 */
#define AMOS_TH_LOW 50          /* ORBmatcher.cc:50 */
#define AMOS_HISTO_LENGTH 30    /* ORBmatcher.cc:51 */

/* Same field order and widths as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave,
 * class_id), so a std::vector<cv::KeyPoint> can be filled with one memcpy. */
typedef struct amos_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} amos_keypoint;

/* The five ORBextractor.* YAML parameters (Tracking.cc:161-177). */
typedef struct amos_orb_params {
    int32_t n_features;
    float scale_factor;
    int32_t n_levels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} amos_orb_params;

typedef struct amos_orb amos_orb;
typedef struct amos_match amos_match;
typedef struct amos_mask_pre amos_mask_pre;

/* Result of a best / second-best reduction over one query's candidate list, ties resolved as the
 * reference's sequential `if(dist<best) ... else if(dist<best2)` loop does (first candidate wins). */
typedef struct amos_best2 {
    int32_t best_idx;     /* train index, -1 if no candidate beat init_dist */
    int32_t best_dist;    /* init_dist if none */
    int32_t second_idx;   /* -1 if none */
    int32_t second_dist;  /* init_dist if none */
} amos_best2;

const char *amos_last_error(void);
/* Number of HIP devices visible, or a negative error code. */
int amos_device_count(void);
/* The calling thread's current HIP device (hipGetDevice), or a negative error code.  The C++ drop-in classes
 * (amos-slam_amd/host) create their handles on it unless AMOS_DEVICE names another: a one-process-per-GPU host
 * that has called hipSetDevice / torch.cuda.set_device gets its objects on ITS GPU (the reference has no such
 * notion: its extractor is CPU code, ORBextractor.cc:492). */
int amos_current_device(void);
/* Which build of the library this is: "default" for the product build, otherwise the names of the timing-experiment
 * switches it was compiled with (AMOS_FAST_EXP=n, AMOS_W24_EXP_*, AMOS_FAST_LDS_PAD=n: builds whose RESULTS ARE WRONG,
 * tools/orb_variants.sh / tools/w24_variants.sh).  bench.py and the tests refuse anything but "default". */
const char *amos_build_variant(void);

/* ---------------------------------------------------------------- ORB extractor ------------- */

/* Allocates every device buffer for frames up to max_width x max_height and batches of up to
 * max_batch frames on HIP device `device`.  `stream` is a hipStream_t to issue on, or NULL to
 * let the handle create its own. */
int amos_orb_create(const amos_orb_params *params, int max_width, int max_height, int max_batch,
                    int device, void *stream, amos_orb **out);
void amos_orb_destroy(amos_orb *h);

/* Host-only capacity check (no device is touched): would a handle created for frames up to
 * max_width x max_height accept a width x height frame?  need / cap (either may be NULL) receive
 * {FAST cells, candidate slots, compacted candidates, per-level keypoint slots, resize tap records,
 * pyramid bytes / 256} of the frame and of the allocation.  Returns AMOS_OK, AMOS_ERR_CAPACITY
 * (a bound is exceeded: amos_orb_create's sizing would be wrong) or AMOS_ERR_INVALID (the frame has a
 * level without a FAST cell, which the reference cannot process either, ORBextractor.cc:1083-1086). */
int amos_orb_geometry_probe(const amos_orb_params *params, int max_width, int max_height, int width,
                            int height, int32_t need[6], int32_t cap[6]);

/* Constructor tables (a1).  Any pointer may be NULL.  Arrays hold n_levels entries, umax 16. */
int amos_orb_tables(const amos_orb *h, float *scale_factor, float *inv_scale_factor,
                    float *level_sigma2, float *inv_level_sigma2, int32_t *features_per_level,
                    int32_t *umax);
/* The same tables from the parameters alone: host arithmetic only, no device is touched and no handle is needed
 * (the reference's constructor is host code too, ORBextractor.cc:492-609).  Arrays hold params->n_levels entries. */
int amos_orb_tables_host(const amos_orb_params *params, float *scale_factor, float *inv_scale_factor,
                         float *level_sigma2, float *inv_level_sigma2, int32_t *features_per_level, int32_t *umax);
/* Level geometry for a w x h input: width/height per level.  Returns n_levels. */
int amos_orb_level_sizes(const amos_orb *h, int width, int height, int32_t *level_w, int32_t *level_h);

/* a7: pyramid + per-cell FAST + quad-tree distribution + orientation for ONE host frame
 * (8-bit gray, `stride` bytes per row).  Afterwards per-level keypoints (level coordinates,
 * not rescaled) are available through amos_orb_level_keypoints(frame = 0). */
int amos_orb_detect(amos_orb *h, const uint8_t *gray, size_t stride, int width, int height);

/* Number of keypoints of (frame, level) after the last detect/gate, or a negative error. */
int amos_orb_level_count(amos_orb *h, int frame, int level);
int amos_orb_level_keypoints(amos_orb *h, int frame, int level, amos_keypoint *out, int cap);
/* Replaces the device-resident list of (frame, level) by the caller's (the reference hands the
 * per-level vectors back into MovingKeyPoints / ProcessDesp, Frame.cc:491-496,633). */
int amos_orb_set_level_keypoints(amos_orb *h, int frame, int level, const amos_keypoint *kps, int n);

/* Bulk forms of the two calls above for the C++ drop-in classes: one transfer for all levels.
 * Layout: the list of level l occupies buf[offsets[l] .. offsets[l] + counts[l]), capacity
 * caps[l]; total = buffer length in keypoints.  offsets/caps depend on the frame size only. */
int amos_orb_level_layout(amos_orb *h, int32_t *offsets, int32_t *caps, int32_t *total);
int amos_orb_fetch_levels(amos_orb *h, int frame, int32_t *counts, amos_keypoint *buf, int buf_len);
int amos_orb_store_levels(amos_orb *h, int frame, const int32_t *counts, const amos_keypoint *buf, int buf_len);

/* a8: closing (dilate then erode, 31x31 ellipse) of the 8-bit mask of the level-0 frame size and
 * removal of every keypoint whose scaled position hits a non-zero closed-mask pixel or a removed
 * cluster.  `labels` (row-major doubles, lstride elements per row), `center_ids`, `rm_vector`
 * describe the SLIC/k-means label gate and may all be NULL (label gate off).  Removed keypoints
 * are returned in the reference's order (level by level, list order).  Operates on frame 0. */
int amos_orb_gate(amos_orb *h, const uint8_t *mask, size_t mask_stride, const double *labels,
                  size_t lstride, const int32_t *center_ids, int n_centers,
                  const int32_t *rm_vector, int n_rm, amos_keypoint *removed, int cap,
                  int *n_removed);
/* The closed mask of the last amos_orb_gate call (for parity tests). */
int amos_orb_closed_mask(amos_orb *h, uint8_t *dst, size_t dst_stride);

/* a9: 7x7 sigma-2 blur of every level, 256-bit rBRIEF of the current per-level lists, rescale of
 * the coordinates to level 0, concatenation in level order.  desc is n x 32 bytes row-major. */
int amos_orb_describe(amos_orb *h, amos_keypoint *kps, uint8_t *desc, int cap, int *n);

/* a11 = detect + describe without gating. */
int amos_orb_extract(amos_orb *h, const uint8_t *gray, size_t stride, int width, int height,
                     amos_keypoint *kps, uint8_t *desc, int cap, int *n);

/* mvImagePyramid[level] of `frame`: the level image (padded != 0: with its 19-px reflect-101
 * border, i.e. (w+38) x (h+38)) copied to host memory. */
int amos_orb_level_image(amos_orb *h, int frame, int level, uint8_t *dst, size_t dst_stride,
                         int padded);
/* All level planes of one frame at once (mvImagePyramid of the C++ class): dst[l] / dst_strides[l] per level (dst[l] NULL skips the level);
 * padded != 0 copies the (w + 38) x (h + 38) plane including the reflect-101 border, as amos_orb_level_image does.  One device-to-host
 * transfer of the frame's pyramid through a pinned staging buffer. */
int amos_orb_pyramid_images(amos_orb *h, int frame, uint8_t *const *dst, const size_t *dst_strides, int padded);
/* The blurred level used by the last describe (unpadded, w x h). */
int amos_orb_blurred_image(amos_orb *h, int frame, int level, uint8_t *dst, size_t dst_stride);
/* FAST candidates of (frame, level) before the quad-tree, in the reference's order
 * (cell-row-major, then row-major inside the cell): x, y relative to (minBorderX, minBorderY),
 * response = score.  For parity tests. */
int amos_orb_level_candidates(amos_orb *h, int frame, int level, amos_keypoint *out, int cap);

/* Batched, device-resident path.  d_gray holds n_frames frames, frame f at
 * d_gray + f * frame_stride, rows `row_stride` bytes apart.  Runs a11 on every frame; results stay
 * on the device (amos_orb_batch_results_device) and nothing is copied to the host.  Asynchronous on
 * the handle's stream; amos_orb_sync() waits. */
int amos_orb_extract_batch_device(amos_orb *h, const uint8_t *d_gray, size_t frame_stride,
                                  size_t row_stride, int width, int height, int n_frames);
/* The same split in stages for the full front-end with the mask (a7 -> a8 -> a9 per frame of the
 * batch, everything device-resident): detect, then gate with n_frames 8-bit masks (frame f at
 * d_masks + f * mask_frame_stride; label gate off), then describe.  Asynchronous. */
int amos_orb_detect_batch_device(amos_orb *h, const uint8_t *d_gray, size_t frame_stride, size_t row_stride,
                                 int width, int height, int n_frames);
int amos_orb_gate_batch_device(amos_orb *h, const uint8_t *d_masks, size_t mask_frame_stride,
                               size_t mask_row_stride);
int amos_orb_describe_batch_device(amos_orb *h);

/* ---- callers either side of the path (SURVEY 8f "next" rows), device resident ----
 * Tracking::GrabImageRGBD's cvtColor(..., CV_{BGR,RGB}[A]2GRAY) (Tracking.cc:308-321) fused into the
 * level-0 import: d_color holds interleaved 8-bit frames with `channels` = 3 or 4 bytes per pixel,
 * rgb_order != 0 for RGB[A] (Tracking's mbRGB), 0 for BGR[A].  Otherwise as amos_orb_extract_batch_device. */
int amos_orb_extract_batch_device_color(amos_orb *h, const uint8_t *d_color, size_t frame_stride, size_t row_stride,
                                        int width, int height, int n_frames, int channels, int rgb_order);
/* Frame::ComputeStereoFromRGBD (Frame.cc:1576-1615) and the cell of Frame::AssignFeaturesToGrid /
 * PosInGrid (Frame.cc:431-461, 1007-1030) for every keypoint of the last batch.  The depth map is 16-bit with Tracking's convertTo(CV_32F,
 * depth_map_factor) applied on the fly (depth_is_u16 != 0) or already float32; d_depth NULL = monocular
 * (grid cells only).  d_kps_un = mvKeysUn from amos_frame_undistort_batch_device, or NULL for a
 * zero-distortion camera (mvKeysUn == mvKeys, as for TUM3): the depth is read at mvKeys, uRight and the
 * cell use mvKeysUn.  Outputs are [n_frames][capacity] arrays on the device (capacity from amos_orb_batch_results_device):
 * u_right / depth = -1 where the depth is not positive, grid_cell = col * 48 + row or -1. */
int amos_frame_rgbd_glue_batch_device(amos_orb *h, const void *d_depth, int depth_is_u16, float depth_map_factor,
                                      size_t depth_frame_stride_bytes, size_t depth_row_stride_bytes, float mbf,
                                      float min_x, float max_x, float min_y, float max_y,
                                      const amos_keypoint *d_kps_un, float *d_u_right, float *d_depth_out,
                                      int32_t *d_grid_cell);
/* Frame::UndistortKeyPoints (Frame.cc:1052-1118) for every keypoint of the last batch:
 * cv::undistortPoints(pts, pts, K, distCoef, Mat(), K) with OpenCV's default criteria (5 iterations, double
 * arithmetic), distCoef = (k1, k2, p1, p2[, k3]) as Tracking reads them from the YAML (n_dist = 4 or 5;
 * n_dist = 0 or k1 == 0 copies the keypoints, Frame.cc:1057-1061).  d_kps_un is [n_frames][capacity]. */
int amos_frame_undistort_batch_device(amos_orb *h, float fx, float fy, float cx, float cy, const float *dist_coef,
                                      int n_dist, amos_keypoint *d_kps_un);
/* Frame::ComputeImageBounds (Frame.cc:1121-1170): bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} from the four
 * undistorted image corners (host; the same point routine as the kernel). */
int amos_frame_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist_coef,
                            int n_dist, float bounds[4]);

/* Device pointers of the batch results: keypoints [max_batch][capacity], descriptors
 * [max_batch][capacity][32], counts [max_batch].  Valid until destroy. */
int amos_orb_batch_results_device(amos_orb *h, const amos_keypoint **d_kps, const uint8_t **d_desc,
                                  const int32_t **d_counts, int *capacity);
/* Copies frame `frame` of the last batch to host buffers. */
int amos_orb_batch_fetch(amos_orb *h, int frame, amos_keypoint *kps, uint8_t *desc, int cap, int *n);
int amos_orb_sync(amos_orb *h);

/* Per-kernel timing with HIP events on the handle's stream (measurement, bench.py's roofline).
 * After amos_orb_timing_enable(h, max_records) every batch / extract pass records one event
 * between consecutive stages; amos_orb_timing_collect() synchronises, returns the average
 * duration of each stage in milliseconds over the recorded passes and resets the record count.
 * Stages: 0 level-0 import (k_pyramid_level0), 1 resize levels (n_levels - 1 launches of
 * k_pyramid_level), 2 FAST cells, 3 quad-tree, 4 orientation, 5 blur (side stream, overlaps
 * FAST), 6 rBRIEF.  max_records = 0 switches timing off. */
#define AMOS_ORB_STAGES 7
int amos_orb_timing_enable(amos_orb *h, int max_records);
int amos_orb_timing_collect(amos_orb *h, float *avg_ms, int *n_records);
/* The hipStream_t the handle issues on. */
void *amos_orb_stream(amos_orb *h);

/* ---------------------------------------------------------------- matcher ------------------- */

int amos_match_create(int device, void *stream, amos_match **out);
void amos_match_destroy(amos_match *m);
int amos_match_sync(amos_match *m);
void *amos_match_stream(amos_match *m);

/* a12, dense: out[i*nt + j] = Hamming(q_i, t_j), descriptors 32 bytes each, row-major. */
int amos_match_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                         uint16_t *out);
/* a12 over candidate lists (CSR): query i owns cand_idx[cand_off[i] .. cand_off[i+1]);
 * out[k] = Hamming(q_i, t_{cand_idx[k]}).  This is what the greedy Search* loops consume. */
int amos_match_list_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                              const int32_t *cand_off, const int32_t *cand_idx, uint16_t *out);
/* a13 inner reduction over candidate lists: best and second best in candidate order.
 * init_dist is the loop's initial bestDist (256 in SearchByProjection, INT_MAX in
 * SearchForInitialization): only distances < init_dist can win. */
int amos_match_list_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                          const int32_t *cand_off, const int32_t *cand_idx, int init_dist,
                          amos_best2 *out);
/* The same with every train descriptor as candidate, in index order (N_q x N_t brute force). */
int amos_match_bruteforce_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                                int init_dist, amos_best2 *out);
/* Which kernel the two brute-force entry points run: 0 = chosen by size (default), 1 = xor + popcount on the vector
 * units, 2 = exact-integer i8 MFMA (ham = |q| + sum_k t_k (1 - 2 q_k), v_mfma_i32_32x32x32_i8).  Identical results;
 * a measurement / test switch. */
int amos_match_set_bruteforce_kernel(amos_match *m, int mode);
/* Device-pointer form for the batched pipeline: for each of n_pairs (query frame, train frame)
 * pairs: descriptors at d_desc + frame * frame_stride_bytes, counts in d_counts[frame]; out is
 * [n_pairs][capacity] amos_best2 on the device.  Asynchronous on the matcher's stream. */
int amos_match_bruteforce_best2_batch_device(amos_match *m, const uint8_t *d_desc,
                                             size_t frame_stride_bytes, const int32_t *d_counts,
                                             const int32_t *d_pairs_q, const int32_t *d_pairs_t,
                                             int n_pairs, int capacity, int init_dist,
                                             amos_best2 *d_out);

/* ---------------------------------------------------------------- resident search (8f-1) ---- */

/* Frame::AssignFeaturesToGrid (Frame.cc:431-461) for a batch on the device.  d_grid_cell is the
 * per-keypoint cell number written by amos_frame_rgbd_glue_batch_device ([n_frames][capacity], -1 =
 * outside the grid); d_counts the keypoint counts.  Output, per frame: CSR d_cell_start
 * [64*48 + 1] (cell = x * 48 + y, as mGrid[x][y]) and d_items [capacity] holding keypoint indices,
 * ascending inside a cell (the reference's push_back order).  Asynchronous on the matcher's stream. */
int amos_frame_grid_build_batch_device(amos_match *m, const int32_t *d_grid_cell, const int32_t *d_counts,
                                       int n_frames, int capacity, int32_t *d_cell_start,
                                       int32_t *d_items);

/* Frame::GetFeaturesInArea (Frame.cc:894-1003) + the best / second-best candidate loop of
 * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1629-1690) for a
 * batch of (query frame, train frame) pairs of resident extraction results.  Query i of pair p is
 * keypoint i of frame d_pairs_q[p]; it is searched in frame d_pairs_t[p] around d_query_uv[p][i]
 * (its own position when d_query_uv is NULL) with radius th * scale_factors[octave] and the level
 * window of mode (0: octave-1..octave+1, 1 = bForward: >= octave, 2 = bBackward: <= octave).  With
 * d_u_right and d_query_invz the stereo gate |u - mbf*invz - uRight[i2]| > radius rejects (:1662-1669).
 * d_out[p][i] = best / second best in candidate order under "dist < init_dist" (256 in the reference).
 * The greedy skip of already-matched features (:1658-1660) is the caller's: the unrestricted best
 * equals the restricted best whenever it is still free. */
typedef struct amos_window_search {
    const amos_keypoint *d_kps;   /* [frames][capacity]      (amos_orb_batch_results_device) */
    const uint8_t *d_desc;        /* [frames][capacity][32] */
    const int32_t *d_counts;      /* [frames] */
    const int32_t *d_cell_start;  /* [frames][64*48+1]       (amos_frame_grid_build_batch_device) */
    const int32_t *d_items;       /* [frames][capacity] */
    const float *d_query_uv;      /* [n_pairs][capacity][2] or NULL */
    const float *d_query_invz;    /* [n_pairs][capacity] or NULL */
    const float *d_u_right;       /* [frames][capacity] or NULL */
    const int32_t *d_pairs_q, *d_pairs_t; /* [n_pairs] */
    const float *scale_factors;   /* host, n_levels entries (mvScaleFactors) */
    int32_t n_pairs, capacity, n_levels;
    int32_t mode, init_dist;
    float th, mbf;
    float min_x, max_x, min_y, max_y; /* Frame::mnMinX .. mnMaxY */
} amos_window_search;
int amos_match_window_best2_batch_device(amos_match *m, const amos_window_search *w, amos_best2 *d_out);

/* ---------------------------------------------------------------- mask pre-processing (8f-4) - */

/* The pre-processing chain of the mask pass on the device, from the raw BGR frame to the network input:
 * yolact::evalImage's marshalling (yolact.cc:220, 385-451: cv::resize to W480 x H640, u8 / 255.0),
 * eval_image's "* 255" and cv2.resize back to 640 x 480 (yolact_interface.py:862-866) and FastBaseTransform
 * (utils/augmentations.py:616-657: bilinear to 550 x 550, (x - MEANS) / STD, BGR -> RGB).
 * d_bgr: [n_frames][height][width][3] uint8; d_out: [n_frames][3][550][550] float32.  Asynchronous on the
 * handle's stream (pass PyTorch's current stream so that the network simply follows). */
int amos_mask_pre_create(int device, void *stream, int width, int height, int max_batch, amos_mask_pre **out);
void amos_mask_pre_destroy(amos_mask_pre *p);
void *amos_mask_pre_stream(amos_mask_pre *p); /* the hipStream_t the handle issues on */
int amos_mask_preprocess_batch_device(amos_mask_pre *p, const uint8_t *d_bgr, int n_frames, float *d_out);

/* SURVEY 8f-4, "one pass over the RGB-D frame": the colour frame is read ONCE for both of its consumers -- the gray
 * conversion into the padded level 0 (Tracking.cc:308-321, then the rest of amos_orb_detect_batch_device) and stage A of
 * the mask pre-processing (yolact.cc:220, 385-451) -- and the network input appears in d_net_input ([n][3][550][550]
 * float32) on the extractor's stream.  Same results as amos_orb_extract_batch_device_color's detect part and
 * amos_mask_preprocess_batch_device; follow with amos_orb_gate_batch_device / amos_orb_describe_batch_device. */
int amos_orb_detect_color_with_mask_pre_batch_device(amos_orb *h, amos_mask_pre *pre, const uint8_t *d_color,
                                                     size_t frame_stride, size_t row_stride, int width, int height,
                                                     int n_frames, int channels, int rgb_order, float *d_net_input);

/* Fused epilogue of one convolution of the mask network, in place on a channels-last (NHWC) float32 tensor of n
 * elements: y = act((y + bias[c]) + residual), act = ReLU when relu != 0, residual may be NULL.  Replaces PyTorch's
 * separate bias-add, residual-add and clamp passes (same summation order: identical bits).  Asynchronous on `stream`
 * (a hipStream_t; pass PyTorch's current stream). */
int amos_mask_bias_act_device(void *stream, float *d_y, const float *d_bias, const float *d_residual, size_t n,
                              int channels, int relu);

/* The stem's tail (backbone.py ResNetBackbone.forward: bn1 folded -> relu -> maxpool): y = max_pool2d(relu(x + bias[c]), 3, stride 2,
 * padding 1) of a channels-last float32 tensor x [n][in_h][in_w][channels] (a convolution's raw output) in ONE pass; y is
 * [n][(in_h - 1) / 2 + 1][(in_w - 1) / 2 + 1][channels].  Bit-identical to the separate passes (max and the monotone bias + ReLU commute). */
int amos_mask_bias_relu_maxpool_device(void *stream, const float *d_x, const float *d_bias, float *d_y, int n, int in_h, int in_w,
                                       int channels);

/* The whole stem of the backbone (backbone.py:77-80,129-132: conv1 = Conv2d(3, 64, 7, stride 2, padding 3) with bn1 folded into weight and
 * bias, ReLU, MaxPool2d(3, stride 2, padding 1)) as ONE kernel on the fp32 MFMA units: the 7 x 7 convolution's output stays in registers /
 * LDS and only the pooled tensor is written.  x is float32 [batch][3][height][width] addressed through ELEMENT strides (frame, channel, row,
 * column: planar and channels-last inputs alike); y is channels-last [batch][ph][pw][64] with ch = (height - 1) / 2 + 1, ph = (ch - 1) / 2 + 1
 * (likewise for the width).  d_packed: the weights as amos_mask_stem_weights_device lays them out (amos_mask_stem_weight_floats() floats)
 * from a [64][3][7][7] tensor addressed through its element strides (out channel, in channel, row, column).  Sums in float32 in the
 * MFMA's order (the library's convolution sums in another order: results agree to float32 rounding of a 147-term sum, not bit for bit);
 * bias + ReLU after the maximum, which is bit-identical to before it.  Replaces F.conv2d + amos_mask_bias_relu_maxpool_device. */
int amos_mask_stem_weight_floats(void);
int amos_mask_stem_weights_device(void *stream, const float *d_w, long long stride_n, long long stride_c, long long stride_y, long long stride_x,
                                  float *d_packed);
int amos_mask_stem_device(void *stream, const float *d_x, long long stride_b, long long stride_c, long long stride_y, long long stride_x,
                          const float *d_packed, const float *d_bias, float *d_y, int batch, int height, int width);

/* A convolution of the mask network (yolact.py / backbone.py as amos-slam_amd/mask/net.py restates them: the ResNet-50
 * bottlenecks, the FPN, the prototype network, the prediction heads) on channels-last float32 tensors as one fp32 MFMA
 * (implicit) GEMM with the epilogue of amos_mask_bias_act_device fused:
 *   y[b][oy][ox][n] = act( sum_{dy,dx,k} x[b][oy*stride - pad + dy][ox*stride - pad + dx][k] * w[n][dy][dx][k]
 *                          + bias[n]  (+ residual[b][oy][ox][n]) )        (zeros outside the image, dilation 1, groups 1)
 * x: [batch][in_h][in_w][cin]; w: [cout][kh][kw][cin], i.e. a Conv2d weight in channels-last memory format (for 1 x 1 either
 * format); y and residual: [batch][oh][ow][cout] with oh = (in_h + 2 pad - kh) / stride + 1; bias and residual may be NULL.
 * Requires cin % 32 == 0, cout % 64 == 0, kernel <= 7 x 7, pad < kernel, stride 1..4, 16-byte aligned pointers
 * (amos_mask_conv_supported returns AMOS_OK for such a shape, AMOS_ERR_INVALID otherwise: the caller then keeps its library
 * convolution).  The sum is taken in another order than MIOpen's: results agree to float32 rounding, not bit for bit.
 * Asynchronous on `stream`.  amos_mask_conv1x1_* are the kh = kw = 1, pad = 0 case. */
int amos_mask_conv_supported(int cin, int cout, int kh, int kw, int stride, int pad);
int amos_mask_conv_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual,
                          float *d_y, int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad,
                          int relu);
/* The same convolution with split-K for SMALL launches (one frame per pass: a 1 x 1 layer at 35 x 35 or 18 x 18 is 3 - 40 work-groups of up
 * to 64 k-stages on 256 CUs): the k loop is cut into splits, one work-group per (tile, split); every split leaves its tile in the workspace
 * and the last work-group to arrive at a tile adds the splits in split order (the same bits whichever group that is) and applies bias,
 * residual and ReLU -- one launch, no zero fill, no reduction pass.  amos_mask_conv_workspace_bytes: scratch this shape wants (0: the plan
 * for it is an ordinary launch; d_workspace may then be NULL).  The first 16 KB of the workspace are tile counters: ZERO before the first
 * use, left zero by every launch; launches that share a workspace must be ordered on one stream. */
size_t amos_mask_conv_workspace_bytes(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad);
int amos_mask_conv_ws_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual, float *d_y,
                             int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                             void *d_workspace, size_t workspace_bytes);
/* Work-group tile of amos_mask_conv_device: -1 = automatic (128 x 128 outputs unless that leaves fewer than 1 024 work-groups or
 * cout % 128 != 0, then 128 x 64), 0 = 128 x 128 wherever cout allows, 1 = always 128 x 64.  Returns the mode in force before
 * the call; any other argument only queries.  The environment's AMOS_GEMM_NARROW (0 / 1) is the initial mode, read once.
 * amos_mask_conv_kernel_name writes the name of the kernel the call with these arguments launches (as a profiler shows it). */
int amos_mask_conv_tile_mode(int mode);
int amos_mask_conv_kernel_name(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, char *name,
                               int name_len);
int amos_mask_conv1x1_supported(int cin, int cout, int stride);

/* The same convolution for 3 x 3 kernels with stride 1 and pad 1 (the prototype network, the prediction head, the FPN's prediction layers
 * and the bottlenecks' conv2 of yolact.py / backbone.py: 70 % of the network's multiply-accumulates) as Winograd F(2 x 2, 3 x 3) on the
 * fp32 MFMA units: 2.25 x fewer multiplies, float32 in and float32 accumulate; the result differs from the direct form by float32
 * rounding of the transforms (tests/test_mask.py holds both to the same bound against a float64 convolution).
 *   amos_mask_winograd_supported(cin, cout): AMOS_OK when cin % 16 == 0, cin >= 32 and cout % 64 == 0.
 *   amos_mask_winograd_weight_floats(cin, cout): size of the transformed weight in floats (16 * cin * cout; 0 if unsupported).
 *   amos_mask_winograd_weights_device: d_w [cout][3][3][cin] (a channels-last Conv2d weight) -> d_u = G g G^T of every filter, laid
 *     out in the order the kernel's MFMA fragments are read; once per layer.
 *   amos_mask_winograd_conv_device: y = act(conv3x3(x, w) + bias (+ residual)); x [batch][h][w][cin] (below 2 GiB), y and residual
 *     [batch][h][w][cout], all 16-byte aligned; bias and residual may be NULL.  Asynchronous on `stream`. */
int amos_mask_winograd_supported(int cin, int cout);
size_t amos_mask_winograd_weight_floats(int cin, int cout);
int amos_mask_winograd_weights_device(void *stream, const float *d_w, float *d_u, int cin, int cout);
int amos_mask_winograd_conv_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual,
                                   float *d_y, int batch, int h, int w, int cin, int cout, int relu);
/* The same layers as Winograd F(2 x 4, 3 x 3): F(2, 3) vertically, F(4, 3) horizontally -- 24 multiplies per 2 x 4 outputs and channel
 * pair (3.0 per output against 4.0 for F(2 x 2) and 9.0 direct); same arguments, same support rule (amos_mask_winograd_supported),
 * same epilogue; the transformed weight G2 g G4^T has 24 * cin * cout floats and its own layout (make it with the matching function).
 * Rounding: the F(4, 3) transforms carry the factors 4, 5, 8 and 1 / 24, so its error against a float64 convolution is a few times
 * that of F(2 x 2) -- tests/test_mask.py holds both to the direct kernels' bound (1e-5 of the sum of |terms|). */
size_t amos_mask_winograd24_weight_floats(int cin, int cout);
int amos_mask_winograd24_weights_device(void *stream, const float *d_w, float *d_u, int cin, int cout);
int amos_mask_winograd24_conv_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual,
                                     float *d_y, int batch, int h, int w, int cin, int cout, int relu);
/* The same convolution with the channel-blocked activation layout on either side: in_blocked != 0 reads d_x as [batch][cin / 8][h][w][8]
 * floats, out_blocked != 0 writes d_y (and reads d_residual) as [batch][cout / 8][h][w][8]; 0 = channels-last [batch][h][w][c] (what
 * amos_mask_winograd24_conv_device takes).  A stage of the kernel reads 8 input channels of every pixel of its patch: blocked, those are
 * contiguous whole cache lines used once; channels-last, a quarter of every line four times.  Same arithmetic, same bits. */
int amos_mask_winograd24_conv_layout_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual,
                                            float *d_y, int batch, int h, int w, int cin, int cout, int relu, int in_blocked, int out_blocked);
/* Which launch form the F(2 x 4) kernel takes: -1 automatic and 0: one work-group per (tile block, channel tile) id (the default form);
 * 1: the PERSISTENT form -- one work-group per CU walking the ids, the next id's first requests travelling under the current id's last
 * stages and epilogue.  Measured 3.5 - 5 % slower on the network's large layers (tools/r5_w24_persist.py), hence not the automatic choice;
 * kept selectable and under test.  Returns the previous mode (-2 or anything else out of range: query only).  Same bits either way;
 * AMOS_W24_PERSIST=0 / 1 in the environment is the initial value. */
int amos_mask_winograd24_persistent_mode(int mode);
/* Output channels per work-group of the F(2 x 4) kernel: -1 by launch size (the default: 32 per group -- twice the groups -- while 64-channel
 * groups would leave most CUs without one, i.e. one-frame launches), 0 always 64, 1 always 32.  Returns the previous mode (out of range:
 * query only).  Same bits either way; AMOS_W24_NARROW=0 / 1 in the environment is the initial value. */
int amos_mask_winograd24_narrow_mode(int mode);
int amos_mask_conv1x1_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual,
                             float *d_y, int batch, int in_h, int in_w, int cin, int cout, int stride, int relu);

/* F.interpolate(x, mode="bilinear", align_corners=False) of a channels-last float32 tensor [n][in_h][in_w][channels] ->
 * [n][out_h][out_w][channels] (the FPN's top-down path and the prototype network's x2 step).  scale_h / scale_w as
 * PyTorch derives them: 1 / scale_factor when a scale factor was given, else in / out (float32).  channels % 4 == 0. */
int amos_mask_bilinear_nhwc_device(void *stream, const float *d_x, float *d_y, int n, int in_h, int in_w, int out_h,
                                   int out_w, int channels, float scale_h, float scale_w);

/* The same with a ReLU on the result when relu != 0 (the prototype network's upsample, yolact.py proto_net[6..7]). */
int amos_mask_bilinear_nhwc_act_device(void *stream, const float *d_x, float *d_y, int n, int in_h, int in_w, int out_h,
                                       int out_w, int channels, float scale_h, float scale_w, int relu);

/* An exact x 2 enlargement (out = 2 x in, scale 0.5: the prototype network's upsampling step) runs a form of the kernel that makes 2 x 2 outputs
 * per thread from the 3 x 3 source pixels they touch (the same bits, fewer reads).  mode 1: where it applies (default), 0: never (A/B runs,
 * tests); any other value only reports.  Returns the mode in force before the call. */
int amos_mask_bilinear_x2_mode(int mode);

/* The suppression term of Fast NMS (layers/functions/detection.py:103-170 fast_nms, layers/box_utils.py jaccard): d_boxes holds
 * n_lists lists of k boxes [x1, y1, x2, y2] sorted by descending score; d_out[list][j] = max over i < j of IoU(box i, box j), 0 for
 * j == 0 -- what `jaccard(boxes, boxes).triu_(diagonal=1).max(dim=1)` yields, in jaccard's float32 arithmetic (bit-identical to
 * PyTorch's elementwise kernels), without materialising the k x k matrices.  1 <= k <= 256. */
int amos_mask_nms_column_max_device(void *stream, const float *d_boxes, float *d_out, int n_lists, int k);

/* Class scores for Fast NMS in the static-shape batch form (layers/functions/detection.py:46-75 as mask/detect.py detect_batch
 * restates it): d_conf [batch][n_priors][n_classes_with_background] (the network's softmax output) -> d_scores
 * [batch][n_classes_with_background - 1][n_priors]: the background column dropped, the class axis first (the layout top-k
 * reads), and -1 for every prior whose best non-background score is not > threshold.  Pure selection: exact. */
int amos_mask_class_scores_device(void *stream, const float *d_conf, float *d_scores, int batch, int n_priors,
                                  int n_classes_with_background, float threshold);

/* The tail of yolact_interface.py postprocess + prep_display (:678-779, :806-832): d_masks [batch][n_det][mask_h][mask_w] are the
 * cropped sigmoid masks, d_flags [batch][n_det] != 0 marks the detections that count (displayed persons); d_out [batch][out_h][out_w]
 * = (number of flagged detections whose mask, upsampled bilinearly (align_corners = False) to the frame, is > 0.5) * 255 modulo 256,
 * the reference's `(m * 255).byte()`.  Same source index and weights as PyTorch's upsample_bilinear2d. */
int amos_mask_person_mask_device(void *stream, const float *d_masks, const uint8_t *d_flags, uint8_t *d_out, int batch, int n_det,
                                 int mask_h, int mask_w, int out_h, int out_w);

/* The prediction head's outputs for one pyramid level (yolact.py PredictionModule.forward and the cat / softmax of Yolact.forward):
 * d_raw [batch][cells][channels_padded] = the merged output convolution WITHOUT bias, channels [anchors x 4 | anchors x
 * n_classes_with_background | anchors x mask_dim | padding]; the level's priors (cell-major, anchor-minor) start at prior_offset of
 * d_loc [batch][n_priors_total][4] = raw + bias, d_conf [..][n_classes_with_background] = softmax(raw + bias), d_coef [..][mask_dim]
 * = tanh(raw + bias).  loc and coef equal PyTorch's bit for bit, conf to float32 rounding (order of the softmax sum). */
int amos_mask_head_outputs_device(void *stream, const float *d_raw, const float *d_bias, float *d_loc, float *d_conf, float *d_coef,
                                  int batch, int cells, int channels_padded, int anchors, int n_classes_with_background, int mask_dim,
                                  int n_priors_total, int prior_offset);
/* The same with Detect's class scores written by the same kernel: d_scores [batch][classes][n_priors_total] (background column dropped, -1 for
 * every prior whose best class is not above `threshold`) exactly as amos_mask_class_scores_device makes them from d_conf -- one pass over the
 * softmax tensor saved, and d_conf itself may be NULL when only the detector reads the head (d_scores may be NULL instead: the plain form). */
int amos_mask_head_outputs_scores_device(void *stream, const float *d_raw, const float *d_bias, float *d_loc, float *d_conf, float *d_coef, float *d_scores,
                                         float threshold, int batch, int cells, int channels_padded, int anchors, int n_classes_with_background,
                                         int mask_dim, int n_priors_total, int prior_offset);

/* Row-wise top-k, sorted descending (the per-class `scores.topk(200)` of Fast NMS, layers/functions/detection.py:103-111): d_x
 * [rows][n] -> d_values [rows][k], d_indices [rows][k] (int64, as torch.topk returns them).  Values equal torch.topk's; among equal
 * values the lower index comes first (torch leaves that order unspecified).  1 <= k <= min(256, n). */
int amos_mask_topk_rows_device(void *stream, const float *d_x, float *d_values, long long *d_indices, int rows, int n, int k);
/* The same result for rows that are mostly ONE value `fill` with everything of interest above it (the class-score rows: -1 for the priors
 * under the confidence threshold): one scan of the row instead of five -- the values above `fill` are compacted and sorted in LDS, a
 * result short of k is completed with `fill` at its first indices.  Rows that do not fit the assumption (more than 1 024 values above
 * `fill`, or values below it when the list is short of k) take the generic path inside the same launch: always amos_mask_topk_rows_device's
 * result. */
int amos_mask_topk_rows_sparse_device(void *stream, const float *d_x, float *d_values, long long *d_indices, int rows, int n, int k, float fill);

/* ---- the post-processing of the mask pass as ONE call: Detect (box decoding, per-class top 200, Fast NMS at IoU 0.5, class-confidence
 * threshold 0.05; layers/functions/detection.py:27-170, layers/box_utils.py:268-312) + postprocess / prep_display (score threshold 0.15, the
 * 15 best detections, masks = sigmoid(proto . coefficients) cropped to the box with 1 px padding, bilinear to the frame, > 0.5, the persons
 * summed, x 255 modulo 256; yolact_interface.py:678-779, 806-832) in the static-shape form of mask/detect.py detect_batch and mask/post.py
 * person_mask_batch.  Inputs (float32, device): loc [batch][P][4], conf [batch][P][classes incl. background] (softmax output), coef
 * [batch][P][mask_dim], priors [P][4], proto [batch][proto_h][proto_w][mask_dim].  Outputs: masks uint8 [batch][out_h][out_w], found uint8
 * [batch] (0: no detection above the score threshold -- the reference raises there and its caller keeps an all-zero mask).
 * d_workspace: amos_mask_post_workspace_bytes(...) bytes of device scratch (16-byte aligned).  Seven launches on `stream`. */
#define AMOS_MASK_NMS_TOP_K 200
#define AMOS_MASK_NMS_THRESH 0.5f
#define AMOS_MASK_CONF_THRESH 0.05f
#define AMOS_MASK_SCORE_THRESHOLD 0.15f
#define AMOS_MASK_TOP_K_DISPLAY 15
#define AMOS_MASK_PERSON_CLASS 0
size_t amos_mask_post_workspace_bytes(int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h, int proto_w);
int amos_mask_person_masks_device(void *stream, const float *d_loc, const float *d_conf, const float *d_coef, const float *d_priors,
                                  const float *d_proto, int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h,
                                  int proto_w, int out_h, int out_w, void *d_workspace, size_t workspace_bytes, uint8_t *d_masks,
                                  uint8_t *d_found);
/* The same from the class scores amos_mask_head_outputs_scores_device wrote (d_scores [batch][classes][n_priors], threshold
 * AMOS_MASK_CONF_THRESH) instead of the softmax tensor: six launches. */
int amos_mask_person_masks_scores_device(void *stream, const float *d_loc, const float *d_scores, const float *d_coef, const float *d_priors,
                                         const float *d_proto, int batch, int n_priors, int n_classes_with_background, int mask_dim, int proto_h,
                                         int proto_w, int out_h, int out_w, void *d_workspace, size_t workspace_bytes, uint8_t *d_masks,
                                         uint8_t *d_found);


/* ---------------------------------------------------------------- SLIC superpixels (8f-2) ---- */

/* ORB_SLAM2::center, include/cluster.h:21-30. */
typedef struct amos_slic_center {
    int32_t x, y; /* column, row */
    int32_t L, A, B, D;
    int32_t label; /* 1 .. n_centers */
    int32_t id;    /* k-means cluster, filled by the caller (cluster.cc:18-24); 0 here */
} amos_slic_center;

/* cluster::SLIC (src/cluster.cc:300-343) from the Lab image on -- initilizeCenters, fituneCenter (Sobel gradient),
 * `iterations` (5 in the reference) rounds of clustering + updateCenter with grid step `len` (5) and weight `m` (10).
 * The cv::cvtColor(image, imageLAB, COLOR_BGR2Lab) of cluster.cc:310 stays with the caller (its 8-bit path is table
 * driven inside OpenCV), as does the k-means over the centres that follows (cluster.cc:345-464).
 *   lab:    height x width x 3 uint8 (imageLAB);   depth: height x width uint16 (imD)
 *   labels: height x width float64 (labelMask / imLS: the label of the centre a pixel belongs to, 0 = none)
 *   centers: amos_slic_center_count(width, height, len) records in creation order (row-major grid) */
typedef struct amos_slic amos_slic;
int amos_slic_center_count(int width, int height, int len, int *nx, int *ny);
int amos_slic_create(int device, void *stream, int max_width, int max_height, int max_batch, amos_slic **out);
void amos_slic_destroy(amos_slic *s);
void *amos_slic_stream(amos_slic *s);
int amos_slic_run(amos_slic *s, const uint8_t *lab, const uint16_t *depth, int width, int height, int len, int m, int iterations,
              double *labels, amos_slic_center *centers, int *n_centers);
/* The same for n_frames resident frames: d_lab [n][h][w][3], d_depth [n][h][w], d_labels [n][h][w],
 * d_centers [n][center_count].  Asynchronous on the handle's stream. */
int amos_slic_batch_device(amos_slic *s, const uint8_t *d_lab, const uint16_t *d_depth, int width, int height, int n_frames,
                           int len, int m, int iterations, double *d_labels, amos_slic_center *d_centers);

/* cv::cvtColor(image, imageLAB, COLOR_BGR2Lab) of cluster::SLIC (src/cluster.cc:310), 8-bit: OpenCV 4.5's fixed-point RGB2Lab_b
 * (gamma and cube-root tables, 12-bit XYZ coefficients over the D65 white point).  Tables are built in double on the host
 * (OpenCV uses its softfloat: PARITY UNPINNED at the level of single table entries; primaries and grays match OpenCV's
 * documented values).  d_bgr / d_lab: n_pixels x 3 bytes; rgb_order != 0 for RGB input.  With amos_slic_batch_device and
 * amos_cluster_kmeans_batch_device this is the whole `cluster` constructor (cluster.cc:9-43) on the device. */
int amos_cluster_bgr2lab_batch_device(amos_slic *s, const uint8_t *d_bgr, size_t n_pixels, int rgb_order, uint8_t *d_lab);

/* cluster::randCent + cluster::kmeans (src/cluster.cc:353-460): the k-means over the SLIC centres that gives every
 * superpixel its cluster id (center::id, read by the label gate of amos_orb_gate through center_ids[label - 1]).
 * k = 15 in the reference (Frame.cc:525).  Distances as cluster::distEclud (:374-387); the loop runs until no
 * assignment changes (at most max_iter passes; *passes = -1 if the bound was hit).  The reference's undefined
 * behaviours are given a definition -- rand() becomes glibc's TYPE_0 generator (state * 1103515245 + 12345, low 31
 * bits) on the caller's seed, the one-past-the-end draw wraps to centre 0, the redraw on zero depth is bounded, the
 * uninitialised accumulator of the update step is zero -- so the result is a function of (centres, k, seed).
 * Centres must carry label = index + 1 (as amos_slic_* writes them).  d_centers: [n_frames][n_centers]. */
int amos_cluster_kmeans_batch_device(amos_slic *s, amos_slic_center *d_centers, int n_centers, int n_frames, int k,
                                     uint32_t seed, int max_iter, int32_t *d_passes /* [n_frames] or NULL */);
int amos_cluster_kmeans(amos_slic *s, amos_slic_center *centers, int n_centers, int k, uint32_t seed, int max_iter,
                        int *passes);

/* ---------------------------------------------------------------- scene flow, point arithmetic (8f-3) ---- */

/* The reference's own per-point arithmetic inside Tracking::GetSceneFlowObj (src/Tracking.cc:850-1186), between its
 * OpenCV calls (goodFeaturesToTrack, cornerSubPix, calcOpticalFlowPyrLK, findFundamentalMat, solvePnPRansac stay with
 * the caller).  Points are interleaved (x, y) float pairs; all pointers are device pointers; asynchronous on `stream`.
 *
 * amos_flow_check_device (:902-925): state_out[i] = 0 when either position lies within 5 px of the image edge or the
 * 3 x 3 sum of absolute gray differences between (last, pre) and (cur, next) exceeds 2520; else state_in[i].  The
 * match lists of the reference are the points with state_out != 0, in order. */
int amos_flow_check_device(void *stream, const uint8_t *d_last_gray, size_t last_stride, const uint8_t *d_cur_gray,
                           size_t cur_stride, int cols, int rows, const float *d_pre_xy, const float *d_next_xy,
                           const uint8_t *d_state_in, int n, uint8_t *d_state_out);
/* (:928-946, 1141-1152): dd[i] = |l . (next, 1)| / sqrt(l0^2 + l1^2) with l = F * (pre, 1), F row-major 3 x 3 doubles on
 * the device; -1 where d_state (may be NULL) is 0.  The reference keeps dd <= 0.5 for the second F and marks dd > 1. */
int amos_flow_epipolar_device(void *stream, const double *d_F, const float *d_pre_xy, const float *d_next_xy,
                              const uint8_t *d_state, int n, double *d_dd);
/* (:955-990, 1153-1183): per match the world point of the LAST frame's pixel (pre_3d, through mLastFrame.mTcw), of the
 * CURRENT frame's pixel (cur_3d, through mRwc / mOw), the flow norm sqrt(dx^2 + dz^2) and a validity flag (z1 > 0 &&
 * z2 > 0): out[i] = {pre_3d.xyz, cur_3d.xyz, sf_norm, valid}.  Depth maps are float32 (rows `stride` floats apart). */
typedef struct amos_scene_flow_camera {
    float cx, cy, invfx, invfy; /* Frame::cx, cy, invfx, invfy */
    float Tlw[12];              /* mLastFrame.mTcw, first three rows (row-major 3 x 4) */
    float Rwc[9], Ow[3];        /* mCurrentFrame.mRwc (row-major), mOw */
} amos_scene_flow_camera;
int amos_flow_scene_flow_device(void *stream, const float *d_depth_last, size_t last_stride, const float *d_depth_cur,
                                size_t cur_stride, const float *d_match_pre_xy, const float *d_match_cur_xy, int n,
                                const amos_scene_flow_camera *cam, float *d_out);
/* Hypothesis scoring for the three RANSACs of GetSceneFlowObj (Tracking.cc:927, 945: cv::findFundamentalMat(pre, cur, FM_RANSAC, 0.1, 0.99);
 * :1006: cv::solvePnPRansac(pre_3d, cur_2d, K, 0, rvec, tvec, false, 500, 0.4, 0.98, inliers, SOLVEPNP_P3P)).  The minimal solvers (7- / 8-point,
 * P3P), the sampling and the iteration-count update stay with the caller; the device takes what touches every point: the error of every
 * correspondence under every hypothesis (OpenCV's FMEstimatorCallback / PnPRansacCallback::computeError restated -- symmetric squared epipolar
 * distance in doubles; squared reprojection error of cv::projectPoints without distortion), the inlier test `err <= (float)(threshold^2)` and
 * the inlier count per hypothesis.  d_F: n_hypotheses x 9 doubles (row-major 3 x 3); d_Rt: n_hypotheses x 12 doubles (R row-major, then t);
 * points as float pairs / triples.  Outputs: d_inliers [n_hypotheses]; optional d_err [n_hypotheses][n] float and d_mask
 * [n_hypotheses][n] uint8 (NULL: not written).  One work-group per hypothesis, no atomics. */
int amos_flow_fundamental_score_device(void *stream, const double *d_F, int n_hypotheses, const float *d_points1_xy, const float *d_points2_xy,
                                       int n, double threshold, float *d_err, int32_t *d_inliers, uint8_t *d_mask);
int amos_flow_pnp_score_device(void *stream, const double *d_Rt, int n_hypotheses, const float *d_object_xyz, const float *d_image_xy, int n,
                               double fx, double fy, double cx, double cy, double reprojection_error, float *d_err, int32_t *d_inliers,
                               uint8_t *d_mask);

/* cv::calcOpticalFlowPyrLK(imlast, gray, prepoint, nextpoint, state, err, Size(22, 22), 5, TermCriteria(COUNT | EPS, 20, 0.01))
 * (Tracking.cc:896) on given points: pyramids of both gray frames (pyrDown, REFLECT_101 borders of the window size), Scharr
 * derivatives of the previous one, the iterative tracker from the top level down -- one wave per point.  Restated from OpenCV
 * 4.5's published algorithm; the float accumulation order is defined as the scalar path's (row by row, left to right) because
 * OpenCV's own depends on the SIMD width of its build: PARITY UNPINNED.  win_size <= 22.
 * amos_lk_levels = buildOpticalFlowPyramid's return (4 for 640 x 480: the 20 x 15 level is smaller than the window). */
typedef struct amos_lk amos_lk;
int amos_lk_create(int device, void *stream, int width, int height, int win_size, int max_level, amos_lk **out);
void amos_lk_destroy(amos_lk *k);
void *amos_lk_stream(amos_lk *k);
int amos_lk_levels(const amos_lk *k);
int amos_lk_track_device(amos_lk *k, const uint8_t *d_prev_gray, size_t prev_stride, const uint8_t *d_next_gray,
                         size_t next_stride, const float *d_prev_xy, int n, int max_count, double epsilon,
                         float min_eig_threshold, float *d_next_xy, uint8_t *d_status, float *d_err /* or NULL */);

/* The corner source of Tracking::GetSceneFlowObj (Tracking.cc:894-895), resident on the device so that the corners go into
 * amos_lk_track_device without a host round trip:
 *   cv::goodFeaturesToTrack(imlast, prepoint, 1000, 0.01, 8, cv::Mat(), 3, true, 0.04)   -> amos_corners_good_features_device
 *   cv::cornerSubPix(imlast, prepoint, Size(10, 10), Size(-1, -1), TermCriteria(ITER | EPS, 20, 0.03)) -> amos_corners_subpix_device
 * Harris detector, blockSize 3, Sobel aperture 3, no mask (what the reference passes).  d_xy receives (x, y) pairs in the library's
 * order (strongest first, equal responses: the later pixel first), d_count their number (<= max_corners, <= xy_capacity);
 * d_response (or NULL) the Harris response plane [height][width].  amos_corners_subpix_device refines d_xy in place for the first
 * *d_count points (d_count == NULL: n points).  Restated from OpenCV 4.5's published algorithms; PARITY UNPINNED like the other
 * OpenCV-derived stages.  A frame with more than 65 536 local maxima above the quality threshold is truncated:
 * amos_corners_candidate_count (synchronising) reports AMOS_ERR_CAPACITY then.  Asynchronous on the handle's stream. */
typedef struct amos_corners amos_corners;
int amos_corners_create(int device, void *stream, int max_width, int max_height, amos_corners **out);
void amos_corners_destroy(amos_corners *c);
void *amos_corners_stream(amos_corners *c);
int amos_corners_good_features_device(amos_corners *c, const uint8_t *d_gray, size_t stride, int width, int height, int max_corners,
                                      double quality_level, double min_distance, double harris_k, float *d_xy, int xy_capacity,
                                      int *d_count, float *d_response /* or NULL */);
int amos_corners_candidate_count(amos_corners *c, int *count);
int amos_corners_subpix_device(amos_corners *c, const uint8_t *d_gray, size_t stride, int width, int height, float *d_xy,
                               const int *d_count /* or NULL */, int n, int win, int max_count, double epsilon);

#ifdef __cplusplus
}
#endif
#endif /* AMOS_FRONTEND_H */
