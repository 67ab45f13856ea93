/*
 * orb_oracle.h -- CPU restatement of the Amos-SLAM front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this.  The product path (amos-slam_amd/, include/) never does.
 *
 * PARITY STATUS: the pixel arithmetic of this path lives in OpenCV (pinned 4.5.1 by the reference,
 * CMakeLists.txt:31), which is absent from the reference tree and from this image, and the
 * reference holds no golden vectors for the path.  The stages that restate OpenCV primitives
 * (resize, copyMakeBorder, FAST, GaussianBlur, fastAtan2, dilate/erode) are therefore
 * "PARITY UNPINNED": they follow OpenCV 4.5's published algorithms as recorded in SURVEY.md
 * Appendix A.  Pinned by the reference itself: the rBRIEF pattern (sha256), umax, the per-level
 * quotas, thresholds, and every line of ORBextractor.cc / ORBmatcher.cc logic restated here.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/amos_frontend.h" /* amos_keypoint, amos_best2, amos_orb_params: shared PODs */
#include "../include/amos_host_types.h" /* amos_frame_view, amos_proj_query, amos_map_query */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor, ORBextractor.cc:492-609 */
orc_extractor *orc_create(const amos_orb_params *p);
void orc_destroy(orc_extractor *e);
void orc_tables(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2,
                float *inv_sigma2, int32_t *features_per_level, int32_t *umax);
int orc_level_sizes(const orc_extractor *e, int width, int height, int32_t *lw, int32_t *lh);

/* 3-arg operator(), ORBextractor.cc:1672-1686.  Returns 0 or a negative error. */
int orc_detect(orc_extractor *e, const uint8_t *gray, size_t stride, int width, int height);
int orc_level_count(const orc_extractor *e, int level);
int orc_level_keypoints(const orc_extractor *e, int level, amos_keypoint *out, int cap);
int orc_set_level_keypoints(orc_extractor *e, int level, const amos_keypoint *kps, int n);
int orc_level_candidates(const orc_extractor *e, int level, amos_keypoint *out, int cap);
/* mvImagePyramid[level]; padded != 0 returns the (w+38)x(h+38) buffer. */
int orc_level_image(const orc_extractor *e, int level, uint8_t *dst, size_t dst_stride, int padded);
int orc_blurred_image(const orc_extractor *e, int level, uint8_t *dst, size_t dst_stride);

/* MovingKeyPoints, ORBextractor.cc:1688-1745 */
int orc_gate(orc_extractor *e, const uint8_t *mask, size_t mask_stride, const double *labels,
             size_t lstride, const int32_t *center_ids, int n_centers, const int32_t *rm_vector,
             int n_rm, amos_keypoint *removed, int cap, int *n_removed);
int orc_closed_mask(const orc_extractor *e, uint8_t *dst, size_t dst_stride);

/* ProcessDesp, ORBextractor.cc:1747-1820 */
int orc_describe(orc_extractor *e, amos_keypoint *kps, uint8_t *desc, int cap, int *n);
/* 4-arg operator(), ORBextractor.cc:1544-1668 */
int orc_extract(orc_extractor *e, const uint8_t *gray, size_t stride, int width, int height,
                amos_keypoint *kps, uint8_t *desc, int cap, int *n);

/* ---- primitives exposed for unit tests ---- */
/* cv::resize(..., INTER_LINEAR) on 8UC1, SURVEY Appendix A.1 */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw,
                          int dh, size_t dstride);
/* cv::FAST(img, kps, threshold, true), TYPE_9_16, SURVEY Appendix A.3.  Returns the count. */
int orc_fast9_16(const uint8_t *img, size_t stride, int w, int h, int threshold,
                 amos_keypoint *out, int cap);
/* cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) on a continuous 8UC1 image, Appendix A.2 */
void orc_gaussian_blur7(const uint8_t *src, size_t sstride, int w, int h, uint8_t *dst,
                        size_t dstride);
/* cv::fastAtan2, Appendix A.4 */
float orc_fast_atan2(float y, float x);
/* sincosf as glibc computes it (ARM optimized-routines algorithm, double arithmetic, no FMA) */
void orc_sincosf(float x, float *s, float *c);
/* DistributeOctTree, ORBextractor.cc:706-1049.  pts: x,y,response used.  Returns count. */
int orc_distribute_octree(const amos_keypoint *pts, int n, int minX, int maxX, int minY, int maxY,
                          int N, amos_keypoint *out, int cap);
/* dilate then erode with the 31x31 MORPH_ELLIPSE element, Appendix A.6 */
void orc_close_ellipse31(const uint8_t *src, size_t sstride, int w, int h, uint8_t *dst,
                         size_t dstride);

/* ---- matcher ---- */
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1913-1933 */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);
void orc_distances(const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *out);
void orc_list_distances(const uint8_t *q, int nq, const uint8_t *t, const int32_t *cand_off,
                        const int32_t *cand_idx, uint16_t *out);
/* The inner loop of ORBmatcher::SearchByProjection (ORBmatcher.cc:127-148) over candidate lists. */
void orc_list_best2(const uint8_t *q, int nq, const uint8_t *t, const int32_t *cand_off,
                    const int32_t *cand_idx, int init_dist, amos_best2 *out);
void orc_bruteforce_best2(const uint8_t *q, int nq, const uint8_t *t, int nt, int init_dist,
                          amos_best2 *out);
/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1866-1908, on bin sizes. */
void orc_three_maxima(const int32_t *histo_sizes, int L, int *ind1, int *ind2, int *ind3);

/* ---- callers either side of the path ---- */
/* cv::cvtColor(src, dst, CV_{BGR,RGB}[A]2GRAY), 8-bit (Tracking.cc:308-321): OpenCV 4.x 15-bit coefficients
 * (RY15 9798, GY15 19235, BY15 3735), dst = (sum + 16384) >> 15.  PARITY UNPINNED (OpenCV primitive). */
void orc_color_to_gray(const uint8_t *src, size_t sstride, int w, int h, int channels, int rgb_order, uint8_t *dst,
                       size_t dstride);
/* Tracking's imDepth.convertTo(CV_32F, factor) on one 16-bit value */
float orc_depth_convert(uint16_t raw, float factor);
/* Frame::ComputeStereoFromRGBD (Frame.cc:1576-1615) + PosInGrid cell (Frame.cc:1007-1030), mvKeysUn == mvKeys */
void orc_undistort_points(const float *xy, int n, float fx, float fy, float cx, float cy, const float *dist, int n_dist, float *out);
void orc_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist, int n_dist, float *bounds);
void orc_rgbd_glue(const amos_keypoint *kps, const amos_keypoint *kps_un, int n, const float *depth, size_t depth_stride_elems, int w, int h,
                   float mbf, float min_x, float max_x, float min_y, float max_y, float *u_right, float *depth_out, int32_t *grid_cell);

/* ---- gated searches of the tracking thread, over plain frame views ---- */
/* Frame::GetFeaturesInArea (Frame.cc:894-1003) on a grid built as Frame::AssignFeaturesToGrid does
 * (Frame.cc:431-461).  Returns the number of indices written. */
int orc_features_in_area(const amos_frame_view *f, float x, float y, float r, int min_level, int max_level,
                         int32_t *out, int cap);
/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), ORBmatcher.cc:1569-1728 */
int orc_search_by_projection_frame(const amos_frame_view *cur, const amos_proj_query *q, int nq, int32_t *cur_match,
                                   const float *scale_factors, float mbf, float th, int forward, int backward,
                                   int check_orientation);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), ORBmatcher.cc:70-175 */
int orc_search_by_projection_points(const amos_frame_view *f, const amos_map_query *q, int nq, int32_t *cur_match,
                                    uint8_t *cur_has_obs, const float *scale_factors, float th, float nn_ratio);
/* ORBmatcher::SearchForInitialization, ORBmatcher.cc:515-643.  prev_matched: n1 x 2 floats, in/out. */
int orc_search_for_initialization(const amos_frame_view *f1, const amos_frame_view *f2, float *prev_matched,
                                  int32_t *matches12, int window_size, float nn_ratio, int check_orientation);

void orc_window_best2(const amos_frame_view *train, const amos_keypoint *qk, const uint8_t *qdesc, int nq, const float *query_uv,
                      const float *query_invz, const float *scale_factors, float th, float mbf, int mode, int init_dist,
                      amos_best2 *out);

int orc_search_by_projection_kf(const amos_frame_view *cur, const amos_kf_query *q, int nq, int32_t *cur_match,
                                const float *scale_factors, float th, int orb_dist, int check_orientation);
int orc_search_by_bow(const amos_bow_view *kf, const amos_bow_view *f, int32_t *matches_f, float nn_ratio, int check_orientation);

int orc_search_by_bow_kf(const amos_bow_view *k1, const amos_bow_view *k2, int32_t *matches12, float nn_ratio, int check_orientation);
int orc_search_for_triangulation(const amos_bow_view *k1, const amos_bow_view *k2, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *pairs, int cap);

int orc_window_search(const amos_frame_view *kf, const amos_window_query *q, int nq, const float *scale_factors,
                      const float *inv_level_sigma2, float th, int max_dist, int32_t *occupied, int32_t *best_idx);
int orc_search_by_sim3(const amos_frame_view *kf1, const amos_frame_view *kf2, const amos_window_query *q12, int n12,
                       const amos_window_query *q21, int n21, const float *scale_factors1, const float *scale_factors2, float th,
                       int32_t *matches12);

int orc_slic(const uint8_t *lab, const uint16_t *depth, int w, int h, int len, int m, int iterations, double *labels,
             amos_slic_center *centers, int cap);

/* cluster::randCent + kmeans with a seeded generator (see orb_oracle.c): writes centers[].id; returns the passes made. */
int orc_kmeans(amos_slic_center *centers, int n, int k, uint32_t seed, int max_iter);

#ifdef __cplusplus
}
#endif
#endif
