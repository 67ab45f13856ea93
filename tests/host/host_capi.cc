// host_capi.cc -- C entry points that drive the C++ drop-in classes (ORB_SLAM2::ORBextractor,
// ORB_SLAM2::ORBmatcher, ORB_SLAM2::yolact of amos-slam_amd/host/libamos_host.so) so that the Python parity
// tests and bench.py's drop-in latency leg can exercise the reference-shaped API itself, not only the
// C ABI underneath it.  Test harness: built as tests/host/libamos_host_test.so, links the product
// library, never the other way round.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amos_host_types.h"
#include "../../amos-slam_amd/host/ORBextractor.h"
#include "../../amos-slam_amd/host/ORBmatcher.h"
#include "../../amos-slam_amd/host/ORBmatcher_adaptors.h"
#include "ref_standins.h"
#include "../../amos-slam_amd/host/yolact.h"

using namespace ORB_SLAM2;

// Every member of the reference-signature matcher is compiled against the stand-in classes (the reference tree
// instantiates the same template with its own Frame / KeyFrame / MapPoint).
template class ORB_SLAM2::ORBmatcherFor<amos_standins::Frame, amos_standins::KeyFrame, amos_standins::MapPoint>;
typedef ORB_SLAM2::ORBmatcherFor<amos_standins::Frame, amos_standins::KeyFrame, amos_standins::MapPoint> RefMatcher;

static thread_local std::string g_host_error;

#define AMOS_HOST_TRY try {
#define AMOS_HOST_CATCH                                        \
    }                                                          \
    catch (const std::exception &e) {                          \
        g_host_error = e.what();                               \
        return -100;                                           \
    }

extern "C" {

const char *amos_host_last_error(void) { return g_host_error.c_str(); }

// 4-arg operator(): ORBextractor(...)(image, Mat(), keypoints, descriptors); optionally mvImagePyramid[level]
int amos_host_extract(const uint8_t *gray, int w, int h, int nfeatures, float scale, int nlevels, int ini, int min, amos_keypoint *kps,
                      uint8_t *desc, int cap, int *n, int pyr_level, uint8_t *pyr_out /* (w_l+38)*(h_l+38) or NULL */)
{
    AMOS_HOST_TRY
    ORBextractor ext(nfeatures, scale, nlevels, ini, min);
    cv::Mat image(h, w, CV_8UC1, (void *)gray, (size_t)w), mask, descriptors;
    std::vector<cv::KeyPoint> keys;
    ext(image, mask, keys, descriptors);
    *n = (int)keys.size();
    if (*n > cap) return -3;
    if (*n) std::memcpy(kps, keys.data(), sizeof(amos_keypoint) * keys.size());
    for (int i = 0; i < *n; i++) std::memcpy(desc + 32 * (size_t)i, descriptors.ptr(i), 32);
    if (descriptors.empty() != (*n == 0)) return -101;
    if (pyr_out) {
        // the ROI must sit inside its padded buffer: read through negative offsets like IC_Angle does
        const cv::Mat &m = ext.mvImagePyramid[pyr_level];
        const unsigned char *origin = m.data - (size_t)AMOS_EDGE_THRESHOLD * m.step - AMOS_EDGE_THRESHOLD;
        for (int y = 0; y < m.rows + 2 * AMOS_EDGE_THRESHOLD; y++)
            std::memcpy(pyr_out + (size_t)y * (m.cols + 2 * AMOS_EDGE_THRESHOLD), origin + (size_t)y * m.step, m.cols + 2 * AMOS_EDGE_THRESHOLD);
    }
    return 0;
    AMOS_HOST_CATCH
}

// The live RGB-D flow of Frame.cc:480-496,633: 3-arg operator() -> MovingKeyPoints -> ProcessDesp
int amos_host_amos_flow(const uint8_t *gray, int w, int h, int nfeatures, float scale, int nlevels, int ini, int min, const uint8_t *mask,
                        const double *labels, const int32_t *center_ids, int ncenters, const int32_t *rm, int nrm, amos_keypoint *removed,
                        int *nremoved, amos_keypoint *kps, uint8_t *desc, int cap, int *n, amos_keypoint *level_lists_after /* cap */,
                        int32_t *level_counts_after)
{
    AMOS_HOST_TRY
    ORBextractor ext(nfeatures, scale, nlevels, ini, min);
    ext.SetPyramidDownload(false);
    cv::Mat image(h, w, CV_8UC1, (void *)gray, (size_t)w), none;
    std::vector<std::vector<cv::KeyPoint>> mvKeysTemp;
    ext(image, none, mvKeysTemp);
    cv::Mat imS(h, w, CV_8UC1, (void *)mask, (size_t)w);
    cv::Mat imLS = labels ? cv::Mat(h, w, CV_64FC1, (void *)labels, (size_t)w * 8) : cv::Mat();
    std::vector<center> centers(ncenters);
    for (int i = 0; i < ncenters; i++) centers[i].id = center_ids[i];
    std::vector<int> rmv(rm, rm + nrm);
    std::vector<cv::KeyPoint> dyn = ext.MovingKeyPoints(image, imS, imLS, centers, rmv, std::vector<bool>(), mvKeysTemp);
    *nremoved = (int)dyn.size();
    if (*nremoved > cap) return -3;
    if (*nremoved) std::memcpy(removed, dyn.data(), sizeof(amos_keypoint) * dyn.size());
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    ext.ProcessDesp(image, none, mvKeysTemp, mvKeys, mDescriptors);
    *n = (int)mvKeys.size();
    if (*n > cap) return -3;
    if (*n) std::memcpy(kps, mvKeys.data(), sizeof(amos_keypoint) * mvKeys.size());
    for (int i = 0; i < *n; i++) std::memcpy(desc + 32 * (size_t)i, mDescriptors.ptr(i), 32);
    int o = 0;
    for (int l = 0; l < nlevels; l++) {  // the caller's vectors after ProcessDesp (rescaled in place)
        level_counts_after[l] = (int)mvKeysTemp[l].size();
        for (const cv::KeyPoint &kp : mvKeysTemp[l]) std::memcpy(&level_lists_after[o++], &kp, sizeof(amos_keypoint));
    }
    return 0;
    AMOS_HOST_CATCH
}

// mvImagePyramid under the default PYRAMID_AUTO mode after the 3-arg operator() (the RGB-D Amos flow): every level's rows / cols are
// valid although no pixel was copied (Frame.cc:1197 reads mvImagePyramid[0].rows); DownloadPyramid() then fills them on demand.
// rows_cols: 2 x nlevels; pyr_out: the padded plane of pyr_level after DownloadPyramid(); device_out: the extractor's device.
int amos_host_pyramid_on_demand(const uint8_t *gray, int w, int h, int nlevels, int32_t *rows_cols, int pyr_level, uint8_t *pyr_out, int32_t *device_out)
{
    AMOS_HOST_TRY
    ORBextractor ext(1000, 1.2f, nlevels, 20, 7);
    cv::Mat image(h, w, CV_8UC1, (void *)gray, (size_t)w), none;
    std::vector<std::vector<cv::KeyPoint> > levels;
    ext(image, none, levels);
    *device_out = ext.GetDevice();
    for (int l = 0; l < nlevels; l++) {
        rows_cols[2 * l] = ext.mvImagePyramid[l].rows;
        rows_cols[2 * l + 1] = ext.mvImagePyramid[l].cols;
    }
    ext.DownloadPyramid();
    const cv::Mat &m = ext.mvImagePyramid[pyr_level];
    const unsigned char *origin = m.data - (size_t)AMOS_EDGE_THRESHOLD * m.step - AMOS_EDGE_THRESHOLD;
    for (int y = 0; y < m.rows + 2 * AMOS_EDGE_THRESHOLD; y++)
        std::memcpy(pyr_out + (size_t)y * (m.cols + 2 * AMOS_EDGE_THRESHOLD), origin + (size_t)y * m.step, m.cols + 2 * AMOS_EDGE_THRESHOLD);
    return 0;
    AMOS_HOST_CATCH
}

int amos_host_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    AMOS_HOST_TRY
    cv::Mat ma(1, 32, CV_8U, (void *)a), mb(1, 32, CV_8U, (void *)b);
    return ORBmatcher::DescriptorDistance(ma, mb);
    AMOS_HOST_CATCH
}

int amos_host_features_in_area(const amos_frame_view *f, float x, float y, float r, int min_level, int max_level, int32_t *out, int cap)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*f);
    std::vector<size_t> v = grid.GetFeaturesInArea(x, y, r, min_level, max_level);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int32_t)v[i];
    return (int)v.size();
    AMOS_HOST_CATCH
}

int amos_host_search_by_projection_frame(const amos_frame_view *cur, const amos_proj_query *q, int nq, int32_t *cur_match,
                                         const float *scale_factors, int nsf, float mbf, float th, int forward, int backward, float nnratio,
                                         int check_orientation)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*cur);
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<amos_proj_query> pts(q, q + nq);
    std::vector<int> match(cur_match, cur_match + cur->n);
    std::vector<float> sf(scale_factors, scale_factors + nsf);
    const int r = matcher.SearchByProjection(grid, pts, match, sf, mbf, th, forward != 0, backward != 0);
    std::memcpy(cur_match, match.data(), sizeof(int) * cur->n);
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_projection_points(const amos_frame_view *f, const amos_map_query *q, int nq, int32_t *cur_match, uint8_t *cur_has_obs,
                                          const float *scale_factors, int nsf, float th, float nnratio)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*f);
    ORBmatcher matcher(nnratio, true);
    std::vector<amos_map_query> pts(q, q + nq);
    std::vector<int> match(cur_match, cur_match + f->n);
    std::vector<bool> obs(f->n);
    for (int i = 0; i < f->n; i++) obs[i] = cur_has_obs[i] != 0;
    std::vector<float> sf(scale_factors, scale_factors + nsf);
    const int r = matcher.SearchByProjection(grid, pts, match, obs, sf, th);
    std::memcpy(cur_match, match.data(), sizeof(int) * f->n);
    for (int i = 0; i < f->n; i++) cur_has_obs[i] = obs[i];
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_projection_kf(const amos_frame_view *cur, const amos_kf_query *q, int nq, int32_t *cur_match, const float *scale_factors,
                                      int nsf, float th, int orb_dist, float nnratio, int check_orientation)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*cur);
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<amos_kf_query> pts(q, q + nq);
    std::vector<int> match(cur_match, cur_match + cur->n);
    std::vector<float> sf(scale_factors, scale_factors + nsf);
    const int r = matcher.SearchByProjection(grid, pts, match, sf, th, orb_dist);
    std::memcpy(cur_match, match.data(), sizeof(int) * cur->n);
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_bow(const amos_bow_view *kf, const amos_bow_view *f, int32_t *matches_f, float nnratio, int check_orientation)
{
    AMOS_HOST_TRY
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<int> m;
    const int r = matcher.SearchByBoW(*kf, *f, m);
    for (int i = 0; i < f->n; i++) matches_f[i] = m[i];
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_bow_kf(const amos_bow_view *kf1, const amos_bow_view *kf2, int32_t *matches12, float nnratio, int check_orientation)
{
    AMOS_HOST_TRY
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<int> m;
    const int r = matcher.SearchByBoW(*kf1, *kf2, m, true);
    for (int i = 0; i < kf1->n; i++) matches12[i] = m[i];
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_for_triangulation(const amos_bow_view *kf1, const amos_bow_view *kf2, const float *f12, float ex, float ey,
                                       const float *scale_factors2, const float *level_sigma2_2, int nlevels, int only_stereo, float nnratio,
                                       int check_orientation, int32_t *pairs, int cap)
{
    AMOS_HOST_TRY
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<float> sf(scale_factors2, scale_factors2 + nlevels), sg(level_sigma2_2, level_sigma2_2 + nlevels);
    std::vector<std::pair<size_t, size_t> > out;
    const int r = matcher.SearchForTriangulation(*kf1, *kf2, f12, ex, ey, sf, sg, out, only_stereo != 0);
    if ((int)out.size() > cap) return -3;
    for (size_t i = 0; i < out.size(); i++) { pairs[2 * i] = (int32_t)out[i].first; pairs[2 * i + 1] = (int32_t)out[i].second; }
    return r;
    AMOS_HOST_CATCH
}

int amos_host_fuse(const amos_frame_view *kf, const amos_window_query *q, int nq, const float *scale_factors, const float *inv_level_sigma2,
                   int nlevels, float th, int32_t *best_idx)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*kf);
    ORBmatcher matcher;
    std::vector<amos_window_query> pts(q, q + nq);
    std::vector<float> sf(scale_factors, scale_factors + nlevels);
    std::vector<int> best;
    int r;
    if (inv_level_sigma2) {
        std::vector<float> is2(inv_level_sigma2, inv_level_sigma2 + nlevels);
        r = matcher.Fuse(grid, pts, sf, is2, th, best);
    } else {
        r = matcher.Fuse(grid, pts, sf, th, best);
    }
    for (int i = 0; i < nq; i++) best_idx[i] = best[i];
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_projection_sim(const amos_frame_view *kf, const amos_window_query *q, int nq, int32_t *matched, const float *scale_factors,
                                       int nlevels, int th)
{
    AMOS_HOST_TRY
    FeatureGrid grid(*kf);
    ORBmatcher matcher;
    std::vector<amos_window_query> pts(q, q + nq);
    std::vector<float> sf(scale_factors, scale_factors + nlevels);
    std::vector<int> m(matched, matched + kf->n);
    const int r = matcher.SearchByProjection(grid, pts, m, sf, th);
    std::memcpy(matched, m.data(), sizeof(int) * kf->n);
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_by_sim3(const amos_frame_view *kf1, const amos_frame_view *kf2, const amos_window_query *q12, int n12,
                             const amos_window_query *q21, int n21, const float *scale_factors1, const float *scale_factors2, int nlevels,
                             float th, int32_t *matches12)
{
    AMOS_HOST_TRY
    FeatureGrid g1(*kf1), g2(*kf2);
    ORBmatcher matcher;
    std::vector<amos_window_query> a(q12, q12 + n12), b(q21, q21 + n21);
    std::vector<float> sf1(scale_factors1, scale_factors1 + nlevels), sf2(scale_factors2, scale_factors2 + nlevels);
    std::vector<int> m;
    const int r = matcher.SearchBySim3(g1, g2, a, b, sf1, sf2, m, th);
    for (int i = 0; i < kf1->n; i++) matches12[i] = m[i];
    return r;
    AMOS_HOST_CATCH
}

int amos_host_search_for_initialization(const amos_frame_view *f1, const amos_frame_view *f2, float *prev_matched, int32_t *matches12,
                                        int window_size, float nnratio, int check_orientation)
{
    AMOS_HOST_TRY
    FeatureGrid grid2(*f2);
    ORBmatcher matcher(nnratio, check_orientation != 0);
    std::vector<cv::Point2f> prev(f1->n);
    for (int i = 0; i < f1->n; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
    std::vector<int> m12;
    const int r = matcher.SearchForInitialization(*f1, grid2, prev, m12, window_size);
    for (int i = 0; i < f1->n; i++) {
        matches12[i] = m12[i];
        prev_matched[2 * i] = prev[i].x;
        prev_matched[2 * i + 1] = prev[i].y;
    }
    return r;
    AMOS_HOST_CATCH
}

// ORB_SLAM2::yolact: construct once per (py file, weights), evaluate one BGR frame
// ---- the reference-signature adaptors (ORBmatcher_adaptors.h) on stand-in Frame / MapPoint objects built from arrays

struct amos_test_camera {
    float fx, fy, cx, cy, mb, mbf;
    float min_x, max_x, min_y, max_y;
    float Tcw[16];
    int32_t n_levels;
    float scale_factors[AMOS_MAX_LEVELS];
};

static void fill_frame(amos_standins::Frame &F, const amos_test_camera *cam, int n, const amos_keypoint *keys, const amos_keypoint *keys_un,
                       const uint8_t *desc, const float *u_right)
{
    F.N = n;
    F.mvKeys.resize(n);
    F.mvKeysUn.resize(n);
    if (n) {
        std::memcpy(F.mvKeys.data(), keys, sizeof(amos_keypoint) * n);
        std::memcpy(F.mvKeysUn.data(), keys_un, sizeof(amos_keypoint) * n);
    }
    F.mDescriptors = cv::Mat(std::max(n, 1), 32, CV_8U);
    if (n) std::memcpy(F.mDescriptors.data, desc, (size_t)32 * n);
    if (u_right) F.mvuRight.assign(u_right, u_right + n);
    F.mvpMapPoints.assign(n, nullptr);
    F.mvbOutlier.assign(n, false);
    F.fx = cam->fx; F.fy = cam->fy; F.cx = cam->cx; F.cy = cam->cy; F.mb = cam->mb; F.mbf = cam->mbf;
    F.mnMinX = cam->min_x; F.mnMaxX = cam->max_x; F.mnMinY = cam->min_y; F.mnMaxY = cam->max_y;
    F.mTcw = cv::Mat(4, 4, CV_32F);
    std::memcpy(F.mTcw.data, cam->Tcw, sizeof(float) * 16);
    F.mnScaleLevels = cam->n_levels;
    F.mvScaleFactors.assign(cam->scale_factors, cam->scale_factors + cam->n_levels);
    F.mfLogScaleFactor = cam->n_levels > 1 ? std::log(cam->scale_factors[1]) : 1.f;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) with the reference's signature.
// Last frame: per feature has_point / outlier / world position / descriptor / observation count of its map point.
// cur_occupant_obs[i2]: -1 = CurrentFrame.mvpMapPoints[i2] NULL on entry, else the Observations() of an occupant.
// Out: cur_match[i2] = index of the last-frame feature whose map point ends up in CurrentFrame.mvpMapPoints[i2],
// -1 for NULL, -2 for an untouched occupant.
int amos_host_ref_search_last_frame(const amos_test_camera *cur_cam, int n_cur, const amos_keypoint *cur_keys_un, const uint8_t *cur_desc,
                                    const float *cur_u_right, const int32_t *cur_occupant_obs, const amos_test_camera *last_cam, int n_last,
                                    const amos_keypoint *last_keys, const amos_keypoint *last_keys_un, const uint8_t *last_has_point,
                                    const uint8_t *last_outlier, const float *last_world /* n x 3 */, const uint8_t *last_mp_desc /* n x 32 */,
                                    const int32_t *last_mp_obs, float th, int mono, float nnratio, int check_orientation, int32_t *cur_match)
{
    AMOS_HOST_TRY
    using namespace amos_standins;
    Frame Cur, Last;
    fill_frame(Cur, cur_cam, n_cur, cur_keys_un, cur_keys_un, cur_desc, cur_u_right);
    std::vector<uint8_t> none((size_t)32 * std::max(n_last, 1), 0);
    fill_frame(Last, last_cam, n_last, last_keys, last_keys_un, none.data(), nullptr);
    std::vector<MapPoint> pts(n_last), occupants(n_cur);
    for (int i = 0; i < n_last; i++) {
        if (!last_has_point[i]) continue;
        MapPoint &p = pts[i];
        for (int k = 0; k < 3; k++) p.mWorldPos.at<float>(k, 0) = last_world[3 * i + k];
        std::memcpy(p.mDescriptor.data, last_mp_desc + 32 * (size_t)i, 32);
        p.mnObs = last_mp_obs[i];
        Last.mvpMapPoints[i] = &p;
        Last.mvbOutlier[i] = last_outlier[i] != 0;
    }
    for (int i2 = 0; i2 < n_cur; i2++)
        if (cur_occupant_obs && cur_occupant_obs[i2] >= 0) {
            occupants[i2].mnObs = cur_occupant_obs[i2];
            Cur.mvpMapPoints[i2] = &occupants[i2];
        }
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByProjection(Cur, Last, th, mono != 0);
    for (int i2 = 0; i2 < n_cur; i2++) {
        MapPoint *p = Cur.mvpMapPoints[i2];
        if (!p) cur_match[i2] = -1;
        else if (p >= pts.data() && p < pts.data() + n_last) cur_match[i2] = (int32_t)(p - pts.data());
        else cur_match[i2] = -2;
    }
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) with the reference's signature:
// the map points carry the mTrack* members Tracking::SearchLocalPoints fills.  cur_match[i] = index into the point list or -1.
int amos_host_ref_search_local_points(const amos_test_camera *cam, int n, const amos_keypoint *keys_un, const uint8_t *desc, const float *u_right,
                                      const amos_map_query *points, const uint8_t *in_view, const uint8_t *bad, int n_points, float th, float nnratio,
                                      int32_t *cur_match)
{
    AMOS_HOST_TRY
    using namespace amos_standins;
    Frame F;
    fill_frame(F, cam, n, keys_un, keys_un, desc, u_right);
    std::vector<MapPoint> pts(n_points);
    std::vector<MapPoint *> vp(n_points);
    for (int i = 0; i < n_points; i++) {
        MapPoint &p = pts[i];
        p.mbTrackInView = in_view[i] != 0;
        p.mbBad = bad[i] != 0;
        p.mTrackProjX = points[i].proj_x;
        p.mTrackProjY = points[i].proj_y;
        p.mTrackProjXR = points[i].proj_xr;
        p.mTrackViewCos = points[i].view_cos;
        p.mnTrackScaleLevel = points[i].level;
        p.mnObs = points[i].has_obs;
        std::memcpy(p.mDescriptor.data, points[i].desc, 32);
        vp[i] = &p;
    }
    RefMatcher matcher(nnratio, true);
    const int nm = matcher.SearchByProjection(F, vp, th);
    for (int i = 0; i < n; i++) cur_match[i] = F.mvpMapPoints[i] ? (int32_t)(F.mvpMapPoints[i] - pts.data()) : -1;
    return nm;
    AMOS_HOST_CATCH
}

// ---- the other eight reference signatures on a small stand-in "map": a table of map points, keyframes / frames whose features point
// into it, DBoW2-style feature vectors.  Map points are reported back as indices into the table (-1 = NULL).

struct amos_test_points {
    int32_t n;
    const float *world, *normal;   // n x 3
    const uint8_t *desc;           // n x 32
    const int32_t *obs;            // Observations()
    const uint8_t *bad;
    const float *min_dist, *max_dist;  // mfMinDistance / mfMaxDistance (GetMin/MaxDistanceInvariance scale them by 0.8 / 1.2)
};

struct amos_test_kf {
    amos_test_camera cam;          // cam.Tcw: the pose (a Frame's mTcw, a KeyFrame's Tcw; Ow = -Rcw^T tcw as KeyFrame::SetPose computes it)
    int32_t n;
    const amos_keypoint *keys, *keys_un;
    const uint8_t *desc;
    const float *u_right;          // NULL: monocular (mvuRight all -1)
    const int32_t *point_of;       // n: index into the point table or -1 (mvpMapPoints)
    int32_t n_nodes;               // mFeatVec
    const uint32_t *node_ids;
    const int32_t *node_off, *node_idx;
};

}  // extern "C"

namespace
{
using namespace amos_standins;

std::vector<MapPoint> make_points(const amos_test_points *t)
{
    std::vector<MapPoint> pts(t->n);
    for (int i = 0; i < t->n; i++) {
        MapPoint &p = pts[i];
        for (int k = 0; k < 3; k++) {
            p.mWorldPos.at<float>(k, 0) = t->world[3 * i + k];
            p.mNormal.at<float>(k, 0) = t->normal ? t->normal[3 * i + k] : 0.f;
        }
        std::memcpy(p.mDescriptor.data, t->desc + 32 * (size_t)i, 32);
        p.mnObs = t->obs ? t->obs[i] : 1;
        p.mbBad = t->bad && t->bad[i];
        p.mfMinDistance = t->min_dist ? t->min_dist[i] : 0.f;
        p.mfMaxDistance = t->max_dist ? t->max_dist[i] : 1e9f;
    }
    return pts;
}

template <class F>
void fill_base(F &f, const amos_test_kf *k)
{
    const int n = k->n;
    f.N = n;
    f.mvKeys.resize(n);
    f.mvKeysUn.resize(n);
    if (n) {
        std::memcpy(f.mvKeys.data(), k->keys, sizeof(amos_keypoint) * n);
        std::memcpy(f.mvKeysUn.data(), k->keys_un, sizeof(amos_keypoint) * n);
    }
    f.mDescriptors = cv::Mat(std::max(n, 1), 32, CV_8U);
    if (n) std::memcpy(f.mDescriptors.data, k->desc, (size_t)32 * n);
    if (k->u_right) f.mvuRight.assign(k->u_right, k->u_right + n);
    else f.mvuRight.assign(n, -1.f);
    const amos_test_camera &c = k->cam;
    f.fx = c.fx; f.fy = c.fy; f.cx = c.cx; f.cy = c.cy; f.mb = c.mb; f.mbf = c.mbf;
    f.mnMinX = c.min_x; f.mnMaxX = c.max_x; f.mnMinY = c.min_y; f.mnMaxY = c.max_y;
    f.mnScaleLevels = c.n_levels;
    f.mvScaleFactors.assign(c.scale_factors, c.scale_factors + c.n_levels);
    f.mvLevelSigma2.resize(c.n_levels);
    f.mvInvLevelSigma2.resize(c.n_levels);
    for (int l = 0; l < c.n_levels; l++) {  // ORBextractor.cc:522-533
        f.mvLevelSigma2[l] = c.scale_factors[l] * c.scale_factors[l];
        f.mvInvLevelSigma2[l] = 1.0f / f.mvLevelSigma2[l];
    }
    f.mfLogScaleFactor = c.n_levels > 1 ? std::log(c.scale_factors[1]) : 1.f;
    for (int j = 0; j < k->n_nodes; j++) {
        std::vector<unsigned int> &v = f.mFeatVec[k->node_ids[j]];
        for (int e = k->node_off[j]; e < k->node_off[j + 1]; e++) v.push_back((unsigned int)k->node_idx[e]);
    }
    f.mvpMapPoints.assign(n, static_cast<MapPoint *>(NULL));
}

void fill_keyframe(KeyFrame &kf, const amos_test_kf *k, std::vector<MapPoint> &pts)
{
    fill_base(kf, k);
    kf.Tcw = cv::Mat(4, 4, CV_32F);
    std::memcpy(kf.Tcw.data, k->cam.Tcw, sizeof(float) * 16);
    const amos_adapt::V3 ow = amos_adapt::centre(amos_adapt::pose_of(kf.Tcw));  // KeyFrame::SetPose: Ow = -Rwc * tcw (one gemm)
    kf.Ow = cv::Mat(3, 1, CV_32F);
    kf.Ow.at<float>(0, 0) = ow.x; kf.Ow.at<float>(1, 0) = ow.y; kf.Ow.at<float>(2, 0) = ow.z;
    for (int i = 0; i < k->n; i++)
        if (k->point_of && k->point_of[i] >= 0) {
            MapPoint *p = &pts[k->point_of[i]];
            kf.mvpMapPoints[i] = p;
            if (!p->mObservations.count(&kf)) p->mObservations[&kf] = i;  // the point table's `obs` is Observations(); this is who observes
        }
}

void fill_frame2(Frame &f, const amos_test_kf *k, std::vector<MapPoint> &pts)
{
    fill_base(f, k);
    f.mvbOutlier.assign(k->n, false);
    f.mTcw = cv::Mat(4, 4, CV_32F);
    std::memcpy(f.mTcw.data, k->cam.Tcw, sizeof(float) * 16);
    for (int i = 0; i < k->n; i++)
        if (k->point_of && k->point_of[i] >= 0) f.mvpMapPoints[i] = &pts[k->point_of[i]];
}

inline int32_t index_of(const MapPoint *p, const std::vector<MapPoint> &pts) { return p ? (int32_t)(p - pts.data()) : -1; }
}  // namespace

extern "C" {

// MapPoint::PredictScale of the stand-in (MapPoint.cc: ceil(log(max / dist) / log scale factor), clamped), for the tests' own pre-filter
int amos_host_standin_predict_scale(float max_dist, float cur_dist, float scale_factor, int n_levels)
{
    MapPoint p;
    p.mfMaxDistance = max_dist;
    FrameBase f;
    f.mfLogScaleFactor = n_levels > 1 ? std::log(scale_factor) : 1.f;  // as fill_base derives it from scale_factors[1]
    f.mnScaleLevels = n_levels;
    return p.PredictScale(cur_dist, &f);
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist), ORBmatcher.cc:1731.
// cur->point_of = the occupants of CurrentFrame.mvpMapPoints on entry; cur_points[i2] = CurrentFrame.mvpMapPoints[i2] on return.
int amos_host_ref_search_reloc(const amos_test_kf *cur, const amos_test_kf *kf, const amos_test_points *points, const uint8_t *already_found,
                               float th, int orb_dist, float nnratio, int check_orientation, int32_t *cur_points)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    Frame Cur;
    KeyFrame KF;
    fill_frame2(Cur, cur, pts);
    fill_keyframe(KF, kf, pts);
    std::set<MapPoint *> found;
    for (int i = 0; i < points->n; i++)
        if (already_found && already_found[i]) found.insert(&pts[i]);
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByProjection(Cur, &KF, found, th, orb_dist);
    for (int i = 0; i < cur->n; i++) cur_points[i] = index_of(Cur.mvpMapPoints[i], pts);
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th), :388.
// vp: indices into the point table; matched: in / out, n_kf entries (point index or -1).
int amos_host_ref_search_kf_scw(const amos_test_kf *kf, const amos_test_points *points, const float *Scw, const int32_t *vp, int n_vp, int32_t *matched,
                                int th, float nnratio)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame KF;
    fill_keyframe(KF, kf, pts);
    cv::Mat S(4, 4, CV_32F);
    std::memcpy(S.data, Scw, sizeof(float) * 16);
    std::vector<MapPoint *> vpPoints(n_vp), vpMatched(kf->n, static_cast<MapPoint *>(NULL));
    for (int i = 0; i < n_vp; i++) vpPoints[i] = &pts[vp[i]];
    for (int i = 0; i < kf->n; i++)
        if (matched[i] >= 0) vpMatched[i] = &pts[matched[i]];
    RefMatcher matcher(nnratio, true);
    const int n = matcher.SearchByProjection(&KF, S, vpPoints, vpMatched, th);
    for (int i = 0; i < kf->n; i++) matched[i] = index_of(vpMatched[i], pts);
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), :230.  matches: F.N entries.
int amos_host_ref_search_bow_kf_frame(const amos_test_kf *kf, const amos_test_kf *frame, const amos_test_points *points, float nnratio,
                                      int check_orientation, int32_t *matches)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame KF;
    Frame F;
    fill_keyframe(KF, kf, pts);
    fill_frame2(F, frame, pts);
    std::vector<MapPoint *> vpMapPointMatches(3, &pts[0]);  // the function re-creates it with F.N NULLs (:234)
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByBoW(&KF, F, vpMapPointMatches);
    if ((int)vpMapPointMatches.size() != frame->n) return -104;
    for (int i = 0; i < frame->n; i++) matches[i] = index_of(vpMapPointMatches[i], pts);
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), :656.  matches12: pKF1->N entries.
int amos_host_ref_search_bow_kf_kf(const amos_test_kf *kf1, const amos_test_kf *kf2, const amos_test_points *points, float nnratio,
                                   int check_orientation, int32_t *matches12)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame K1, K2;
    fill_keyframe(K1, kf1, pts);
    fill_keyframe(K2, kf2, pts);
    std::vector<MapPoint *> vpMatches12;
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByBoW(&K1, &K2, vpMatches12);
    if ((int)vpMatches12.size() != kf1->n) return -104;
    for (int i = 0; i < kf1->n; i++) matches12[i] = index_of(vpMatches12[i], pts);
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize), :515
int amos_host_ref_search_initialization(const amos_test_kf *f1, const amos_test_kf *f2, float *prev_matched, int32_t *matches12, int window_size,
                                        float nnratio, int check_orientation)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> none;
    Frame F1, F2;
    fill_frame2(F1, f1, none);
    fill_frame2(F2, f2, none);
    std::vector<cv::Point2f> prev(f1->n);
    for (int i = 0; i < f1->n; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
    std::vector<int> m12;
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchForInitialization(F1, F2, prev, m12, window_size);
    for (int i = 0; i < f1->n; i++) {
        matches12[i] = m12[i];
        prev_matched[2 * i] = prev[i].x;
        prev_matched[2 * i + 1] = prev[i].y;
    }
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, bOnlyStereo), :810
int amos_host_ref_search_triangulation(const amos_test_kf *kf1, const amos_test_kf *kf2, const amos_test_points *points, const float *F12,
                                       int only_stereo, float nnratio, int check_orientation, int32_t *pairs, int cap)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame K1, K2;
    fill_keyframe(K1, kf1, pts);
    fill_keyframe(K2, kf2, pts);
    cv::Mat F(3, 3, CV_32F);
    std::memcpy(F.data, F12, sizeof(float) * 9);
    std::vector<std::pair<size_t, size_t> > vMatchedPairs;
    RefMatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchForTriangulation(&K1, &K2, F, vMatchedPairs, only_stereo != 0);
    if ((int)vMatchedPairs.size() > cap) return -3;
    for (size_t i = 0; i < vMatchedPairs.size(); i++) {
        pairs[2 * i] = (int32_t)vMatchedPairs[i].first;
        pairs[2 * i + 1] = (int32_t)vMatchedPairs[i].second;
    }
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
// const cv::Mat &t12, const float th), :1314.  matches12: in / out, pKF1->N entries (point index or -1).
int amos_host_ref_search_sim3(const amos_test_kf *kf1, const amos_test_kf *kf2, const amos_test_points *points, int32_t *matches12, float s12,
                              const float *R12, const float *t12, float th, float nnratio)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame K1, K2;
    fill_keyframe(K1, kf1, pts);
    fill_keyframe(K2, kf2, pts);
    cv::Mat R(3, 3, CV_32F), t(3, 1, CV_32F);
    std::memcpy(R.data, R12, sizeof(float) * 9);
    std::memcpy(t.data, t12, sizeof(float) * 3);
    std::vector<MapPoint *> vpMatches12(kf1->n, static_cast<MapPoint *>(NULL));
    for (int i = 0; i < kf1->n; i++)
        if (matches12[i] >= 0) vpMatches12[i] = &pts[matches12[i]];
    RefMatcher matcher(nnratio, true);
    const int n = matcher.SearchBySim3(&K1, &K2, vpMatches12, s12, R, t, th);
    for (int i = 0; i < kf1->n; i++) matches12[i] = index_of(vpMatches12[i], pts);
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), :1020.  vp: point indices, -1 = a NULL entry.
// Out: the keyframe's map points afterwards, and per table entry replaced_by (mpReplaced or -1), Observations() and isBad().
int amos_host_ref_fuse(const amos_test_kf *kf, const amos_test_points *points, const int32_t *vp, int n_vp, float th, float nnratio,
                       int32_t *kf_points, int32_t *replaced_by, int32_t *obs_after, uint8_t *bad_after)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame KF;
    fill_keyframe(KF, kf, pts);
    std::vector<MapPoint *> vpMapPoints(n_vp);
    for (int i = 0; i < n_vp; i++) vpMapPoints[i] = vp[i] >= 0 ? &pts[vp[i]] : static_cast<MapPoint *>(NULL);
    RefMatcher matcher(nnratio, true);
    const int n = matcher.Fuse(&KF, vpMapPoints, th);
    for (int i = 0; i < kf->n; i++) kf_points[i] = index_of(KF.mvpMapPoints[i], pts);
    for (int i = 0; i < points->n; i++) {
        replaced_by[i] = index_of(pts[i].mpReplaced, pts);
        obs_after[i] = pts[i].Observations();
        bad_after[i] = pts[i].isBad();
    }
    return n;
    AMOS_HOST_CATCH
}

// ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint), :1179.
// replace_point: in / out, n_vp entries (point index or -1).
int amos_host_ref_fuse_scw(const amos_test_kf *kf, const amos_test_points *points, const float *Scw, const int32_t *vp, int n_vp, float th, float nnratio,
                           int32_t *replace_point, int32_t *kf_points, int32_t *obs_after)
{
    AMOS_HOST_TRY
    std::vector<MapPoint> pts = make_points(points);
    KeyFrame KF;
    fill_keyframe(KF, kf, pts);
    cv::Mat S(4, 4, CV_32F);
    std::memcpy(S.data, Scw, sizeof(float) * 16);
    std::vector<MapPoint *> vpPoints(n_vp), vpReplace(n_vp, static_cast<MapPoint *>(NULL));
    for (int i = 0; i < n_vp; i++) {
        vpPoints[i] = &pts[vp[i]];
        if (replace_point[i] >= 0) vpReplace[i] = &pts[replace_point[i]];
    }
    RefMatcher matcher(nnratio, true);
    const int n = matcher.Fuse(&KF, S, vpPoints, th, vpReplace);
    for (int i = 0; i < n_vp; i++) replace_point[i] = index_of(vpReplace[i], pts);
    for (int i = 0; i < kf->n; i++) kf_points[i] = index_of(KF.mvpMapPoints[i], pts);
    for (int i = 0; i < points->n; i++) obs_after[i] = pts[i].Observations();
    return n;
    AMOS_HOST_CATCH
}

// ---- SURVEY 8b threading: "distinct ORBextractor instances are used concurrently (stereo: two std::threads, Frame.cc:165-170); ORBmatcher
// is re-entrant and used from the Tracking, LocalMapping and LoopClosing threads simultaneously".  n_ext extractor threads (each owns one
// ORBextractor for its frame, as Tracking owns mpORBextractorLeft / Right, and calls the 4-arg operator() `iters` times) run beside n_match
// matcher threads (each constructs an ORBmatcher ON THE STACK per iteration, as every reference call site does, and runs
// SearchByProjection(CurrentFrame, LastFrame) on its own inputs).  Every iteration must reproduce the first one's result; the caller holds
// the results to the oracle.

struct amos_thread_extract_job {
    const uint8_t *gray;
    int32_t w, h;
    amos_keypoint *kps;   // cap
    uint8_t *desc;        // cap x 32
    int32_t cap, n, rc;   // rc: 0 ok, -1 an iteration differed from the first, -100 exception
};

struct amos_thread_match_job {
    const amos_frame_view *cur;
    const amos_proj_query *q;
    int32_t nq;
    const int32_t *cur_match_in;  // cur->n
    int32_t *cur_match_out;       // cur->n
    const float *scale_factors;
    int32_t nsf;
    float mbf, th;
    int32_t forward, backward, result, rc;
};

int amos_host_run_threads(amos_thread_extract_job *ext, int n_ext, amos_thread_match_job *mat, int n_match, int iters, int *pool_handles)
{
    AMOS_HOST_TRY
    std::vector<std::thread> threads;
    std::vector<std::string> errors(n_ext + n_match);
    for (int t = 0; t < n_ext; t++)
        threads.emplace_back([&, t] {
            amos_thread_extract_job &j = ext[t];
            try {
                ORBextractor extractor(1000, 1.2f, 8, 20, 7);
                j.rc = 0;
                std::vector<cv::KeyPoint> first;
                cv::Mat firstDesc;
                for (int it = 0; it < iters; it++) {
                    cv::Mat image(j.h, j.w, CV_8UC1, (void *)j.gray, (size_t)j.w), mask, descriptors;
                    std::vector<cv::KeyPoint> keys;
                    extractor(image, mask, keys, descriptors);
                    if (it == 0) {
                        first = keys;
                        firstDesc = descriptors.clone();
                    } else if (keys.size() != first.size() || (keys.size() && (std::memcmp(keys.data(), first.data(), sizeof(cv::KeyPoint) * keys.size()) != 0 ||
                                                                               std::memcmp(descriptors.data, firstDesc.data, 32 * keys.size()) != 0))) {
                        j.rc = -1;
                    }
                }
                j.n = (int)first.size();
                if (j.n > j.cap) { j.rc = -3; return; }
                if (j.n) {
                    std::memcpy(j.kps, first.data(), sizeof(amos_keypoint) * first.size());
                    std::memcpy(j.desc, firstDesc.data, (size_t)32 * first.size());
                }
            } catch (const std::exception &e) {
                errors[t] = e.what();
                j.rc = -100;
            }
        });
    for (int t = 0; t < n_match; t++)
        threads.emplace_back([&, t] {
            amos_thread_match_job &j = mat[t];
            try {
                FeatureGrid grid(*j.cur);
                std::vector<amos_proj_query> pts(j.q, j.q + j.nq);
                std::vector<float> sf(j.scale_factors, j.scale_factors + j.nsf);
                j.rc = 0;
                for (int it = 0; it < iters; it++) {
                    ORBmatcher matcher(0.9f, true);  // Tracking.cc:1910
                    std::vector<int> match(j.cur_match_in, j.cur_match_in + j.cur->n);
                    const int r = matcher.SearchByProjection(grid, pts, match, sf, j.mbf, j.th, j.forward != 0, j.backward != 0);
                    if (it == 0) {
                        j.result = r;
                        std::memcpy(j.cur_match_out, match.data(), sizeof(int) * j.cur->n);
                    } else if (r != j.result || std::memcmp(j.cur_match_out, match.data(), sizeof(int) * j.cur->n) != 0) {
                        j.rc = -1;
                    }
                }
            } catch (const std::exception &e) {
                errors[n_ext + t] = e.what();
                j.rc = -100;
            }
        });
    for (std::thread &th : threads) th.join();
    if (pool_handles) *pool_handles = ORBmatcher::PoolHandlesCreated();
    for (const std::string &e : errors)
        if (!e.empty()) { g_host_error = e; return -100; }
    return 0;
    AMOS_HOST_CATCH
}

// ---- what Tracking.cc:366 + Frame.cc:480-496,633 + Tracking.cc:1910 execute per frame, through the C++ classes themselves, host buffers
// in and out:  yolact::evalImage(bgr) -> ORBextractor::operator()(gray, Mat(), levels) -> MovingKeyPoints(mask) -> ProcessDesp ->
// a STACK-constructed ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, LastFrame, 15, false) on stand-in frames (the last frame's
// features carry map points back-projected at 2 m; identity poses: the projection window sits on the feature's own pixel).
// frames: n_frames gray (h x w) and, with py_file, BGR (h x w x 3) frames, used round-robin.  out_ms[6]: mean per frame of
// {evalImage, 3-arg operator(), MovingKeyPoints, ProcessDesp, ORBmatcher ctor + SearchByProjection + dtor, whole frame};
// out_counts[3]: keypoints and matches of the last frame, evalImage calls that returned false.
int amos_host_frame_latency(const char *py_file, const char *weights, const uint8_t *bgr, const uint8_t *gray, int n_frames, int w, int h,
                            int warm, int iters, int pyramid_mode /* -1: the class's default, else ORBextractor::PyramidMode */, double *out_ms,
                            int32_t *out_counts)
{
    AMOS_HOST_TRY
    using namespace amos_standins;
    typedef std::chrono::steady_clock clk;
    yolact *seg = nullptr;
    if (py_file && *py_file) {
        seg = new yolact(py_file, weights ? weights : "", 20);  // (kept alive: the Python side keeps its session, as System.cc:106 keeps its own)
        if (!seg->isInitializedResult()) { g_host_error = seg->getErrorDescriptionString(); return -102; }
    }
    ORBextractor ext(1000, 1.2f, 8, 20, 7);  // Tracking.cc:172
    if (pyramid_mode >= 0) ext.SetPyramidMode((ORBextractor::PyramidMode)pyramid_mode);
    std::vector<float> sf = ext.GetScaleFactors();
    const float fx = 535.4f, fy = 539.2f, cx = 320.1f, cy = 247.6f;  // TUM3.yaml
    auto make_frame = [&](Frame &F, const std::vector<cv::KeyPoint> &keys, const cv::Mat &desc) {
        const int n = (int)keys.size();
        F.N = n;
        F.mvKeys = keys;
        F.mvKeysUn = keys;
        F.mDescriptors = desc.empty() ? cv::Mat(1, 32, CV_8U) : desc;
        F.mvuRight.assign(n, -1.f);
        F.mvpMapPoints.assign(n, static_cast<MapPoint *>(NULL));
        F.mvbOutlier.assign(n, false);
        F.fx = fx; F.fy = fy; F.cx = cx; F.cy = cy; F.mb = 0.08f; F.mbf = 40.f;
        F.mnMinX = 0.f; F.mnMaxX = (float)w; F.mnMinY = 0.f; F.mnMaxY = (float)h;
        F.mTcw = cv::Mat::zeros(4, 4, CV_32F);
        for (int i = 0; i < 4; i++) F.mTcw.at<float>(i, i) = 1.f;
        F.mnScaleLevels = (int)sf.size();
        F.mvScaleFactors = sf;
        F.mfLogScaleFactor = std::log(sf.size() > 1 ? sf[1] : 1.2f);
    };
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int failed_masks = 0, last_kp = 0, last_matches = 0;
    Frame Last;
    std::vector<MapPoint> lastPoints;
    bool haveLast = false;
    for (int it = 0; it < warm + iters; it++) {
        const int f = it % n_frames;
        const bool timed = it >= warm;
        cv::Mat imGray(h, w, CV_8UC1, (void *)(gray + (size_t)f * w * h), (size_t)w), none;
        const clk::time_point t0 = clk::now();
        cv::Mat SegMask = cv::Mat::zeros(h, w, CV_8UC1);  // Tracking.cc:305
        if (seg) {
            cv::Mat imRGB(h, w * 3, CV_8UC1, (void *)(bgr + (size_t)f * w * h * 3), (size_t)w * 3), m;
            if (seg->evalImage(imRGB, m)) SegMask = m;
            else failed_masks += timed;
        }
        const clk::time_point t1 = clk::now();
        std::vector<std::vector<cv::KeyPoint> > mvKeysTemp;
        ext(imGray, none, mvKeysTemp);  // Frame.cc:484
        const clk::time_point t2 = clk::now();
        std::vector<cv::KeyPoint> dyn = ext.MovingKeyPoints(imGray, SegMask, cv::Mat(), std::vector<center>(), std::vector<int>(), std::vector<bool>(), mvKeysTemp);  // Frame.cc:633
        const clk::time_point t3 = clk::now();
        std::vector<cv::KeyPoint> mvKeys;
        cv::Mat mDescriptors;
        ext.ProcessDesp(imGray, none, mvKeysTemp, mvKeys, mDescriptors);  // Frame.cc:496
        const clk::time_point t4 = clk::now();
        Frame Cur;
        make_frame(Cur, mvKeys, mDescriptors);
        int nmatches = 0;
        clk::time_point t5 = t4, t6 = t4;
        if (haveLast) {
            t5 = clk::now();
            RefMatcher matcher(0.9f, true);  // Tracking.cc:1910: on the stack, every frame
            nmatches = matcher.SearchByProjection(Cur, Last, 15.f, false);  // Tracking.cc:1937
            t6 = clk::now();
        }
        // the current frame becomes the last one: every feature gets a map point 2 m in front of its pixel
        lastPoints.clear();
        lastPoints.resize(mvKeys.size());  // (each default-constructed: a copied stand-in MapPoint would share its Mats)
        for (size_t i = 0; i < mvKeys.size(); i++) {
            MapPoint &p = lastPoints[i];
            const float z = 2.f;
            p.mWorldPos.at<float>(0, 0) = (mvKeys[i].pt.x - cx) * z / fx;
            p.mWorldPos.at<float>(1, 0) = (mvKeys[i].pt.y - cy) * z / fy;
            p.mWorldPos.at<float>(2, 0) = z;
            std::memcpy(p.mDescriptor.data, mDescriptors.ptr((int)i), 32);
            p.mnObs = 1;
        }
        make_frame(Last, mvKeys, mDescriptors.clone());
        for (size_t i = 0; i < mvKeys.size(); i++) Last.mvpMapPoints[i] = &lastPoints[i];
        haveLast = true;
        const clk::time_point t7 = clk::now();
        if (timed) {
            auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            acc[0] += ms(t0, t1); acc[1] += ms(t1, t2); acc[2] += ms(t2, t3); acc[3] += ms(t3, t4); acc[4] += ms(t5, t6);
            acc[5] += ms(t0, t4) + ms(t5, t6);  // (building the stand-in frames is the harness's work, not the path's)
            (void)t7;
        }
        last_kp = (int)mvKeys.size();
        last_matches = nmatches;
    }
    for (int k = 0; k < 6; k++) out_ms[k] = acc[k] / std::max(iters, 1);
    out_counts[0] = last_kp;
    out_counts[1] = last_matches;
    out_counts[2] = failed_masks;
    return 0;
    AMOS_HOST_CATCH
}

int amos_host_yolact_eval(const char *py_file, const char *weights, const uint8_t *bgr, int w, int h, uint8_t *mask_out, int *mask_w,
                          int *mask_h)
{
    AMOS_HOST_TRY
    static yolact *seg = nullptr;
    static std::string key;
    const std::string want = std::string(py_file) + "|" + weights;
    if (!seg || key != want) {
        delete seg;
        seg = new yolact(py_file, weights, 20);
        key = want;
    }
    if (!seg->isInitializedResult()) { g_host_error = seg->getErrorDescriptionString(); return -102; }
    cv::Mat frame(h, w * 3, CV_8UC1, (void *)bgr, (size_t)w * 3), mask;
    if (!seg->evalImage(frame, mask)) { g_host_error = seg->getErrorDescriptionString(); return -103; }
    *mask_w = mask.cols;
    *mask_h = mask.rows;
    for (int y = 0; y < mask.rows; y++) std::memcpy(mask_out + (size_t)y * mask.cols, mask.ptr(y), mask.cols);
    return 0;
    AMOS_HOST_CATCH
}

}  // extern "C"
