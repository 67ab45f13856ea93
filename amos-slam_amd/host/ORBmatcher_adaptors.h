// ORBmatcher_adaptors.h -- the ten ORBmatcher searches with the REFERENCE's signatures
// (include/ORBmatcher.h:57-215: Frame&, KeyFrame*, MapPoint*), implemented over the view-based
// searches of ORBmatcher.h.  Each adaptor does what the reference function does before and after
// its candidate loop -- the geometric pre-filter of every map point (projection, image bounds,
// distance invariance, viewing angle, predicted level: the file:line is cited per function) and
// the write-back (mvpMapPoints / vpMatched / Replace / AddObservation) -- and hands the candidate
// loops themselves (the GPU-backed part) to the base class.
//
// Templated on the three classes so that it compiles (a) inside the reference tree, against the
// reference's own Frame.h / KeyFrame.h / MapPoint.h: define AMOS_REFERENCE_TREE before including
// and the class `ORBmatcher` below IS the drop-in (Tracking.cc, LocalMapping.cc, LoopClosing.cc
// compile unchanged); and (b) here, against the minimal stand-ins of
// tests/host/ref_standins.h that carry exactly the members these functions read.
//
// Matrix arithmetic: the reference writes `Rcw*x3Dw+tcw` etc. on CV_32F cv::Mat; OpenCV evaluates
// such an expression as ONE gemm with double accumulation and a single rounding to float
// (parity unpinned like every OpenCV-derived step).  The helpers below do that.
#ifndef ORBMATCHER_ADAPTORS_H
#define ORBMATCHER_ADAPTORS_H

#include <cmath>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

#include "ORBmatcher.h"

namespace ORB_SLAM2
{
namespace amos_adapt
{
struct V3 {
    float x, y, z;
};
struct Pose {  // rotation (row-major) and translation of a world -> camera transform
    float R[9];
    float t[3];
};

inline V3 vec3(const cv::Mat &m) { return V3{m.at<float>(0, 0), m.at<float>(1, 0), m.at<float>(2, 0)}; }
inline Pose pose_of(const cv::Mat &R, const cv::Mat &t)
{
    Pose p;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) p.R[3 * r + c] = R.at<float>(r, c);
        p.t[r] = t.at<float>(r, 0);
    }
    return p;
}
inline Pose pose_of(const cv::Mat &T)  // 4x4 (or 3x4) [R | t]
{
    Pose p;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) p.R[3 * r + c] = T.at<float>(r, c);
        p.t[r] = T.at<float>(r, 3);
    }
    return p;
}
// R * x + t as one gemm: double accumulation, one rounding
inline V3 transform(const Pose &p, const V3 &x)
{
    float o[3];
    for (int r = 0; r < 3; r++)
        o[r] = (float)((double)p.R[3 * r] * x.x + (double)p.R[3 * r + 1] * x.y + (double)p.R[3 * r + 2] * x.z + (double)p.t[r]);
    return V3{o[0], o[1], o[2]};
}
// -R^T * t (camera centre in world coordinates)
inline V3 centre(const Pose &p)
{
    float o[3];
    for (int c = 0; c < 3; c++) o[c] = (float)(-((double)p.R[c] * p.t[0] + (double)p.R[3 + c] * p.t[1] + (double)p.R[6 + c] * p.t[2]));
    return V3{o[0], o[1], o[2]};
}
inline V3 sub(const V3 &a, const V3 &b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const V3 &a, const V3 &b) { return (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z; }  // Mat::dot
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }                                                // cv::norm
// Scw -> (Rcw, tcw) with the scale divided out (ORBmatcher.cc:397-400, 1188-1191): Mat / float = Mat * (1 / s)
inline Pose unscaled(const cv::Mat &Scw)
{
    Pose p = pose_of(Scw);
    const float scw = (float)std::sqrt((double)p.R[0] * p.R[0] + (double)p.R[1] * p.R[1] + (double)p.R[2] * p.R[2]);
    const float inv = (float)(1.0 / (double)scw);
    for (float &v : p.R) v *= inv;
    for (float &v : p.t) v *= inv;
    return p;
}

template <class DescMat>
inline void copy_desc(uint8_t *dst, const DescMat &d) { std::memcpy(dst, d.template ptr<unsigned char>(0), 32); }

// the members a search reads of the frame / keyframe it searches IN
template <class FrameLike>
inline amos_frame_view view_of(const FrameLike &f)
{
    amos_frame_view v;
    v.n = f.N;
    v.keys_un = reinterpret_cast<const amos_keypoint *>(f.mvKeysUn.data());  // cv::KeyPoint has amos_keypoint's layout
    v.descriptors = f.mDescriptors.template ptr<unsigned char>(0);           // N x 32, continuous (ORBextractor allocates it so)
    v.u_right = f.mvuRight.empty() ? nullptr : f.mvuRight.data();
    v.min_x = f.mnMinX;
    v.max_x = f.mnMaxX;
    v.min_y = f.mnMinY;
    v.max_y = f.mnMaxY;
    return v;
}

// DBoW2::FeatureVector (std::map<node id, std::vector<unsigned>>) as the flat arrays of amos_bow_view
struct FlatFeatVec {
    std::vector<uint32_t> ids;
    std::vector<int32_t> off, idx;
    template <class FV>
    explicit FlatFeatVec(const FV &fv)
    {
        off.push_back(0);
        for (typename FV::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            ids.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
            off.push_back((int32_t)idx.size());
        }
    }
};
template <class FrameLike>
inline amos_bow_view bow_view_of(const FrameLike &f, const std::vector<cv::KeyPoint> &keys, const FlatFeatVec &fv, const std::vector<uint8_t> *has_point,
                                 bool with_right)
{
    amos_bow_view v;
    v.n = f.N;
    v.keys = reinterpret_cast<const amos_keypoint *>(keys.data());
    v.descriptors = f.mDescriptors.template ptr<unsigned char>(0);
    v.has_point = has_point ? has_point->data() : nullptr;
    v.u_right = with_right && !f.mvuRight.empty() ? f.mvuRight.data() : nullptr;
    v.n_nodes = (int32_t)fv.ids.size();
    v.node_ids = fv.ids.data();
    v.node_off = fv.off.data();
    v.node_idx = fv.idx.data();
    return v;
}
}  // namespace amos_adapt

template <class FrameT, class KeyFrameT, class MapPointT>
class ORBmatcherFor : public AMOS_VIEW_MATCHER
{
public:
    ORBmatcherFor(float nnratio = 0.6, bool checkOri = true) : AMOS_VIEW_MATCHER(nnratio, checkOri) {}

    // the view-based forms stay callable
    using AMOS_VIEW_MATCHER::Fuse;
    using AMOS_VIEW_MATCHER::SearchByBoW;
    using AMOS_VIEW_MATCHER::SearchByProjection;
    using AMOS_VIEW_MATCHER::SearchBySim3;
    using AMOS_VIEW_MATCHER::SearchForInitialization;
    using AMOS_VIEW_MATCHER::SearchForTriangulation;

    // ORBmatcher.cc:70-175 (Tracking::SearchLocalPoints)
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3)
    {
        std::vector<amos_map_query> q;
        std::vector<MapPointT *> src;
        for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
            MapPointT *pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView || pMP->isBad()) continue;  // :79-83
            amos_map_query e;
            e.proj_x = pMP->mTrackProjX;
            e.proj_y = pMP->mTrackProjY;
            e.proj_xr = pMP->mTrackProjXR;
            e.view_cos = pMP->mTrackViewCos;
            e.level = pMP->mnTrackScaleLevel;
            e.has_obs = pMP->Observations() > 0;
            amos_adapt::copy_desc(e.desc, pMP->GetDescriptor());
            q.push_back(e);
            src.push_back(pMP);
        }
        std::vector<int> vnCurMatch(F.N, -1);
        std::vector<bool> vbCurHasObs(F.N, false);
        for (int i = 0; i < F.N; i++) vbCurHasObs[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;  // :121-123
        const FeatureGrid grid(amos_adapt::view_of(F));
        const int n = SearchByProjection(grid, q, vnCurMatch, vbCurHasObs, F.mvScaleFactors, th);
        for (int i = 0; i < F.N; i++)
            if (vnCurMatch[i] >= 0) F.mvpMapPoints[i] = src[vnCurMatch[i]];  // :168
        return n;
    }

    // ORBmatcher.cc:1569-1728 (Tracking::TrackWithMotionModel)
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
    {
        using namespace amos_adapt;
        const Pose cw = pose_of(CurrentFrame.mTcw), lw = pose_of(LastFrame.mTcw);
        const V3 twc = centre(cw);
        const V3 tlc = transform(lw, twc);                                   // Rlw * twc + tlw, :1596
        const bool bForward = tlc.z > CurrentFrame.mb && !bMono;             // :1598-1599
        const bool bBackward = -tlc.z > CurrentFrame.mb && !bMono;
        std::vector<amos_proj_query> q;
        std::vector<MapPointT *> src;
        for (int i = 0; i < LastFrame.N; i++) {
            MapPointT *pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;                   // :1604-1608
            const V3 x3Dc = transform(cw, vec3(pMP->GetWorldPos()));
            const float xc = x3Dc.x, yc = x3Dc.y;
            const float invzc = 1.0 / x3Dc.z;
            if (invzc < 0) continue;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            amos_proj_query e;
            e.u = u;
            e.v = v;
            e.invz = invzc;
            e.octave = LastFrame.mvKeys[i].octave;
            e.angle = LastFrame.mvKeysUn[i].angle;
            e.has_obs = pMP->Observations() > 0;
            copy_desc(e.desc, pMP->GetDescriptor());
            q.push_back(e);
            src.push_back(pMP);
        }
        // :1658-1660 tests the CURRENT frame's occupant.  Tracking clears mvpMapPoints before this search and a point
        // assigned during the search is one of the queries; occupants that were there on entry join the query list as
        // unsearchable entries (a projection far outside the grid has no candidate): they only answer "has observations".
        std::vector<int> vnCurMatch(CurrentFrame.N, -1);
        for (int i2 = 0; i2 < CurrentFrame.N; i2++)
            if (CurrentFrame.mvpMapPoints[i2]) {
                amos_proj_query e;
                std::memset(&e, 0, sizeof(e));
                e.u = e.v = -1.0e9f;
                e.has_obs = CurrentFrame.mvpMapPoints[i2]->Observations() > 0;
                vnCurMatch[i2] = (int)q.size();
                q.push_back(e);
                src.push_back(CurrentFrame.mvpMapPoints[i2]);
            }
        const FeatureGrid grid(view_of(CurrentFrame));
        const int n = SearchByProjection(grid, q, vnCurMatch, CurrentFrame.mvScaleFactors, CurrentFrame.mbf, th, bForward, bBackward);
        for (int i2 = 0; i2 < CurrentFrame.N; i2++) CurrentFrame.mvpMapPoints[i2] = vnCurMatch[i2] >= 0 ? src[vnCurMatch[i2]] : static_cast<MapPointT *>(NULL);
        return n;
    }

    // ORBmatcher.cc:1731-1863 (Tracking::Relocalization)
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound, const float th, const int ORBdist)
    {
        using namespace amos_adapt;
        const Pose cw = pose_of(CurrentFrame.mTcw);
        const V3 Ow = centre(cw);
        const std::vector<MapPointT *> vpMPs = pKF->GetMapPointMatches();
        std::vector<amos_kf_query> q;
        std::vector<MapPointT *> src;
        for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
            MapPointT *pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;  // :1753-1757
            const V3 x3Dw = vec3(pMP->GetWorldPos());
            const V3 x3Dc = transform(cw, x3Dw);
            const float xc = x3Dc.x, yc = x3Dc.y;
            const float invzc = 1.0 / x3Dc.z;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const float dist3D = norm(sub(x3Dw, Ow));
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            amos_kf_query e;
            e.u = u;
            e.v = v;
            e.level = pMP->PredictScale(dist3D, &CurrentFrame);
            e.angle = pKF->mvKeysUn[i].angle;
            copy_desc(e.desc, pMP->GetDescriptor());
            q.push_back(e);
            src.push_back(pMP);
        }
        std::vector<int> vnCurMatch(CurrentFrame.N, AMOS_MATCH_FREE);
        for (int i2 = 0; i2 < CurrentFrame.N; i2++)
            if (CurrentFrame.mvpMapPoints[i2]) vnCurMatch[i2] = AMOS_MATCH_TAKEN;  // :1816-1817
        const FeatureGrid grid(view_of(CurrentFrame));
        const int n = SearchByProjection(grid, q, vnCurMatch, CurrentFrame.mvScaleFactors, th, ORBdist);
        for (int i2 = 0; i2 < CurrentFrame.N; i2++)
            if (vnCurMatch[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = src[vnCurMatch[i2]];
        return n;
    }

    // ORBmatcher.cc:388-512 (LoopClosing::ComputeSim3)
    int SearchByProjection(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, std::vector<MapPointT *> &vpMatched, int th)
    {
        std::set<MapPointT *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT *>(NULL));
        std::vector<amos_window_query> q;
        std::vector<MapPointT *> src;
        KeyFrameQueries(pKF, amos_adapt::unscaled(Scw), vpPoints, spAlreadyFound, /*skipInKF=*/false, /*withRight=*/false, q, src, nullptr);
        std::vector<int> vnMatched(pKF->N, AMOS_MATCH_FREE);
        for (size_t i = 0; i < vpMatched.size() && i < (size_t)pKF->N; i++)
            if (vpMatched[i]) vnMatched[i] = AMOS_MATCH_TAKEN;  // :480-481
        const FeatureGrid grid(amos_adapt::view_of(*pKF));
        const int n = SearchByProjection(grid, q, vnMatched, pKF->mvScaleFactors, th);
        for (int i = 0; i < pKF->N; i++)
            if (vnMatched[i] >= 0) vpMatched[i] = src[vnMatched[i]];  // :500
        return n;
    }

    // ORBmatcher.cc:230-382 (Tracking::TrackReferenceKeyFrame, Relocalization)
    int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches)
    {
        const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT *>(F.N, static_cast<MapPointT *>(NULL));
        std::vector<uint8_t> has(pKF->N, 0);
        for (int i = 0; i < pKF->N; i++) has[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :268-274
        const amos_adapt::FlatFeatVec fvKF(pKF->mFeatVec), fvF(F.mFeatVec);
        const amos_bow_view vKF = amos_adapt::bow_view_of(*pKF, pKF->mvKeysUn, fvKF, &has, false);
        const amos_bow_view vF = amos_adapt::bow_view_of(F, F.mvKeys, fvF, nullptr, false);  // :318 reads F.mvKeys[bestIdxF].angle
        std::vector<int> vnMatchesF;
        const int n = SearchByBoW(vKF, vF, vnMatchesF);
        for (int iF = 0; iF < F.N; iF++)
            if (vnMatchesF[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[vnMatchesF[iF]];
        return n;
    }

    // ORBmatcher.cc:656-808 (LoopClosing::ComputeSim3)
    int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12)
    {
        const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPointT *>(vpMapPoints1.size(), static_cast<MapPointT *>(NULL));
        std::vector<uint8_t> has1(pKF1->N, 0), has2(pKF2->N, 0);
        for (int i = 0; i < pKF1->N; i++) has1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
        for (int i = 0; i < pKF2->N; i++) has2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
        const amos_adapt::FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
        const amos_bow_view v1 = amos_adapt::bow_view_of(*pKF1, pKF1->mvKeysUn, fv1, &has1, false);
        const amos_bow_view v2 = amos_adapt::bow_view_of(*pKF2, pKF2->mvKeysUn, fv2, &has2, false);
        std::vector<int> vnMatches12;
        const int n = SearchByBoW(v1, v2, vnMatches12, true);
        for (int i1 = 0; i1 < pKF1->N; i1++)
            if (vnMatches12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[vnMatches12[i1]];
        return n;
    }

    // ORBmatcher.cc:515-643 (Tracking::MonocularInitialization)
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        const FeatureGrid grid2(amos_adapt::view_of(F2));
        return SearchForInitialization(amos_adapt::view_of(F1), grid2, vbPrevMatched, vnMatches12, windowSize);
    }

    // ORBmatcher.cc:810-1018 (LocalMapping::CreateNewMapPoints)
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
    {
        using namespace amos_adapt;
        // the epipole of pKF1's camera centre in pKF2, :818-826
        const V3 C2 = transform(pose_of(pKF2->GetRotation(), pKF2->GetTranslation()), vec3(pKF1->GetCameraCenter()));
        const float invz = 1.0f / C2.z;
        const float ex = pKF2->fx * C2.x * invz + pKF2->cx;
        const float ey = pKF2->fy * C2.y * invz + pKF2->cy;
        std::vector<uint8_t> has1(pKF1->N, 0), has2(pKF2->N, 0);
        for (int i = 0; i < pKF1->N; i++) has1[i] = pKF1->GetMapPoint(i) != NULL;  // :846-849
        for (int i = 0; i < pKF2->N; i++) has2[i] = pKF2->GetMapPoint(i) != NULL;  // :874-877
        const FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
        const amos_bow_view v1 = bow_view_of(*pKF1, pKF1->mvKeysUn, fv1, &has1, true);
        const amos_bow_view v2 = bow_view_of(*pKF2, pKF2->mvKeysUn, fv2, &has2, true);
        float f[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) f[3 * r + c] = F12.at<float>(r, c);
        return SearchForTriangulation(v1, v2, f, ex, ey, pKF2->mvScaleFactors, pKF2->mvLevelSigma2, vMatchedPairs, bOnlyStereo);
    }

    // ORBmatcher.cc:1314-1565 (LoopClosing::ComputeSim3)
    int SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                     const float th)
    {
        using namespace amos_adapt;
        const Pose w1 = pose_of(pKF1->GetRotation(), pKF1->GetTranslation()), w2 = pose_of(pKF2->GetRotation(), pKF2->GetTranslation());
        Pose c12, c21;  // camera 2 -> camera 1 and back: sR12 | t12, sR21 | t21 (:1331-1335)
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) {
                c12.R[3 * r + c] = (float)((double)s12 * R12.at<float>(r, c));
                c21.R[3 * r + c] = (float)((1.0 / s12) * R12.at<float>(c, r));
            }
            c12.t[r] = t12.at<float>(r, 0);
        }
        for (int r = 0; r < 3; r++)
            c21.t[r] = (float)(-((double)c21.R[3 * r] * c12.t[0] + (double)c21.R[3 * r + 1] * c12.t[1] + (double)c21.R[3 * r + 2] * c12.t[2]));
        const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {  // :1345-1355
            MapPointT *pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one direction: the points of `from` through (its world pose, the Sim3 hop) into `into`
        auto direction = [&](const std::vector<MapPointT *> &pts, const std::vector<bool> &done, const Pose &wFrom, const Pose &hop, KeyFrameT *into,
                             std::vector<amos_window_query> &out) {
            for (int i = 0; i < (int)pts.size(); i++) {
                MapPointT *pMP = pts[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                const V3 pc = transform(hop, transform(wFrom, vec3(pMP->GetWorldPos())));
                if (pc.z < 0.0) continue;
                const float invz = 1.0 / pc.z;
                const float x = pc.x * invz, y = pc.y * invz;
                const float u = pKF1->fx * x + pKF1->cx, v = pKF1->fy * y + pKF1->cy;  // the reference uses pKF1's intrinsics both ways (:1316-1319)
                if (!into->IsInImage(u, v)) continue;
                const float dist3D = norm(pc);
                if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
                amos_window_query e;
                e.u = u;
                e.v = v;
                e.ur = 0.f;
                e.level = pMP->PredictScale(dist3D, into);
                e.src = i;
                copy_desc(e.desc, pMP->GetDescriptor());
                out.push_back(e);
            }
        };
        std::vector<amos_window_query> v1in2, v2in1;
        direction(vpMapPoints1, vbAlreadyMatched1, w1, c21, pKF2, v1in2);
        direction(vpMapPoints2, vbAlreadyMatched2, w2, c12, pKF1, v2in1);
        const FeatureGrid g1(view_of(*pKF1)), g2(view_of(*pKF2));
        std::vector<int> vnMatches12;
        const int n = SearchBySim3(g1, g2, v1in2, v2in1, pKF1->mvScaleFactors, pKF2->mvScaleFactors, vnMatches12, th);
        for (int i1 = 0; i1 < N1; i1++)
            if (vnMatches12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[vnMatches12[i1]];
        return n;
    }

    // ORBmatcher.cc:1020-1177 (LocalMapping::SearchInNeighbors)
    int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th = 3.0)
    {
        std::vector<amos_window_query> q;
        std::vector<MapPointT *> src;
        const amos_adapt::Pose cw = amos_adapt::pose_of(pKF->GetRotation(), pKF->GetTranslation());
        const amos_adapt::V3 Ow = amos_adapt::vec3(pKF->GetCameraCenter());
        KeyFrameQueries(pKF, cw, vpMapPoints, std::set<MapPointT *>(), /*skipInKF=*/true, /*withRight=*/true, q, src, &Ow);
        const FeatureGrid grid(amos_adapt::view_of(*pKF));
        std::vector<int> vnBestIdx;
        Fuse(grid, q, pKF->mvScaleFactors, pKF->mvInvLevelSigma2, th, vnBestIdx);
        // The reference filters, searches and writes back point by point (:1038-1172); here all searches come first.  The search of a
        // point reads only the keyframe's features, which the write-backs do not touch, so the order matters through two tests alone:
        // a point made bad by an earlier Replace of this call, or added to this keyframe by an earlier AddObservation (the same
        // point twice in vpMapPoints), is skipped by the reference's filter (:1046, :1052) when its turn comes -- re-tested here.
        // (A Replace also recomputes the surviving point's descriptor; the survivor is then in this keyframe, i.e. filtered, so the
        // descriptor copied before the searches is never one the reference would have searched with after the change.)
        int nFused = 0;
        for (size_t i = 0; i < q.size(); i++) {  // :1146-1172, in the reference's order
            const int bestIdx = vnBestIdx[i];
            if (bestIdx < 0) continue;
            if (src[i]->isBad() || src[i]->IsInKeyFrame(pKF)) continue;
            MapPointT *pMP = src[i], *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations())
                        pMP->Replace(pMPinKF);
                    else
                        pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // ORBmatcher.cc:1179-1312 (LoopClosing::SearchAndFuse)
    int Fuse(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint)
    {
        const std::set<MapPointT *> spAlreadyFound = pKF->GetMapPoints();
        std::vector<amos_window_query> q;
        std::vector<MapPointT *> src;
        std::vector<int> at;  // position of each query in vpPoints
        KeyFrameQueries(pKF, amos_adapt::unscaled(Scw), vpPoints, spAlreadyFound, /*skipInKF=*/false, /*withRight=*/false, q, src, nullptr, &at);
        const FeatureGrid grid(amos_adapt::view_of(*pKF));
        std::vector<int> vnBestIdx;
        Fuse(grid, q, pKF->mvScaleFactors, th, vnBestIdx);
        int nFused = 0;
        for (size_t i = 0; i < q.size(); i++) {  // :1291-1308 (spAlreadyFound is taken once on entry, :1196, so only isBad can change under way)
            const int bestIdx = vnBestIdx[i];
            if (bestIdx < 0) continue;
            if (src[i]->isBad()) continue;
            MapPointT *pMP = src[i], *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[at[i]] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

protected:
    // The pre-filter the four keyframe-side searches share (Fuse :1033-1080, :1199-1244; SearchByProjection(pKF, Scw)
    // :410-455): projection, IsInImage, distance invariance, viewing angle under 60 degrees, predicted level.
    void KeyFrameQueries(KeyFrameT *pKF, const amos_adapt::Pose &cw, const std::vector<MapPointT *> &pts, const std::set<MapPointT *> &skip,
                         bool skipInKF, bool withRight, std::vector<amos_window_query> &q, std::vector<MapPointT *> &src, const amos_adapt::V3 *pOw,
                         std::vector<int> *at = nullptr)
    {
        using namespace amos_adapt;
        const V3 Ow = pOw ? *pOw : centre(cw);
        for (int i = 0, n = (int)pts.size(); i < n; i++) {
            MapPointT *pMP = pts[i];
            if (!pMP) continue;  // (only Fuse(pKF, vpMapPoints) tests this; the other callers never pass NULL)
            if (pMP->isBad() || skip.count(pMP)) continue;
            if (skipInKF && pMP->IsInKeyFrame(pKF)) continue;
            const V3 p3Dw = vec3(pMP->GetWorldPos());
            const V3 p3Dc = transform(cw, p3Dw);
            if (p3Dc.z < 0.0f) continue;
            const float invz = 1 / p3Dc.z;
            const float x = p3Dc.x * invz, y = p3Dc.y * invz;
            const float u = pKF->fx * x + pKF->cx, v = pKF->fy * y + pKF->cy;
            if (!pKF->IsInImage(u, v)) continue;
            const V3 PO = sub(p3Dw, Ow);
            const float dist3D = norm(PO);
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            if (dot(PO, vec3(pMP->GetNormal())) < 0.5 * dist3D) continue;
            amos_window_query e;
            e.u = u;
            e.v = v;
            e.ur = withRight ? u - pKF->mbf * invz : 0.f;
            e.level = pMP->PredictScale(dist3D, pKF);
            e.src = i;
            copy_desc(e.desc, pMP->GetDescriptor());
            q.push_back(e);
            src.push_back(pMP);
            if (at) at->push_back(i);
        }
    }
};

#ifdef AMOS_REFERENCE_TREE
// Inside the reference tree: Frame.h, KeyFrame.h and MapPoint.h are included before this header (the reference's
// include/ORBmatcher.h does so at its lines 32-34), so the three classes are complete here.
class ORBmatcher : public ORBmatcherFor<Frame, KeyFrame, MapPoint>
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : ORBmatcherFor<Frame, KeyFrame, MapPoint>(nnratio, checkOri) {}
};
#endif

}  // namespace ORB_SLAM2

#endif
