// ORBmatcher.h -- host side of the matcher: ORB_SLAM2::ORBmatcher with the reference's constants,
// constructor, DescriptorDistance and ComputeThreeMaxima (include/ORBmatcher.h:57-215), and the
// gated searches of the tracking thread restated over plain frame views (include/amos_host_types.h).
//
// Structure of every search: the host enumerates each query's candidates exactly as the reference
// does (Frame::GetFeaturesInArea order), ONE call computes all candidate distances on the GPU
// (amos_match_list_distances), and the reference's sequential greedy loop -- "already matched"
// skips, right-coordinate gate, best / second best, thresholds, rotation histogram -- then runs on
// the host over those distances in the reference's order, so tie-breaks are identical.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <cstddef>
#include <utility>
#include <vector>

#include "../../include/amos_host_types.h"
#include "amos_cv.h"

struct amos_match;

namespace ORB_SLAM2
{

// Frame::mGrid + AssignFeaturesToGrid + PosInGrid + GetFeaturesInArea (Frame.cc:431-461, 894-1030)
class FeatureGrid
{
public:
    explicit FeatureGrid(const amos_frame_view &frame);
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                          const int maxLevel = -1) const;
    const amos_frame_view &Frame() const { return mFrame; }

private:
    bool PosInGrid(const amos_keypoint &kp, int &posX, int &posY) const;
    amos_frame_view mFrame;
    float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<size_t> mGrid[AMOS_FRAME_GRID_COLS][AMOS_FRAME_GRID_ROWS];
};

// Inside the reference tree (AMOS_REFERENCE_TREE) the name ORBmatcher belongs to the class of ORBmatcher_adaptors.h,
// which carries the reference's signatures and derives from this one.
#ifdef AMOS_REFERENCE_TREE
#define AMOS_VIEW_MATCHER ORBmatcherViews
#else
#define AMOS_VIEW_MATCHER ORBmatcher
#endif

class AMOS_VIEW_MATCHER
{
public:
    AMOS_VIEW_MATCHER(float nnratio = 0.6, bool checkOri = true);
    ~AMOS_VIEW_MATCHER();
    AMOS_VIEW_MATCHER(const AMOS_VIEW_MATCHER &) = delete;
    AMOS_VIEW_MATCHER &operator=(const AMOS_VIEW_MATCHER &) = delete;

    // ORBmatcher.cc:1913: one pair.  (Batches go through DescriptorDistances; a single pair still
    // runs on the device so that there is one implementation of the distance.)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    // all pairs: out[i * nt + j]
    void DescriptorDistances(const uint8_t *q, int nq, const uint8_t *t, int nt, std::vector<uint16_t> &out);

    // ORBmatcher.cc:1569-1728, SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono).
    //   vnCurMatch[i2]  : -1, or the index of the query whose map point sits in CurrentFrame.mvpMapPoints[i2]
    //   mvScaleFactors  : CurrentFrame.mvScaleFactors;  mbf: CurrentFrame.mbf
    //   bForward / bBackward as the reference derives them from tlc (:1598-1599)
    int SearchByProjection(const FeatureGrid &CurrentFrame, const std::vector<amos_proj_query> &vLastPoints, std::vector<int> &vnCurMatch,
                           const std::vector<float> &mvScaleFactors, float mbf, const float th, const bool bForward, const bool bBackward);

    // ORBmatcher.cc:70-175, SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th).
    //   vbCurHasObs[i]: F.mvpMapPoints[i] && Observations() > 0 on entry; updated as points are assigned
    //   vnCurMatch[i]  : receives the index of the map query assigned to feature i (or keeps its value)
    int SearchByProjection(const FeatureGrid &F, const std::vector<amos_map_query> &vpMapPoints, std::vector<int> &vnCurMatch,
                           std::vector<bool> &vbCurHasObs, const std::vector<float> &mvScaleFactors, const float th = 3);

    // ORBmatcher.cc:1731-1863, SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)
    // (relocalisation).  vnCurMatch[i2]: AMOS_MATCH_FREE, AMOS_MATCH_TAKEN (CurrentFrame.mvpMapPoints[i2] set on
    // entry) or, on return, the index of the query assigned to feature i2.
    int SearchByProjection(const FeatureGrid &CurrentFrame, const std::vector<amos_kf_query> &vKFPoints, std::vector<int> &vnCurMatch,
                           const std::vector<float> &mvScaleFactors, const float th, const int ORBdist);

    // ORBmatcher.cc:230-382, SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches) (TrackReferenceKeyFrame,
    // Relocalization).  vnMatchesF[iF] = index of the keyframe feature whose map point F's feature iF receives, or -1.
    int SearchByBoW(const amos_bow_view &KF, const amos_bow_view &F, std::vector<int> &vnMatchesF);

    // ORBmatcher.cc:656-808, SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (loop closing).
    // vnMatches12[idx1] = the feature of KF2 whose map point KF1's feature idx1 is matched to, or -1.
    int SearchByBoW(const amos_bow_view &KF1, const amos_bow_view &KF2, std::vector<int> &vnMatches12, const bool bBothKeyFrames);

    // ORBmatcher.cc:810-1018, SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (local mapping).
    //   F12: row-major 3x3;  ex, ey: the epipole in KF2 (:818-826);  mvScaleFactors2 / mvLevelSigma2_2: pKF2's tables
    int SearchForTriangulation(const amos_bow_view &KF1, const amos_bow_view &KF2, const float F12[9], float ex, float ey,
                               const std::vector<float> &mvScaleFactors2, const std::vector<float> &mvLevelSigma2_2,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);
    // ORBmatcher.cc:188-215 (takes the three numbers it reads of pKF2 / F12 instead of the objects)
    static bool CheckDistEpipolarLine(const amos_keypoint &kp1, const amos_keypoint &kp2, const float F12[9], float sigma2_kp2);

    // ORBmatcher.cc:1020-1177, Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th): the search of every map
    // point (window th * scale, levels nPredictedLevel-1 .. nPredictedLevel, chi2 gate 5.99 / 7.8 on the reprojection
    // error, best <= TH_LOW).  vnBestIdx[q] = feature of pKF the point fuses with or -1; the MapPoint bookkeeping of
    // :1150-1172 (Replace / AddObservation) is the caller's.  KF.Frame().u_right = pKF->mvuRight.
    int Fuse(const FeatureGrid &KF, const std::vector<amos_window_query> &vpMapPoints, const std::vector<float> &mvScaleFactors,
             const std::vector<float> &mvInvLevelSigma2, const float th, std::vector<int> &vnBestIdx);
    // ORBmatcher.cc:1179-1312, Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (loop closing): no chi2 gate.
    int Fuse(const FeatureGrid &KF, const std::vector<amos_window_query> &vpPoints, const std::vector<float> &mvScaleFactors, const float th,
             std::vector<int> &vnBestIdx);
    // ORBmatcher.cc:388-512, SearchByProjection(pKF, Scw, vpPoints, vpMatched, th).  vnMatched[idx]: AMOS_MATCH_FREE,
    // AMOS_MATCH_TAKEN (vpMatched[idx] set on entry) or, on return, the index of the query assigned to feature idx.
    int SearchByProjection(const FeatureGrid &KF, const std::vector<amos_window_query> &vpPoints, std::vector<int> &vnMatched,
                           const std::vector<float> &mvScaleFactors, const int th);
    // ORBmatcher.cc:1314-1565, SearchBySim3.  v1in2: the map points of KF1 (src = i1) projected into KF2 with S21, v2in1 the
    // other way (src = i2), both already filtered as :1361-1393 / :1441-1473 do.  vnMatches12[i1] = i2 where both
    // directions agree, else -1.
    int SearchBySim3(const FeatureGrid &KF1, const FeatureGrid &KF2, const std::vector<amos_window_query> &v1in2,
                     const std::vector<amos_window_query> &v2in1, const std::vector<float> &mvScaleFactors1,
                     const std::vector<float> &mvScaleFactors2, std::vector<int> &vnMatches12, const float th);

    // ORBmatcher.cc:515-643, SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
    int SearchForInitialization(const amos_frame_view &F1, const FeatureGrid &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                std::vector<int> &vnMatches12, int windowSize = 10);

    // ORBmatcher.cc:1866-1908 (protected in the reference; public here for the tests)
    static void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    float RadiusByViewingCos(const float &viewCos);
    // candidates of every query in pKF->GetFeaturesInArea(u, v, th * scale) order that pass the level gate (and the chi2
    // gate when mvInvLevelSigma2 is given), with their distances (one GPU call)
    void WindowCandidates(const FeatureGrid &KF, const std::vector<amos_window_query> &q, const std::vector<float> &mvScaleFactors,
                          const float th, const std::vector<float> *mvInvLevelSigma2, std::vector<int> &off, std::vector<int> &idx,
                          std::vector<uint16_t> &dist);
    void ListDistances(const amos_frame_view &train, const uint8_t *queries, int nq, const std::vector<int> &off, const std::vector<int> &idx,
                       std::vector<uint16_t> &dist);
    void ListDistances(const uint8_t *train, int nt, const uint8_t *queries, int nq, const std::vector<int> &off, const std::vector<int> &idx,
                       std::vector<uint16_t> &dist);

    // the device handle of this object: borrowed from the process-wide pool on first use (ORBmatcher.cc), returned by the destructor
    amos_match *Handle();

    float mfNNratio;
    bool mbCheckOrientation;
    amos_match *mpMatch;
    int mnDevice;

public:
    // handles the pool has created since the process started (tests: stack-constructed matchers reuse them)
    static int PoolHandlesCreated();
};

}  // namespace ORB_SLAM2

#endif
