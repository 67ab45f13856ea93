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

class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    ~ORBmatcher();
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

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
    void ListDistances(const amos_frame_view &train, const uint8_t *queries, int nq, const std::vector<int> &off, const std::vector<int> &idx,
                       std::vector<uint16_t> &dist);
    void ListDistances(const uint8_t *train, int nt, const uint8_t *queries, int nq, const std::vector<int> &off, const std::vector<int> &idx,
                       std::vector<uint16_t> &dist);

    float mfNNratio;
    bool mbCheckOrientation;
    amos_match *mpMatch;
};

}  // namespace ORB_SLAM2

#endif
