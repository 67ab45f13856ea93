// ref_standins.h -- minimal Frame / KeyFrame / MapPoint with exactly the members the ten reference-signature
// searches of amos-slam_amd/host/ORBmatcher_adaptors.h read (names and types as in the reference's include/Frame.h,
// KeyFrame.h, MapPoint.h), so that the adaptors compile and run here without the rest of ORB-SLAM2.  Test harness.
#pragma once
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "../../amos-slam_amd/host/amos_cv.h"

#ifndef AMOS_STANDIN_NS
#define AMOS_STANDIN_NS amos_standins
#endif

namespace AMOS_STANDIN_NS
{
class KeyFrame;
class Frame;
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;  // DBoW2::FeatureVector

class MapPoint
{
public:
    MapPoint() : mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(1.f), mTrackProjX(0), mTrackProjY(0), mTrackProjXR(0), mWorldPos(3, 1, CV_32F),
                 mNormal(3, 1, CV_32F), mDescriptor(1, 32, CV_8U), mnObs(0), mbBad(false), mfMinDistance(0.f), mfMaxDistance(1e9f), mpReplaced(nullptr)
    {
    }
    bool isBad() { return mbBad; }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormal.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return mnObs; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    template <class F> int PredictScale(const float &currentDist, F *pF)  // MapPoint.cc: the level whose scale matches the distance ratio
    {
        const float ratio = mfMaxDistance / currentDist;
        int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
        return nScale;
    }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    inline void AddObservation(KeyFrame *pKF, size_t idx);  // MapPoint.cc: a stereo observation counts twice
    inline void Replace(MapPoint *pMP);                     // MapPoint.cc:244-310: observations move to pMP, the keyframes are updated

    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos, mTrackProjX, mTrackProjY, mTrackProjXR;
    cv::Mat mWorldPos, mNormal, mDescriptor;
    int mnObs;
    bool mbBad;
    float mfMinDistance, mfMaxDistance;
    MapPoint *mpReplaced;
    std::map<KeyFrame *, size_t> mObservations;
};

struct FrameBase {
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
};

class Frame : public FrameBase
{
public:
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
};

class KeyFrame : public FrameBase
{
public:
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints()
    {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints)
            if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = static_cast<MapPoint *>(NULL); }
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }
    cv::Mat GetRotation() { return cv::Mat(Tcw, cv::Rect(0, 0, 3, 3)).clone(); }
    cv::Mat GetTranslation() { return cv::Mat(Tcw, cv::Rect(3, 0, 1, 3)).clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    std::vector<MapPoint *> mvpMapPoints;
    cv::Mat Tcw, Ow;
};

inline void MapPoint::AddObservation(KeyFrame *pKF, size_t idx)
{
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    mnObs += (idx < pKF->mvuRight.size() && pKF->mvuRight[idx] >= 0) ? 2 : 1;
}

inline void MapPoint::Replace(MapPoint *pMP)
{
    if (pMP == this) return;
    const std::map<KeyFrame *, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (std::map<KeyFrame *, size_t>::const_iterator mit = obs.begin(); mit != obs.end(); ++mit) {
        KeyFrame *pKF = mit->first;
        if (!pMP->IsInKeyFrame(pKF)) {
            pKF->ReplaceMapPointMatch(mit->second, pMP);
            pMP->AddObservation(pKF, mit->second);
        } else {
            pKF->EraseMapPointMatch(mit->second);
        }
    }
}
}  // namespace AMOS_STANDIN_NS
