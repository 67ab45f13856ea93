// Compile check (g++ -fsyntax-only, tests/test_host_adaptors_cpu.py): with AMOS_REFERENCE_TREE defined the class named
// ORB_SLAM2::ORBmatcher is the reference-signature matcher, and the call sites of the reference's Tracking.cc,
// LocalMapping.cc and LoopClosing.cc -- written here in their original form -- compile against it.  Frame / KeyFrame /
// MapPoint are the stand-ins, placed in namespace ORB_SLAM2 as the reference's classes are.
#define AMOS_REFERENCE_TREE
#define AMOS_STANDIN_NS ORB_SLAM2
#include "ref_standins.h"
#include "../../amos-slam_amd/host/ORBmatcher_adaptors.h"

namespace ORB_SLAM2
{
int call_sites(Frame &mCurrentFrame, Frame &mLastFrame, Frame &mInitialFrame, KeyFrame *mpReferenceKF, KeyFrame *pKF2, std::vector<MapPoint *> &vpMapPoints,
               std::set<MapPoint *> &sFound, cv::Mat Scw, cv::Mat F12, cv::Mat R12, cv::Mat t12, std::vector<cv::Point2f> &mvbPrevMatched,
               std::vector<int> &mvIniMatches)
{
    int n = 0;
    ORBmatcher matcher(0.9, true);
    n += matcher.SearchByProjection(mCurrentFrame, mLastFrame, 15, false);                    // Tracking.cc:1928
    n += matcher.SearchByProjection(mCurrentFrame, vpMapPoints, 3);                           // Tracking.cc:2379
    n += matcher.SearchByProjection(mCurrentFrame, mpReferenceKF, sFound, 10, 100);           // Tracking.cc:2644
    std::vector<MapPoint *> vpMapPointMatches;
    n += matcher.SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches);                // Tracking.cc:1757
    n += matcher.SearchForInitialization(mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100);  // Tracking.cc:1346
    std::vector<std::pair<size_t, size_t> > vMatchedIndices;
    n += matcher.SearchForTriangulation(mpReferenceKF, pKF2, F12, vMatchedIndices, false);    // LocalMapping.cc:344
    n += matcher.Fuse(mpReferenceKF, vpMapPoints);                                            // LocalMapping.cc:663
    std::vector<MapPoint *> vpMatches12, vpReplacePoints(vpMapPoints.size(), static_cast<MapPoint *>(NULL));
    n += matcher.SearchByBoW(mpReferenceKF, pKF2, vpMatches12);                               // LoopClosing.cc:308
    n += matcher.SearchBySim3(mpReferenceKF, pKF2, vpMatches12, 1.0f, R12, t12, 7.5);         // LoopClosing.cc:390
    n += matcher.SearchByProjection(mpReferenceKF, Scw, vpMapPoints, vpMatches12, 10);        // LoopClosing.cc:469
    n += matcher.Fuse(mpReferenceKF, Scw, vpMapPoints, 4, vpReplacePoints);                   // LoopClosing.cc:796
    const cv::Mat d1 = vpMapPoints[0]->GetDescriptor(), d2 = vpMapPoints[1]->GetDescriptor();
    n += ORBmatcher::DescriptorDistance(d1, d2);                                              // MapPoint.cc:396
    n += ORBmatcher::TH_LOW + ORBmatcher::TH_HIGH + ORBmatcher::HISTO_LENGTH;
    return n;
}
}  // namespace ORB_SLAM2
