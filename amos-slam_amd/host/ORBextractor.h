// ORBextractor.h -- drop-in for the reference's include/ORBextractor.h:93-168 (class
// ORB_SLAM2::ORBextractor): same constructor, call operators, MovingKeyPoints, ProcessDesp, getters
// and the public mvImagePyramid member, implemented over the C ABI of include/amos_frontend.h
// (every method launches the HIP kernels of libamos_frontend.so; nothing is computed on the host).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#include "amos_cv.h"

struct amos_orb;

namespace ORB_SLAM2
{

#ifndef AMOS_HAVE_CLUSTER_H
// include/cluster.h:22-31 of the reference; only `id` is read by MovingKeyPoints.
struct center {
    int x, y, L, A, B, D, label, id;
};
#endif

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // ORBextractor.cc:1544 -- keypoints (level-0 coordinates) + descriptors; mask is ignored
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);
    // ORBextractor.cc:1672 -- per-level keypoints only (level coordinates)
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<std::vector<cv::KeyPoint>> &_keypoints);
    // ORBextractor.cc:1747
    void ProcessDesp(cv::InputArray image, cv::InputArray mask, std::vector<std::vector<cv::KeyPoint>> &_allKeypoints,
                     std::vector<cv::KeyPoint> &_mKeypoints, cv::OutputArray descriptors);
    // ORBextractor.cc:1688
    std::vector<cv::KeyPoint> MovingKeyPoints(const cv::Mat &imGray, const cv::Mat &imS, const cv::Mat &imLS,
                                              std::vector<center> centers, std::vector<int> rm_vector, std::vector<bool> DynaFlag,
                                              std::vector<std::vector<cv::KeyPoint>> &mvKeysT);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // ROIs into padded buffers, as in the reference.  Pixel data is copied from the device after every
    // extraction unless SetPyramidDownload(false): the RGB-D flow only reads mvImagePyramid[0].rows.
    std::vector<cv::Mat> mvImagePyramid;
    void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }
    // The device handle (batch API, streams): see include/amos_frontend.h.
    amos_orb *Handle() { return mpHandle; }

protected:
    void EnsureHandle(int width, int height);
    void Detect(const cv::Mat &image);
    void FetchLevels(std::vector<std::vector<cv::KeyPoint>> &levels);
    void StoreLevels(const std::vector<std::vector<cv::KeyPoint>> &levels);
    void UpdatePyramid(int width, int height);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    amos_orb *mpHandle;
    int mnHandleW, mnHandleH;
    bool mbDownloadPyramid;
    std::vector<int> mvLevelOffset, mvLevelCap;
    int mnLevelTotal;
};

}  // namespace ORB_SLAM2

#endif
