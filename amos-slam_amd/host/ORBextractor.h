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

    // ROIs into padded buffers, as in the reference; rows / cols / step are always those of the level.  The PIXELS live on the device:
    // by default (PYRAMID_AUTO) the 4-arg operator() -- the mono / stereo Frame constructors, whose ComputeStereoMatches reads
    // mvImagePyramid pixels (Frame.cc:1401,1434) -- copies every level back, the 3-arg operator() of the RGB-D Amos flow, which
    // only reads mvImagePyramid[0].rows (Frame.cc:1197), does not (1.16 MB per frame saved).  SetPyramidDownload(true / false) forces
    // either for both; DownloadPyramid() fills the Mats of the last extraction on demand.
    std::vector<cv::Mat> mvImagePyramid;
    enum PyramidMode { PYRAMID_AUTO = 0, PYRAMID_ALWAYS = 1, PYRAMID_NEVER = 2 };
    void SetPyramidDownload(bool on) { mnPyramidMode = on ? PYRAMID_ALWAYS : PYRAMID_NEVER; }
    void SetPyramidMode(PyramidMode m) { mnPyramidMode = m; }
    void DownloadPyramid();
    // The HIP device the handle lives on: by default the calling thread's current device at the first extraction
    // (amos_current_device), or the one AMOS_DEVICE names; SetDevice before the first extraction overrides both.
    void SetDevice(int device);
    int GetDevice() const { return mnDevice; }
    // The device handle (batch API, streams): see include/amos_frontend.h.
    amos_orb *Handle() { return mpHandle; }

protected:
    void EnsureHandle(int width, int height);
    void Detect(const cv::Mat &image);
    void FetchLevels(std::vector<std::vector<cv::KeyPoint>> &levels);
    void StoreLevels(const std::vector<std::vector<cv::KeyPoint>> &levels);
    void UpdatePyramid(int width, int height, bool download);

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
    int mnDevice;  // -1 until chosen
    int mnHandleW, mnHandleH;
    int mnPyramidMode;
    std::vector<cv::Mat> mvPyramidStore;  // the padded buffers mvImagePyramid's ROIs point into, kept from frame to frame
    bool mbPyramidOnHost;                 // the Mats hold the last extraction's pixels
    std::vector<int> mvLevelOffset, mvLevelCap;
    int mnLevelTotal;
    // what the device-resident per-level lists hold (as last fetched or stored): a caller handing back unchanged vectors
    // (MovingKeyPoints straight after operator(), ProcessDesp straight after MovingKeyPoints) costs no transfer
    std::vector<int32_t> mvDeviceCounts;
    std::vector<cv::KeyPoint> mvDeviceLists;
    bool mbDeviceListsKnown;
    std::vector<cv::KeyPoint> mvStage;  // marshalling scratch (level lists; keypoints of the final result)
    std::vector<uint8_t> mvStageDesc;
};

}  // namespace ORB_SLAM2

#endif
