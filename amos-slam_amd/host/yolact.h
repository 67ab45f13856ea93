// yolact.h -- drop-in for the reference's include/yolact.h:57-101 (class ORB_SLAM2::yolact): the C++
// object Tracking calls once per frame for the dynamic-object mask.  Same constructor, evalImage,
// status accessors and (optional) worker-thread members; it embeds CPython and calls
// amos-slam_amd/mask/yolact_interface.py, whose network runs on PyTorch-ROCm.
//
// Differences that a caller can observe: none in the API.  Internally the frame is handed to Python
// as raw BGR bytes and the reference's marshalling (cv::resize to 480x640, /255, CHW; yolact.cc:220,
// 385-451) happens on the GPU (mask/pre.py) instead of on the CPU; the NumPy C API is not needed.
#ifndef AMOS_YOLACT_H
#define AMOS_YOLACT_H

#include <cstddef>
#include <mutex>
#include <string>

#include "amos_cv.h"

#define EVAL_PY_FUNCTION_NAME "yolact_eval"
#define INIT_PY_FUNCTION_NAME "yolact_init"

namespace ORB_SLAM2
{

class Tracking;  // consumer, not part of this library

class yolact
{
public:
    yolact(const std::string &pyFilePath, const std::string &modelPath, const size_t &categories);
    ~yolact();

    // yolact.cc:203-318: false (and confidenceImage untouched) on any Python-side failure, including
    // "nothing detected above the score threshold"
    bool evalImage(const cv::Mat &inputImage, cv::Mat &confidenceImage);

    // worker-thread interface of the reference (yolact.cc:123-201; disabled there, System.cc:139)
    void Run();
    void RequestFinish() { mbFinishRequested = true; }
    void SetTracker(Tracking *pTracker) { mpTracker = pTracker; }
    bool isNewImgArrived();
    void ProduceImgSegment();
    Tracking *mpTracker;
    std::mutex mMutexGetNewImg;
    std::mutex mMutexNewImgSegment;
    bool mbNewImgFlag;
    int mSkipIndex;
    int imgIndex;
    cv::Mat mImg;
    cv::Mat mMask;

    inline bool isInitializedResult(void) const { return mbIsPythonInitializedOK & mbIsLEDNETInitializedOK; }
    inline const std::string &getErrorDescriptionString(void) const { return mstrErrDescription; }
    inline size_t getCLassNum(void) const { return mnCategories; }

private:
    bool parseFilePathAndName(const std::string &strFilePathAndName);
    void FetchPythonError(const std::string &context);

    std::string mstrPyMoudlePath;
    std::string mstrPyMoudleName;
    void *mpPyEvalModule;  // PyObject*
    void *mpPyEvalFunc;    // PyObject*: yolact_eval_bgr_bytes (the frame as bytes: engines that do not run on a GPU)
    // the per-frame session of mask/yolact_interface.py (yolact_frame_session / yolact_eval_session): pinned host buffers the frame is
    // written to and the mask read from, the pass between them replayed as one HIP graph
    bool evalThroughSession(const cv::Mat &inputImage, cv::Mat &confidenceImage, bool &handled);
    void *mpPySessionEval;  // PyObject*: yolact_eval_session
    unsigned char *mpSessionFrame, *mpSessionMask;
    int mnSessionH, mnSessionW, mnSessionMaskRows, mnSessionMaskCols;
    bool mbSessionUnavailable;
    size_t mnCategories;
    bool mbIsLEDNETInitializedOK;
    bool mbIsPythonInitializedOK;
    bool mbOwnsInterpreter;
    volatile bool mbFinishRequested;
    std::string mstrErrDescription;
};

}  // namespace ORB_SLAM2

#endif
