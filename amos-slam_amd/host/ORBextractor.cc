// ORBextractor.cc -- host side of the drop-in ORB_SLAM2::ORBextractor: argument marshalling between
// cv::Mat / std::vector<cv::KeyPoint> and the C ABI.  cv::KeyPoint and amos_keypoint share one layout
// (tests/test_constants.py), so keypoint vectors move with memcpy.
#include "ORBextractor.h"

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/amos_frontend.h"

namespace ORB_SLAM2
{

static_assert(sizeof(cv::KeyPoint) == sizeof(amos_keypoint), "cv::KeyPoint layout");

static void Check(int rc, const char *what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + amos_last_error());
}

static inline amos_keypoint *AsAmos(cv::KeyPoint *p) { return reinterpret_cast<amos_keypoint *>(p); }
static inline const amos_keypoint *AsAmos(const cv::KeyPoint *p) { return reinterpret_cast<const amos_keypoint *>(p); }

// AMOS_DEVICE, else the calling thread's current HIP device: a one-process-per-GPU host that selected its GPU
// (hipSetDevice / torch.cuda.set_device / HIP_VISIBLE_DEVICES) gets its objects there
int AmosPickDevice()
{
    const char *e = std::getenv("AMOS_DEVICE");
    if (e && *e) return std::atoi(e);
    const int d = amos_current_device();
    Check(d, "amos_current_device");
    return d;
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mpHandle(nullptr),
      mnDevice(-1), mnHandleW(0), mnHandleH(0), mnPyramidMode(PYRAMID_AUTO), mbPyramidOnHost(false), mnLevelTotal(0), mbDeviceListsKnown(false)
{
    mvImagePyramid.resize(nlevels);
    mvPyramidStore.resize(nlevels);
    // The tables (ORBextractor.cc:500-608) come from the library so both sides agree bit for bit; host arithmetic there
    // as here: constructing an extractor touches no device (Tracking.cc:172-185 constructs two or three of them).
    amos_orb_params p = {nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST};
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    Check(amos_orb_tables_host(&p, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                               mnFeaturesPerLevel.data(), umax.data()), "amos_orb_tables_host");
}

ORBextractor::~ORBextractor()
{
    if (mpHandle) amos_orb_destroy(mpHandle);
}

void ORBextractor::SetDevice(int device)
{
    if (mpHandle && device != mnDevice) throw std::runtime_error("ORBextractor::SetDevice after the first extraction");
    mnDevice = device;
}

void ORBextractor::EnsureHandle(int width, int height)
{
    if (mpHandle && width <= mnHandleW && height <= mnHandleH) return;
    if (mpHandle) amos_orb_destroy(mpHandle);
    mpHandle = nullptr;
    if (mnDevice < 0) mnDevice = AmosPickDevice();
    amos_orb_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    mnHandleW = std::max(width, mnHandleW);
    mnHandleH = std::max(height, mnHandleH);
    Check(amos_orb_create(&p, mnHandleW, mnHandleH, 1, mnDevice, nullptr, &mpHandle), "amos_orb_create");
    mbDeviceListsKnown = false;
}

// mvImagePyramid[l] = temp(wholeSize)(ROI), ORBextractor.cc:1835-1838.  The padded buffers are kept from frame to frame
// (the reference rewrites mvImagePyramid on every call too); pixels only when asked for.
void ORBextractor::UpdatePyramid(int width, int height, bool download)
{
    std::vector<int32_t> lw(nlevels), lh(nlevels);
    amos_orb_level_sizes(mpHandle, width, height, lw.data(), lh.data());
    for (int l = 0; l < nlevels; l++) {
        const int rows = lh[l] + 2 * AMOS_EDGE_THRESHOLD, cols = lw[l] + 2 * AMOS_EDGE_THRESHOLD;
        cv::Mat &temp = mvPyramidStore[l];
        if (temp.rows != rows || temp.cols != cols || temp.type() != CV_8UC1) temp = cv::Mat(rows, cols, CV_8UC1);
        mvImagePyramid[l] = temp(cv::Rect(AMOS_EDGE_THRESHOLD, AMOS_EDGE_THRESHOLD, lw[l], lh[l]));
    }
    mbPyramidOnHost = false;
    if (download) DownloadPyramid();
}

void ORBextractor::DownloadPyramid()
{
    if (!mpHandle) throw std::runtime_error("ORBextractor::DownloadPyramid before operator()");
    if (mbPyramidOnHost) return;
    std::vector<uint8_t *> dst(nlevels);
    std::vector<size_t> strides(nlevels);
    for (int l = 0; l < nlevels; l++) {
        dst[l] = mvPyramidStore[l].data;
        strides[l] = mvPyramidStore[l].step;
    }
    Check(amos_orb_pyramid_images(mpHandle, 0, dst.data(), strides.data(), 1), "amos_orb_pyramid_images");  // one transfer for all levels
    mbPyramidOnHost = true;
}

void ORBextractor::Detect(const cv::Mat &image)
{
    EnsureHandle(image.cols, image.rows);
    Check(amos_orb_detect(mpHandle, image.data, image.step, image.cols, image.rows), "amos_orb_detect");
    mvLevelOffset.resize(nlevels);
    mvLevelCap.resize(nlevels);
    Check(amos_orb_level_layout(mpHandle, mvLevelOffset.data(), mvLevelCap.data(), &mnLevelTotal), "amos_orb_level_layout");
    mbDeviceListsKnown = false;
    UpdatePyramid(image.cols, image.rows, mnPyramidMode == PYRAMID_ALWAYS);
}

void ORBextractor::FetchLevels(std::vector<std::vector<cv::KeyPoint>> &levels)
{
    mvDeviceCounts.resize(nlevels);
    mvDeviceLists.resize(mnLevelTotal);
    Check(amos_orb_fetch_levels(mpHandle, 0, mvDeviceCounts.data(), AsAmos(mvDeviceLists.data()), mnLevelTotal), "amos_orb_fetch_levels");
    mbDeviceListsKnown = true;
    levels.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        levels[l].resize(mvDeviceCounts[l]);
        if (mvDeviceCounts[l]) std::memcpy(levels[l].data(), mvDeviceLists.data() + mvLevelOffset[l], sizeof(amos_keypoint) * mvDeviceCounts[l]);
    }
}

void ORBextractor::StoreLevels(const std::vector<std::vector<cv::KeyPoint>> &levels)
{
    bool same = mbDeviceListsKnown;
    for (int l = 0; l < nlevels; l++) {
        const int n = l < (int)levels.size() ? (int)levels[l].size() : 0;
        if (n > mvLevelCap[l]) throw std::runtime_error("ORBextractor: more keypoints on a level than the extractor produced");
        same = same && n == mvDeviceCounts[l] && (n == 0 || std::memcmp(levels[l].data(), mvDeviceLists.data() + mvLevelOffset[l], sizeof(amos_keypoint) * n) == 0);
    }
    if (same) return;  // the device already holds exactly these lists
    mvDeviceCounts.assign(nlevels, 0);
    mvDeviceLists.resize(mnLevelTotal);
    for (int l = 0; l < nlevels && l < (int)levels.size(); l++) {
        mvDeviceCounts[l] = (int)levels[l].size();
        if (mvDeviceCounts[l]) std::memcpy(mvDeviceLists.data() + mvLevelOffset[l], levels[l].data(), sizeof(amos_keypoint) * mvDeviceCounts[l]);
    }
    Check(amos_orb_store_levels(mpHandle, 0, mvDeviceCounts.data(), AsAmos(mvDeviceLists.data()), mnLevelTotal), "amos_orb_store_levels");
    mbDeviceListsKnown = true;
}

// ORBextractor.cc:1544-1668
void ORBextractor::operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    EnsureHandle(image.cols, image.rows);
    int total = 0;
    amos_orb_level_layout(mpHandle, nullptr, nullptr, &total);  // may fail before the first frame: capacity below covers it
    const int cap = std::max(total, nfeatures * 2 + 64 * nlevels);
    mvStage.resize(cap);
    mvStageDesc.resize((size_t)cap * 32);
    int n = 0;
    Check(amos_orb_extract(mpHandle, image.data, image.step, image.cols, image.rows, AsAmos(mvStage.data()), mvStageDesc.data(), cap, &n), "amos_orb_extract");
    mvLevelOffset.resize(nlevels);
    mvLevelCap.resize(nlevels);
    Check(amos_orb_level_layout(mpHandle, mvLevelOffset.data(), mvLevelCap.data(), &mnLevelTotal), "amos_orb_level_layout");
    mbDeviceListsKnown = false;
    UpdatePyramid(image.cols, image.rows, mnPyramidMode != PYRAMID_NEVER);  // AUTO: this is the entry the stereo / mono constructors use
    if (n == 0) {
        _descriptors.release();  // :1590
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous()) std::memcpy(d.ptr(0), mvStageDesc.data(), (size_t)n * 32);
        else for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), mvStageDesc.data() + (size_t)i * 32, 32);
    }
    _keypoints.resize(n);
    if (n) std::memcpy(_keypoints.data(), mvStage.data(), sizeof(amos_keypoint) * n);
}

// ORBextractor.cc:1672-1686
void ORBextractor::operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<std::vector<cv::KeyPoint>> &_keypoints)
{
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    Detect(image);
    FetchLevels(_keypoints);
}

// ORBextractor.cc:1688-1745.  imGray, DynaFlag are unused there too.
std::vector<cv::KeyPoint> ORBextractor::MovingKeyPoints(const cv::Mat &imGray, const cv::Mat &imS, const cv::Mat &imLS,
                                                        std::vector<center> centers, std::vector<int> rm_vector,
                                                        std::vector<bool> DynaFlag, std::vector<std::vector<cv::KeyPoint>> &mvKeysT)
{
    if (!mpHandle) throw std::runtime_error("ORBextractor::MovingKeyPoints before operator()");
    StoreLevels(mvKeysT);
    std::vector<int32_t> ids(centers.size());
    for (size_t i = 0; i < centers.size(); i++) ids[i] = centers[i].id;
    std::vector<cv::KeyPoint> DynaPt(mnLevelTotal + 1);
    int nrem = 0;
    const bool labels = !imLS.empty() && !ids.empty() && !rm_vector.empty();
    Check(amos_orb_gate(mpHandle, imS.data, imS.step, labels ? imLS.ptr<double>() : nullptr, labels ? imLS.step / sizeof(double) : 0,
                        labels ? ids.data() : nullptr, (int)ids.size(), labels ? rm_vector.data() : nullptr, (int)rm_vector.size(),
                        AsAmos(DynaPt.data()), (int)DynaPt.size(), &nrem), "amos_orb_gate");
    FetchLevels(mvKeysT);
    DynaPt.resize(nrem);
    return DynaPt;
}

// ORBextractor.cc:1747-1820
void ORBextractor::ProcessDesp(cv::InputArray _image, cv::InputArray _mask, std::vector<std::vector<cv::KeyPoint>> &_allKeypoints,
                               std::vector<cv::KeyPoint> &_mKeypoints, cv::OutputArray _descriptors)
{
    if (!mpHandle) throw std::runtime_error("ORBextractor::ProcessDesp before operator()");
    StoreLevels(_allKeypoints);
    const int cap = mnLevelTotal + 1;
    mvStage.resize(cap);
    mvStageDesc.resize((size_t)cap * 32);
    int n = 0;
    Check(amos_orb_describe(mpHandle, AsAmos(mvStage.data()), mvStageDesc.data(), cap, &n), "amos_orb_describe");
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous()) std::memcpy(d.ptr(0), mvStageDesc.data(), (size_t)n * 32);
        else for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), mvStageDesc.data() + (size_t)i * 32, 32);
    }
    _mKeypoints.resize(n);
    if (n) std::memcpy(_mKeypoints.data(), mvStage.data(), sizeof(amos_keypoint) * n);
    // the reference rescales the caller's per-level vectors in place (:1804-1813)
    for (int level = 1; level < nlevels && level < (int)_allKeypoints.size(); level++) {
        const float scale = mvScaleFactor[level];
        for (cv::KeyPoint &kp : _allKeypoints[level]) kp.pt *= scale;
    }
}

}  // namespace ORB_SLAM2
