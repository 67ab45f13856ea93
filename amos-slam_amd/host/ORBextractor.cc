// ORBextractor.cc -- host side of the drop-in ORB_SLAM2::ORBextractor: argument marshalling between
// cv::Mat / std::vector<cv::KeyPoint> and the C ABI.  cv::KeyPoint and amos_keypoint share one layout
// (tests/test_constants.py), so keypoint vectors move with memcpy.
#include "ORBextractor.h"

#include <algorithm>
#include <cassert>
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

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mpHandle(nullptr),
      mnHandleW(0), mnHandleH(0), mbDownloadPyramid(true), mnLevelTotal(0)
{
    mvImagePyramid.resize(nlevels);
    // The tables (ORBextractor.cc:500-608) come from the library so both sides agree bit for bit;
    // they do not depend on the frame size, so a small probe handle is enough.
    amos_orb_params p = {nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST};
    amos_orb *probe = nullptr;
    int side = 64;
    for (int l = 1; l < nlevels; l++) side = (int)(side * _scaleFactor) + 1;
    Check(amos_orb_create(&p, side + 64, side + 64, 1, 0, nullptr, &probe), "amos_orb_create");
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    Check(amos_orb_tables(probe, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                          mnFeaturesPerLevel.data(), umax.data()), "amos_orb_tables");
    amos_orb_destroy(probe);
}

ORBextractor::~ORBextractor()
{
    if (mpHandle) amos_orb_destroy(mpHandle);
}

void ORBextractor::EnsureHandle(int width, int height)
{
    if (mpHandle && width <= mnHandleW && height <= mnHandleH) return;
    if (mpHandle) amos_orb_destroy(mpHandle);
    mpHandle = nullptr;
    amos_orb_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    mnHandleW = std::max(width, mnHandleW);
    mnHandleH = std::max(height, mnHandleH);
    Check(amos_orb_create(&p, mnHandleW, mnHandleH, 1, 0, nullptr, &mpHandle), "amos_orb_create");
}

void ORBextractor::UpdatePyramid(int width, int height)
{
    std::vector<int32_t> lw(nlevels), lh(nlevels);
    amos_orb_level_sizes(mpHandle, width, height, lw.data(), lh.data());
    for (int l = 0; l < nlevels; l++) {
        // temp(wholeSize) + ROI, ORBextractor.cc:1835-1838
        cv::Mat temp(lh[l] + 2 * AMOS_EDGE_THRESHOLD, lw[l] + 2 * AMOS_EDGE_THRESHOLD, CV_8UC1);
        if (mbDownloadPyramid) Check(amos_orb_level_image(mpHandle, 0, l, temp.data, temp.step, 1), "amos_orb_level_image");
        mvImagePyramid[l] = temp(cv::Rect(AMOS_EDGE_THRESHOLD, AMOS_EDGE_THRESHOLD, lw[l], lh[l]));
    }
}

void ORBextractor::Detect(const cv::Mat &image)
{
    EnsureHandle(image.cols, image.rows);
    Check(amos_orb_detect(mpHandle, image.data, image.step, image.cols, image.rows), "amos_orb_detect");
    mvLevelOffset.resize(nlevels);
    mvLevelCap.resize(nlevels);
    Check(amos_orb_level_layout(mpHandle, mvLevelOffset.data(), mvLevelCap.data(), &mnLevelTotal), "amos_orb_level_layout");
    UpdatePyramid(image.cols, image.rows);
}

void ORBextractor::FetchLevels(std::vector<std::vector<cv::KeyPoint>> &levels)
{
    std::vector<int32_t> counts(nlevels);
    std::vector<amos_keypoint> buf(mnLevelTotal);
    Check(amos_orb_fetch_levels(mpHandle, 0, counts.data(), buf.data(), mnLevelTotal), "amos_orb_fetch_levels");
    levels.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        levels[l].resize(counts[l]);
        if (counts[l]) std::memcpy(levels[l].data(), buf.data() + mvLevelOffset[l], sizeof(amos_keypoint) * counts[l]);
    }
}

void ORBextractor::StoreLevels(const std::vector<std::vector<cv::KeyPoint>> &levels)
{
    std::vector<int32_t> counts(nlevels, 0);
    std::vector<amos_keypoint> buf(mnLevelTotal);
    for (int l = 0; l < nlevels && l < (int)levels.size(); l++) {
        counts[l] = (int)levels[l].size();
        if (counts[l] > mvLevelCap[l]) throw std::runtime_error("ORBextractor: more keypoints on a level than the extractor produced");
        if (counts[l]) std::memcpy(buf.data() + mvLevelOffset[l], levels[l].data(), sizeof(amos_keypoint) * counts[l]);
    }
    Check(amos_orb_store_levels(mpHandle, 0, counts.data(), buf.data(), mnLevelTotal), "amos_orb_store_levels");
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
    std::vector<amos_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    Check(amos_orb_extract(mpHandle, image.data, image.step, image.cols, image.rows, kps.data(), desc.data(), cap, &n), "amos_orb_extract");
    mvLevelOffset.resize(nlevels);
    mvLevelCap.resize(nlevels);
    Check(amos_orb_level_layout(mpHandle, mvLevelOffset.data(), mvLevelCap.data(), &mnLevelTotal), "amos_orb_level_layout");
    UpdatePyramid(image.cols, image.rows);
    if (n == 0) {
        _descriptors.release();  // :1590
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc.data() + (size_t)i * 32, 32);
    }
    _keypoints.resize(n);
    if (n) std::memcpy(_keypoints.data(), kps.data(), sizeof(amos_keypoint) * n);
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
    std::vector<amos_keypoint> removed(mnLevelTotal + 1);
    int nrem = 0;
    const bool labels = !imLS.empty() && !ids.empty() && !rm_vector.empty();
    Check(amos_orb_gate(mpHandle, imS.data, imS.step, labels ? imLS.ptr<double>() : nullptr, labels ? imLS.step / sizeof(double) : 0,
                        labels ? ids.data() : nullptr, (int)ids.size(), labels ? rm_vector.data() : nullptr, (int)rm_vector.size(),
                        removed.data(), (int)removed.size(), &nrem), "amos_orb_gate");
    FetchLevels(mvKeysT);
    std::vector<cv::KeyPoint> DynaPt(nrem);
    if (nrem) std::memcpy(DynaPt.data(), removed.data(), sizeof(amos_keypoint) * nrem);
    return DynaPt;
}

// ORBextractor.cc:1747-1820
void ORBextractor::ProcessDesp(cv::InputArray _image, cv::InputArray _mask, std::vector<std::vector<cv::KeyPoint>> &_allKeypoints,
                               std::vector<cv::KeyPoint> &_mKeypoints, cv::OutputArray _descriptors)
{
    if (!mpHandle) throw std::runtime_error("ORBextractor::ProcessDesp before operator()");
    StoreLevels(_allKeypoints);
    const int cap = mnLevelTotal + 1;
    std::vector<amos_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    Check(amos_orb_describe(mpHandle, kps.data(), desc.data(), cap, &n), "amos_orb_describe");
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc.data() + (size_t)i * 32, 32);
    }
    _mKeypoints.resize(n);
    if (n) std::memcpy(_mKeypoints.data(), kps.data(), sizeof(amos_keypoint) * n);
    // the reference rescales the caller's per-level vectors in place (:1804-1813)
    for (int level = 1; level < nlevels && level < (int)_allKeypoints.size(); level++) {
        const float scale = mvScaleFactor[level];
        for (cv::KeyPoint &kp : _allKeypoints[level]) kp.pt *= scale;
    }
}

}  // namespace ORB_SLAM2
