// pin_oracle_with_opencv.cpp -- turns "parity unpinned" into a one-command check WHERE OpenCV EXISTS.
//
// The oracle (oracle/orb_oracle.c) restates OpenCV 4.5 primitives from their published algorithms (SURVEY.md
// Appendix A) because OpenCV is not in the image this project is built in; every one sits behind one small function.
// This program runs each of them beside the real library on seeded inputs and reports, per primitive, the first
// mismatch (or "identical").  A mismatch localises the wrong recalled constant / rounding rule; all identical pins the
// oracle, and with it every GPU parity test, to OpenCV.
//
//   g++ -O2 -std=c++17 tools/pin/pin_oracle_with_opencv.cpp oracle/orb_oracle.c oracle/lk_oracle.c oracle/corner_oracle.c -Iinclude -Ioracle \
//       $(pkg-config --cflags --libs opencv4) -lm -o pin_oracle && ./pin_oracle
//
// NOT BUILT in this project's image (no OpenCV headers there): it is a tool for the reference's maintainer, outside
// the product and outside the test suite.  It uses only documented OpenCV 4.x API.
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/video/tracking.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
#include "orb_oracle.h"
void orc_bgr_to_lab(const uint8_t *src, size_t n_px, int blue_idx, uint8_t *dst);  // oracle/lk_oracle.c
int orc_lk_track(const uint8_t *prev, size_t prev_stride, const uint8_t *next, size_t next_stride, int w, int h, const float *prev_pts, int n, int win,
                 int max_level, int max_count, double epsilon, float min_eig_threshold, float *next_pts, uint8_t *status, float *err);  // oracle/lk_oracle.c
void orc_corner_harris(const uint8_t *img, size_t stride, int w, int h, double k, float *dst);                                          // oracle/corner_oracle.c
int orc_good_features_to_track(const uint8_t *img, size_t stride, int w, int h, int max_corners, double quality, double min_distance, double k,
                               float *xy, int cap, float *response_out);                                                              // oracle/corner_oracle.c
int orc_corner_subpix(const uint8_t *img, size_t stride, int w, int h, float *xy, int n, int win, int max_count, double epsilon);      // oracle/corner_oracle.c
}

static int g_failures = 0;

static cv::Mat textured(int w, int h, unsigned seed)
{
    std::mt19937 rng(seed);
    cv::Mat m(h, w, CV_8UC1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) m.at<uchar>(y, x) = (uchar)((90 * (((x / 16) + (y / 12)) % 2)) + rng() % 60 + (x * 7 + y * 3) % 40);
    for (int k = 0; k < 60; k++) {  // rectangles and discs: FAST corners with all kinds of scores
        const int cx = rng() % w, cy = rng() % h, r = 4 + rng() % 25;
        if (k & 1) cv::rectangle(m, cv::Rect(cx, cy, r, r + 3) & cv::Rect(0, 0, w, h), cv::Scalar(rng() % 256), cv::FILLED);
        else cv::circle(m, cv::Point(cx, cy), r, cv::Scalar(rng() % 256), cv::FILLED);
    }
    return m;
}

static void report(const char *what, const cv::Mat &want, const cv::Mat &got)
{
    if (want.size() != got.size() || want.type() != got.type()) {
        std::printf("%-34s SIZE / TYPE MISMATCH\n", what);
        g_failures++;
        return;
    }
    for (int y = 0; y < want.rows; y++)
        if (std::memcmp(want.ptr(y), got.ptr(y), (size_t)want.cols * want.elemSize()) != 0) {
            int x = 0;
            while (std::memcmp(want.ptr(y) + (size_t)x * want.elemSize(), got.ptr(y) + (size_t)x * want.elemSize(), want.elemSize()) == 0) x++;
            std::printf("%-34s FIRST MISMATCH at row %d col %d\n", what, y, x);
            g_failures++;
            return;
        }
    std::printf("%-34s identical (%d x %d)\n", what, want.cols, want.rows);
}

int main()
{
    std::printf("OpenCV %s\n", CV_VERSION);
    // A.1 cv::resize INTER_LINEAR, 8UC1, the pyramid's scale steps (ORBextractor.cc:1848)
    for (const cv::Size dst : {cv::Size(533, 400), cv::Size(444, 333), cv::Size(179, 134), cv::Size(480, 640)}) {
        const cv::Mat src = textured(640, 480, 1);
        cv::Mat want, got(dst, CV_8UC1);
        cv::resize(src, want, dst, 0, 0, cv::INTER_LINEAR);
        orc_resize_linear_u8(src.data, src.cols, src.rows, src.step, got.data, dst.width, dst.height, got.step);
        char name[64];
        std::snprintf(name, sizeof(name), "resize 640x480 -> %dx%d", dst.width, dst.height);
        report(name, want, got);
    }
    // A.2 GaussianBlur 7x7 sigma 2, REFLECT_101 (ORBextractor.cc:1793)
    {
        const cv::Mat src = textured(533, 400, 2);
        cv::Mat want, got(src.size(), CV_8UC1);
        cv::GaussianBlur(src, want, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        orc_gaussian_blur7(src.data, src.step, src.cols, src.rows, got.data, got.step);
        report("GaussianBlur 7x7 sigma 2", want, got);
    }
    // A.3 cv::FAST TYPE_9_16 with non-max suppression, thresholds 20 and 7 (ORBextractor.cc:1126, 1135)
    for (const int th : {20, 7}) {
        const cv::Mat src = textured(36, 36, 3 + th);  // one cell with its 3-pixel halo
        std::vector<cv::KeyPoint> want;
        cv::FAST(src, want, th, true);
        std::vector<amos_keypoint> got(4096);
        const int n = orc_fast9_16(src.data, src.step, src.cols, src.rows, th, got.data(), (int)got.size());
        bool same = n == (int)want.size();
        for (int i = 0; same && i < n; i++)
            same = want[i].pt.x == got[i].x && want[i].pt.y == got[i].y && want[i].response == got[i].response && want[i].size == got[i].size;
        std::printf("%-34s %s (%d vs %d keypoints)\n", th == 20 ? "FAST threshold 20" : "FAST threshold 7", same ? "identical" : "MISMATCH", (int)want.size(), n);
        g_failures += !same;
    }
    // A.4 cv::fastAtan2 over a dense sample of moment pairs
    {
        std::mt19937 rng(5);
        int bad = 0;
        for (int i = 0; i < 2000000 && bad == 0; i++) {
            const float y = (float)((int)(rng() % 400001) - 200000), x = (float)((int)(rng() % 400001) - 200000);
            const float w = cv::fastAtan2(y, x), g = orc_fast_atan2(y, x);
            if (std::memcmp(&w, &g, 4) != 0) { std::printf("%-34s MISMATCH at (%g, %g): %.9g vs %.9g\n", "fastAtan2", y, x, w, g); bad = 1; }
        }
        if (!bad) std::printf("%-34s identical (2 000 000 samples)\n", "fastAtan2");
        g_failures += bad;
    }
    // A.5 copyMakeBorder REFLECT_101 by 19 (ORBextractor.cc:1859, 1880) -- through the oracle's padded level 0
    {
        const cv::Mat src = textured(640, 480, 6);
        cv::Mat want;
        cv::copyMakeBorder(src, want, 19, 19, 19, 19, cv::BORDER_REFLECT_101);
        const amos_orb_params p = {1000, 1.2f, 8, 20, 7};
        orc_extractor *e = orc_create(&p);
        orc_detect(e, src.data, src.step, src.cols, src.rows);
        cv::Mat got(480 + 38, 640 + 38, CV_8UC1);
        orc_level_image(e, 0, got.data, got.step, 1);
        report("copyMakeBorder REFLECT_101 19", want, got);
        orc_destroy(e);
    }
    // A.6 dilate + erode with the 31x31 ellipse (ORBextractor.cc:1699-1704)
    {
        cv::Mat mask = cv::Mat::zeros(480, 640, CV_8UC1);
        cv::ellipse(mask, cv::Point(300, 250), cv::Size(70, 150), 10, 0, 360, cv::Scalar(255), cv::FILLED);
        cv::circle(mask, cv::Point(310, 240), 9, cv::Scalar(0), cv::FILLED);
        cv::rectangle(mask, cv::Rect(600, 440, 40, 40), cv::Scalar(254), cv::FILLED);
        const cv::Mat kernel = cv::getStructuringElement(cv::MORPH_ELLIPSE, cv::Size(31, 31), cv::Point(15, 15));
        cv::Mat dil, want, got(mask.size(), CV_8UC1);
        cv::dilate(mask, dil, kernel);
        cv::erode(dil, want, kernel);
        orc_close_ellipse31(mask.data, mask.step, mask.cols, mask.rows, got.data, got.step);
        report("dilate + erode, 31x31 ellipse", want, got);
    }
    // cvtColor BGR2GRAY / RGB2GRAY (Tracking.cc:308-321)
    for (const int rgb : {0, 1}) {
        std::mt19937 rng(7);
        cv::Mat src(480, 640, CV_8UC3);
        for (int i = 0; i < 480 * 640 * 3; i++) src.data[i] = (uchar)rng();
        cv::Mat want, got(480, 640, CV_8UC1);
        cv::cvtColor(src, want, rgb ? cv::COLOR_RGB2GRAY : cv::COLOR_BGR2GRAY);
        orc_color_to_gray(src.data, src.step, 640, 480, 3, rgb, got.data, got.step);
        report(rgb ? "cvtColor RGB2GRAY" : "cvtColor BGR2GRAY", want, got);
    }
    // cvtColor BGR2Lab, 8-bit (cluster.cc:310): every (B, G) pair over a coarse R grid
    {
        cv::Mat src(256, 256 * 16, CV_8UC3);
        for (int b = 0; b < 256; b++)
            for (int g = 0; g < 256; g++)
                for (int r = 0; r < 16; r++) src.at<cv::Vec3b>(b, g * 16 + r) = cv::Vec3b((uchar)b, (uchar)g, (uchar)(r * 17));
        cv::Mat want, got(src.size(), CV_8UC3);
        cv::cvtColor(src, want, cv::COLOR_BGR2Lab);
        orc_bgr_to_lab(src.data, (size_t)src.rows * src.cols, 0, got.data);
        report("cvtColor BGR2Lab", want, got);
    }
    // cv::undistortPoints with the TUM1 coefficients (Frame.cc:1052-1118)
    {
        const float fx = 517.306408f, fy = 516.469215f, cx = 318.643040f, cy = 255.313989f, dist[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
        cv::Mat K = (cv::Mat_<float>(3, 3) << fx, 0, cx, 0, fy, cy, 0, 0, 1), D(1, 5, CV_32F, (void *)dist);
        std::mt19937 rng(8);
        cv::Mat pts(1000, 2, CV_32F);
        for (int i = 0; i < 1000; i++) { pts.at<float>(i, 0) = (float)(rng() % 6400) / 10.f; pts.at<float>(i, 1) = (float)(rng() % 4800) / 10.f; }
        cv::Mat want = pts.clone().reshape(2);
        cv::undistortPoints(want, want, K, D, cv::Mat(), K);
        want = want.reshape(1);
        cv::Mat got(1000, 2, CV_32F);
        orc_undistort_points(pts.ptr<float>(), 1000, fx, fy, cx, cy, dist, 5, got.ptr<float>());
        report("undistortPoints (TUM1)", want, got);
    }
    // cv::calcOpticalFlowPyrLK as Tracking.cc:896 calls it: 22 x 22 window, maxLevel 5, 20 iterations / eps 0.01, default flags and
    // minEigThreshold 1e-4.  The library accumulates the 2 x 2 system and the residual through the SIMD lanes of its build, so the
    // last bits of the tracked positions belong to that binary: reported as identical / first mismatch like the others, plus the
    // largest position difference in pixels so that "one ulp apart" can be told from "another algorithm".
    {
        const cv::Mat prev = textured(640, 480, 9);
        cv::Mat next;
        const cv::Mat shift = (cv::Mat_<double>(2, 3) << 1, 0, 3.25, 0, 1, -1.5);
        cv::warpAffine(prev, next, shift, prev.size(), cv::INTER_LINEAR, cv::BORDER_REFLECT_101);
        std::mt19937 rng(10);
        std::vector<cv::Point2f> p0(1000), p1;
        for (auto &p : p0) p = cv::Point2f((float)(rng() % 6400) / 10.f, (float)(rng() % 4800) / 10.f);
        std::vector<uchar> st;
        std::vector<float> er;
        cv::calcOpticalFlowPyrLK(prev, next, p0, p1, st, er, cv::Size(22, 22), 5, cv::TermCriteria(cv::TermCriteria::COUNT | cv::TermCriteria::EPS, 20, 0.01));
        cv::Mat want(1000, 2, CV_32F), got(1000, 2, CV_32F), wantSt(1000, 1, CV_8U), gotSt(1000, 1, CV_8U), wantEr(1000, 1, CV_32F), gotEr(1000, 1, CV_32F);
        for (int i = 0; i < 1000; i++) {
            want.at<float>(i, 0) = p1[i].x;
            want.at<float>(i, 1) = p1[i].y;
            wantSt.at<uchar>(i) = st[i];
            wantEr.at<float>(i) = er[i];
        }
        orc_lk_track(prev.data, prev.step, next.data, next.step, 640, 480, &p0[0].x, 1000, 22, 5, 20, 0.01, 1e-4f, got.ptr<float>(), gotSt.data, gotEr.ptr<float>());
        report("calcOpticalFlowPyrLK status", wantSt, gotSt);
        report("calcOpticalFlowPyrLK positions", want, got);
        report("calcOpticalFlowPyrLK error", wantEr, gotEr);
        double worst = 0;
        for (int i = 0; i < 1000; i++)
            if (st[i] && gotSt.at<uchar>(i)) worst = std::max(worst, (double)std::max(std::fabs(want.at<float>(i, 0) - got.at<float>(i, 0)), std::fabs(want.at<float>(i, 1) - got.at<float>(i, 1))));
        std::printf("%-34s largest position difference %.3g px\n", "calcOpticalFlowPyrLK", worst);
    }
    // cv::goodFeaturesToTrack + cv::cornerSubPix as Tracking.cc:894-895 call them (the corner source of GetSceneFlowObj): 1000 corners,
    // quality 0.01, minDistance 8, blockSize 3, Harris detector with k = 0.04; refinement window 10 x 10 (half size), no dead zone,
    // 20 iterations / eps 0.03.  cornerHarris itself first (its response plane decides everything after it), then the selected corners
    // in the library's order, then the refined positions (float arithmetic of the library's build: largest difference reported too).
    {
        const cv::Mat img = textured(640, 480, 11);
        cv::Mat wantR, gotR(480, 640, CV_32F);
        cv::cornerHarris(img, wantR, 3, 3, 0.04, cv::BORDER_DEFAULT);
        orc_corner_harris(img.data, img.step, 640, 480, 0.04, gotR.ptr<float>());
        report("cornerHarris(3, 3, 0.04)", wantR, gotR);
        std::vector<cv::Point2f> pts;
        cv::goodFeaturesToTrack(img, pts, 1000, 0.01, 8, cv::Mat(), 3, true, 0.04);
        std::vector<float> xy(2 * 1000);
        const int n = orc_good_features_to_track(img.data, img.step, 640, 480, 1000, 0.01, 8.0, 0.04, xy.data(), 1000, nullptr);
        cv::Mat want((int)pts.size(), 2, CV_32F), got(n, 2, CV_32F);
        for (int i = 0; i < (int)pts.size(); i++) { want.at<float>(i, 0) = pts[i].x; want.at<float>(i, 1) = pts[i].y; }
        for (int i = 0; i < n; i++) { got.at<float>(i, 0) = xy[2 * i]; got.at<float>(i, 1) = xy[2 * i + 1]; }
        if ((int)pts.size() != n) { std::printf("%-34s MISMATCH: %zu corners, oracle %d\n", "goodFeaturesToTrack count", pts.size(), n); g_failures++; }
        else report("goodFeaturesToTrack (Harris, 8 px)", want, got);
        if (!pts.empty() && (int)pts.size() == n) {
            cv::cornerSubPix(img, pts, cv::Size(10, 10), cv::Size(-1, -1), cv::TermCriteria(cv::TermCriteria::COUNT | cv::TermCriteria::EPS, 20, 0.03));
            orc_corner_subpix(img.data, img.step, 640, 480, xy.data(), n, 10, 20, 0.03);
            double worst = 0;
            for (int i = 0; i < n; i++) {
                want.at<float>(i, 0) = pts[i].x; want.at<float>(i, 1) = pts[i].y;
                got.at<float>(i, 0) = xy[2 * i]; got.at<float>(i, 1) = xy[2 * i + 1];
                worst = std::max(worst, (double)std::max(std::fabs(pts[i].x - xy[2 * i]), std::fabs(pts[i].y - xy[2 * i + 1])));
            }
            report("cornerSubPix(10 x 10, 20 / 0.03)", want, got);
            std::printf("%-34s largest position difference %.3g px\n", "cornerSubPix", worst);
        }
    }
    std::printf(g_failures ? "\n%d primitive(s) differ: the oracle is NOT pinned by this OpenCV build\n" : "\nall identical: the oracle is pinned to this OpenCV build\n", g_failures);
    return g_failures ? 1 : 0;
}
