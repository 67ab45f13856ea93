"""CPU: the reference-signature ORBmatcher (amos-slam_amd/host/ORBmatcher_adaptors.h) compiles both ways -- as the
template instantiated on the stand-in classes inside libamos_host.so, and as the class named ORB_SLAM2::ORBmatcher of
the reference tree (AMOS_REFERENCE_TREE) with the reference's call sites."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_call_sites_compile_against_the_drop_in_class():
    src = os.path.join(ROOT, "tests", "host", "ref_tree_compile_check.cc")
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-parameter", "-Wno-class-memaccess", src],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]


def test_every_reference_signature_is_declared():
    """include/ORBmatcher.h:57-215 of the reference: ten searches + DescriptorDistance, names and parameter lists."""
    hdr = open(os.path.join(ROOT, "amos-slam_amd", "host", "ORBmatcher_adaptors.h")).read()
    flat = re.sub(r"\s+", " ", hdr)
    for sig in ("int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3)",
                "int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)",
                "int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound, const float th, const int ORBdist)",
                "int SearchByProjection(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, std::vector<MapPointT *> &vpMatched, int th)",
                "int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches)",
                "int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12)",
                "int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)",
                "int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)",
                "int SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)",
                "int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th = 3.0)",
                "int Fuse(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint)"):
        assert sig in flat, sig
