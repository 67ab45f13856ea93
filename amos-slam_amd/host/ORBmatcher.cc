// ORBmatcher.cc -- see ORBmatcher.h.  Line references are to the reference's src/ORBmatcher.cc and
// src/Frame.cc.
#include "ORBmatcher.h"

#include <climits>
#include <cmath>
#include <mutex>
#include <stdexcept>
#include <string>

using namespace std;

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static void Check(int rc, const char *what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + amos_last_error());
}

// ---------------------------------------------------------------------------------------------
FeatureGrid::FeatureGrid(const amos_frame_view &frame) : mFrame(frame)
{
    mfGridElementWidthInv = static_cast<float>(AMOS_FRAME_GRID_COLS) / static_cast<float>(frame.max_x - frame.min_x);   // Frame.cc:302
    mfGridElementHeightInv = static_cast<float>(AMOS_FRAME_GRID_ROWS) / static_cast<float>(frame.max_y - frame.min_y);
    for (int i = 0; i < frame.n; i++) {  // AssignFeaturesToGrid, Frame.cc:431-461
        int nGridPosX, nGridPosY;
        if (PosInGrid(frame.keys_un[i], nGridPosX, nGridPosY)) mGrid[nGridPosX][nGridPosY].push_back(i);
    }
}

bool FeatureGrid::PosInGrid(const amos_keypoint &kp, int &posX, int &posY) const
{
    posX = round((kp.x - mFrame.min_x) * mfGridElementWidthInv);  // Frame.cc:1019-1020
    posY = round((kp.y - mFrame.min_y) * mfGridElementHeightInv);
    if (posX < 0 || posX >= AMOS_FRAME_GRID_COLS || posY < 0 || posY >= AMOS_FRAME_GRID_ROWS) return false;
    return true;
}

// Frame.cc:894-1003
vector<size_t> FeatureGrid::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel, const int maxLevel) const
{
    vector<size_t> vIndices;
    vIndices.reserve(mFrame.n);
    const int nMinCellX = max(0, (int)floor((x - mFrame.min_x - r) * mfGridElementWidthInv));
    if (nMinCellX >= AMOS_FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = min((int)AMOS_FRAME_GRID_COLS - 1, (int)ceil((x - mFrame.min_x + r) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = max(0, (int)floor((y - mFrame.min_y - r) * mfGridElementHeightInv));
    if (nMinCellY >= AMOS_FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = min((int)AMOS_FRAME_GRID_ROWS - 1, (int)ceil((y - mFrame.min_y + r) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const vector<size_t> &vCell = mGrid[ix][iy];
            if (vCell.empty()) continue;
            for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                const amos_keypoint &kpUn = mFrame.keys_un[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x;
                const float disty = kpUn.y - y;
                if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
            }
        }
    }
    return vIndices;
}

// ---------------------------------------------------------------------------------------------
ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri), mpMatch(nullptr)
{
    Check(amos_match_create(0, nullptr, &mpMatch), "amos_match_create");
}

ORBmatcher::~ORBmatcher()
{
    if (mpMatch) amos_match_destroy(mpMatch);
}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    static std::mutex mtx;
    static amos_match *shared = nullptr;
    std::lock_guard<std::mutex> lock(mtx);
    if (!shared) Check(amos_match_create(0, nullptr, &shared), "amos_match_create");
    uint16_t d = 0;
    Check(amos_match_distances(shared, a.ptr(), 1, b.ptr(), 1, &d), "amos_match_distances");
    return d;
}

void ORBmatcher::DescriptorDistances(const uint8_t *q, int nq, const uint8_t *t, int nt, std::vector<uint16_t> &out)
{
    out.resize((size_t)nq * nt);
    Check(amos_match_distances(mpMatch, q, nq, t, nt, out.data()), "amos_match_distances");
}

void ORBmatcher::ListDistances(const amos_frame_view &train, const uint8_t *queries, int nq, const std::vector<int> &off,
                               const std::vector<int> &idx, std::vector<uint16_t> &dist)
{
    dist.resize(idx.size());
    if (idx.empty()) return;
    Check(amos_match_list_distances(mpMatch, queries, nq, train.descriptors, train.n, off.data(), idx.data(), dist.data()),
          "amos_match_list_distances");
}

void ORBmatcher::ListDistances(const uint8_t *train, int nt, const uint8_t *queries, int nq, const std::vector<int> &off,
                               const std::vector<int> &idx, std::vector<uint16_t> &dist)
{
    dist.resize(idx.size());
    if (idx.empty()) return;
    Check(amos_match_list_distances(mpMatch, queries, nq, train, nt, off.data(), idx.data(), dist.data()), "amos_match_list_distances");
}

float ORBmatcher::RadiusByViewingCos(const float &viewCos)
{
    if (viewCos > 0.998)
        return 2.5;
    else
        return 4.0;
}

// ORBmatcher.cc:1569-1728
int ORBmatcher::SearchByProjection(const FeatureGrid &CurrentFrame, const vector<amos_proj_query> &vLastPoints, vector<int> &vnCurMatch,
                                   const vector<float> &mvScaleFactors, float mbf, const float th, const bool bForward, const bool bBackward)
{
    const amos_frame_view &F = CurrentFrame.Frame();
    const int nq = (int)vLastPoints.size();
    // 1. candidates of every query, in GetFeaturesInArea order (:1627-1637)
    vector<int> off(nq + 1, 0), idx;
    vector<float> radius(nq);
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) {
        const amos_proj_query &p = vLastPoints[i];
        memcpy(&qdesc[(size_t)i * 32], p.desc, 32);
        const int nLastOctave = p.octave;
        radius[i] = th * mvScaleFactors[nLastOctave];
        vector<size_t> vIndices2;
        if (bForward)
            vIndices2 = CurrentFrame.GetFeaturesInArea(p.u, p.v, radius[i], nLastOctave);
        else if (bBackward)
            vIndices2 = CurrentFrame.GetFeaturesInArea(p.u, p.v, radius[i], 0, nLastOctave);
        else
            vIndices2 = CurrentFrame.GetFeaturesInArea(p.u, p.v, radius[i], nLastOctave - 1, nLastOctave + 1);
        for (size_t k = 0; k < vIndices2.size(); k++) idx.push_back((int)vIndices2[k]);
        off[i + 1] = (int)idx.size();
    }
    // 2. all candidate distances in one GPU call
    vector<uint16_t> dist;
    ListDistances(F, qdesc.data(), nq, off, idx, dist);
    // 3. the reference's greedy loop over the precomputed distances
    int nmatches = 0;
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < nq; i++) {
        if (off[i] == off[i + 1]) continue;
        const amos_proj_query &p = vLastPoints[i];
        int bestDist = 256;
        int bestIdx2 = -1;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int i2 = idx[k];
            if (vnCurMatch[i2] >= 0)
                if (vLastPoints[vnCurMatch[i2]].has_obs) continue;  // :1658-1660
            if (F.u_right && F.u_right[i2] > 0) {                    // :1662-1669
                const float ur = p.u - mbf * p.invz;
                const float er = fabs(ur - F.u_right[i2]);
                if (er > radius[i]) continue;
            }
            const int d = dist[k];
            if (d < bestDist) {
                bestDist = d;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            vnCurMatch[bestIdx2] = i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = p.angle - F.keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    vnCurMatch[rotHist[i][j]] = -1;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// ORBmatcher.cc:70-175
int ORBmatcher::SearchByProjection(const FeatureGrid &Fg, const vector<amos_map_query> &vpMapPoints, vector<int> &vnCurMatch,
                                   vector<bool> &vbCurHasObs, const vector<float> &mvScaleFactors, const float th)
{
    const amos_frame_view &F = Fg.Frame();
    const int nq = (int)vpMapPoints.size();
    const bool bFactor = th != 1.0;
    vector<int> off(nq + 1, 0), idx;
    vector<float> rr(nq);
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int iMP = 0; iMP < nq; iMP++) {
        const amos_map_query &mp = vpMapPoints[iMP];
        memcpy(&qdesc[(size_t)iMP * 32], mp.desc, 32);
        const int nPredictedLevel = mp.level;
        float r = RadiusByViewingCos(mp.view_cos);
        if (bFactor) r *= th;
        rr[iMP] = r;
        const vector<size_t> vIndices =
            Fg.GetFeaturesInArea(mp.proj_x, mp.proj_y, r * mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
        for (size_t k = 0; k < vIndices.size(); k++) idx.push_back((int)vIndices[k]);
        off[iMP + 1] = (int)idx.size();
    }
    vector<uint16_t> dist;
    ListDistances(F, qdesc.data(), nq, off, idx, dist);
    int nmatches = 0;
    for (int iMP = 0; iMP < nq; iMP++) {
        if (off[iMP] == off[iMP + 1]) continue;
        const amos_map_query &mp = vpMapPoints[iMP];
        const int nPredictedLevel = mp.level;
        const float r = rr[iMP];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = off[iMP]; k < off[iMP + 1]; k++) {
            const int i = idx[k];
            if (vbCurHasObs[i]) continue;  // :121-123
            if (F.u_right && F.u_right[i] > 0) {
                const float er = fabs(mp.proj_xr - F.u_right[i]);
                if (er > r * mvScaleFactors[nPredictedLevel]) continue;
            }
            const int d = dist[k];
            if (d < bestDist) {
                bestDist2 = bestDist;
                bestDist = d;
                bestLevel2 = bestLevel;
                bestLevel = F.keys_un[i].octave;
                bestIdx = i;
            } else if (d < bestDist2) {
                bestLevel2 = F.keys_un[i].octave;
                bestDist2 = d;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;  // :163-167
            vnCurMatch[bestIdx] = iMP;
            vbCurHasObs[bestIdx] = mp.has_obs != 0;
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher.cc:1731-1863 (relocalisation): window nPredictedLevel-1 .. +1, ANY occupied feature is skipped,
// best only, accepted at bestDist <= ORBdist, rotation histogram pruning unconditional.
int ORBmatcher::SearchByProjection(const FeatureGrid &CurrentFrame, const vector<amos_kf_query> &vKFPoints, vector<int> &vnCurMatch,
                                   const vector<float> &mvScaleFactors, const float th, const int ORBdist)
{
    const amos_frame_view &F = CurrentFrame.Frame();
    const int nq = (int)vKFPoints.size();
    vector<int> off(nq + 1, 0), idx;
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) {
        const amos_kf_query &p = vKFPoints[i];
        memcpy(&qdesc[(size_t)i * 32], p.desc, 32);
        const int nPredictedLevel = p.level;
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const vector<size_t> vIndices2 = CurrentFrame.GetFeaturesInArea(p.u, p.v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        for (size_t k = 0; k < vIndices2.size(); k++) idx.push_back((int)vIndices2[k]);
        off[i + 1] = (int)idx.size();
    }
    vector<uint16_t> dist;
    ListDistances(F, qdesc.data(), nq, off, idx, dist);
    int nmatches = 0;
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < nq; i++) {
        if (off[i] == off[i + 1]) continue;  // :1802-1803
        int bestDist = 256;
        int bestIdx2 = -1;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int i2 = idx[k];
            if (vnCurMatch[i2] != AMOS_MATCH_FREE) continue;  // :1816-1817
            const int d = dist[k];
            if (d < bestDist) {
                bestDist = d;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= ORBdist) {
            vnCurMatch[bestIdx2] = i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = vKFPoints[i].angle - F.keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    vnCurMatch[rotHist[i][j]] = AMOS_MATCH_FREE;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// ORBmatcher.cc:230-382.  Candidates of a keyframe feature = the frame's features in the same vocabulary node,
// in the FeatureVector's order; the two-iterator merge over the ascending node ids is the reference's
// (lower_bound on a std::map = first node id >= the other side's).
int ORBmatcher::SearchByBoW(const amos_bow_view &KF, const amos_bow_view &F, vector<int> &vnMatchesF)
{
    vnMatchesF.assign(F.n, -1);
    // 1. candidate lists, in the order the reference visits the keyframe features
    vector<int> qKF, off(1, 0), idx;
    {
        int a = 0, b = 0;
        while (a < KF.n_nodes && b < F.n_nodes) {
            if (KF.node_ids[a] == F.node_ids[b]) {
                for (int k = KF.node_off[a]; k < KF.node_off[a + 1]; k++) {
                    const int realIdxKF = KF.node_idx[k];
                    if (KF.has_point && !KF.has_point[realIdxKF]) continue;  // !pMP || pMP->isBad()
                    qKF.push_back(realIdxKF);
                    for (int m = F.node_off[b]; m < F.node_off[b + 1]; m++) idx.push_back(F.node_idx[m]);
                    off.push_back((int)idx.size());
                }
                a++;
                b++;
            } else if (KF.node_ids[a] < F.node_ids[b]) {
                while (a < KF.n_nodes && KF.node_ids[a] < F.node_ids[b]) a++;  // lower_bound(Fit->first)
            } else {
                while (b < F.n_nodes && F.node_ids[b] < KF.node_ids[a]) b++;
            }
        }
    }
    const int nq = (int)qKF.size();
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) memcpy(&qdesc[(size_t)i * 32], KF.descriptors + (size_t)qKF[i] * 32, 32);
    // 2. all candidate distances in one GPU call
    vector<uint16_t> dist;
    ListDistances(F.descriptors, F.n, qdesc.data(), nq, off, idx, dist);
    // 3. the reference's greedy loop
    int nmatches = 0;
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < nq; i++) {
        const int realIdxKF = qKF[i];
        int bestDist1 = 256;
        int bestIdxF = -1;
        int bestDist2 = 256;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int realIdxF = idx[k];
            if (vnMatchesF[realIdxF] >= 0) continue;  // :288-289
            const int d = dist[k];
            if (d < bestDist1) {
                bestDist2 = bestDist1;
                bestDist1 = d;
                bestIdxF = realIdxF;
            } else if (d < bestDist2) {
                bestDist2 = d;
            }
        }
        if (bestDist1 <= TH_LOW) {
            if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                vnMatchesF[bestIdxF] = realIdxKF;
                if (mbCheckOrientation) {
                    float rot = KF.keys[realIdxKF].angle - F.keys[bestIdxF].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdxF);
                }
                nmatches++;
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vnMatchesF[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// ORBmatcher.cc:515-643
int ORBmatcher::SearchForInitialization(const amos_frame_view &F1, const FeatureGrid &F2g, vector<cv::Point2f> &vbPrevMatched,
                                        vector<int> &vnMatches12, int windowSize)
{
    const amos_frame_view &F2 = F2g.Frame();
    int nmatches = 0;
    vnMatches12 = vector<int>(F1.n, -1);
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    vector<int> vMatchedDistance(F2.n, INT_MAX);
    vector<int> vnMatches21(F2.n, -1);
    // candidates (level 0 features of F1 only, :540-552) and their distances
    vector<int> off(F1.n + 1, 0), idx;
    for (int i1 = 0; i1 < F1.n; i1++) {
        const int level1 = F1.keys_un[i1].octave;
        if (level1 <= 0) {
            vector<size_t> vIndices2 = F2g.GetFeaturesInArea(vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, level1, level1);
            for (size_t k = 0; k < vIndices2.size(); k++) idx.push_back((int)vIndices2[k]);
        }
        off[i1 + 1] = (int)idx.size();
    }
    vector<uint16_t> dist;
    ListDistances(F2, F1.descriptors, F1.n, off, idx, dist);
    for (int i1 = 0; i1 < F1.n; i1++) {
        if (F1.keys_un[i1].octave > 0) continue;
        if (off[i1] == off[i1 + 1]) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = off[i1]; k < off[i1 + 1]; k++) {
            const int i2 = idx[k];
            const int d = dist[k];
            if (vMatchedDistance[i2] <= d) continue;
            if (d < bestDist) {
                bestDist2 = bestDist;
                bestDist = d;
                bestIdx2 = i2;
            } else if (d < bestDist2) {
                bestDist2 = d;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = F1.keys_un[i1].angle - F2.keys_un[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)
        if (vnMatches12[i1] >= 0) {
            vbPrevMatched[i1].x = F2.keys_un[vnMatches12[i1]].x;
            vbPrevMatched[i1].y = F2.keys_un[vnMatches12[i1]].y;
        }
    return nmatches;
}

// ORBmatcher.cc:1866-1908
void ORBmatcher::ComputeThreeMaxima(vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i].size();
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

}  // namespace ORB_SLAM2
