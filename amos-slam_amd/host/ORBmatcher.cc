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

const int AMOS_VIEW_MATCHER::TH_HIGH = 100;
const int AMOS_VIEW_MATCHER::TH_LOW = 50;
const int AMOS_VIEW_MATCHER::HISTO_LENGTH = 30;

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
// The reference constructs a matcher ON THE STACK at every call site (Tracking.cc:1492,1744,1910,2378,2609,2663, LocalMapping.cc:324,669,
// LoopClosing.cc:346,813): several per frame, from three threads.  Its constructor stores two numbers (ORBmatcher.cc:49-51), so must this
// one: the device handle (HIP stream + scratch buffers that grow to the largest search seen) is borrowed from a process-wide pool on the
// FIRST search of the object and handed back by the destructor -- no HIP call in either, no hipMalloc / hipFree / stream churn per frame.
// A handle serves one matcher object at a time (it is re-entrant across objects, not within one); the pool is per device and lives as long
// as the process (like the extractors Tracking never frees, Tracking.cc:172-185): destroying HIP objects from a static destructor after the
// runtime has shut down is not defined.
namespace
{
struct MatchPool {
    std::mutex mutex;
    std::vector<std::vector<amos_match *> > idle;  // [device]
    int created = 0;
};
MatchPool &Pool()
{
    static MatchPool *pool = new MatchPool();  // never deleted, see above
    return *pool;
}
}  // namespace

extern int AmosPickDevice();  // ORBextractor.cc: AMOS_DEVICE, else the calling thread's current HIP device

amos_match *AMOS_VIEW_MATCHER::Handle()
{
    if (mpMatch) return mpMatch;
    mnDevice = AmosPickDevice();
    MatchPool &pool = Pool();
    {
        std::lock_guard<std::mutex> lock(pool.mutex);
        if ((int)pool.idle.size() > mnDevice && !pool.idle[mnDevice].empty()) {
            mpMatch = pool.idle[mnDevice].back();
            pool.idle[mnDevice].pop_back();
            return mpMatch;
        }
        pool.created++;
    }
    Check(amos_match_create(mnDevice, nullptr, &mpMatch), "amos_match_create");
    return mpMatch;
}

int AMOS_VIEW_MATCHER::PoolHandlesCreated()
{
    MatchPool &pool = Pool();
    std::lock_guard<std::mutex> lock(pool.mutex);
    return pool.created;
}

AMOS_VIEW_MATCHER::AMOS_VIEW_MATCHER(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri), mpMatch(nullptr), mnDevice(-1)
{
}

AMOS_VIEW_MATCHER::~AMOS_VIEW_MATCHER()
{
    if (!mpMatch) return;
    MatchPool &pool = Pool();
    std::lock_guard<std::mutex> lock(pool.mutex);
    if ((int)pool.idle.size() <= mnDevice) pool.idle.resize(mnDevice + 1);
    pool.idle[mnDevice].push_back(mpMatch);  // every call on a handle is synchronous: nothing of this object is in flight
}

// ORBmatcher.cc:1913-1933: one pair.  Callers use it inside host loops (MapPoint::ComputeDistinctiveDescriptors,
// Frame / KeyFrame bookkeeping), so a single pair is eight host popcounts, not a kernel launch; sets of descriptors
// go through DescriptorDistances / the Search* functions (GPU).
int AMOS_VIEW_MATCHER::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const unsigned char *pa = a.ptr(), *pb = b.ptr();
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, pa + 4 * i, 4);
        memcpy(&y, pb + 4 * i, 4);
        dist += __builtin_popcount(x ^ y);
    }
    return dist;
}

void AMOS_VIEW_MATCHER::DescriptorDistances(const uint8_t *q, int nq, const uint8_t *t, int nt, std::vector<uint16_t> &out)
{
    out.resize((size_t)nq * nt);
    Check(amos_match_distances(Handle(), q, nq, t, nt, out.data()), "amos_match_distances");
}

void AMOS_VIEW_MATCHER::ListDistances(const amos_frame_view &train, const uint8_t *queries, int nq, const std::vector<int> &off,
                               const std::vector<int> &idx, std::vector<uint16_t> &dist)
{
    dist.resize(idx.size());
    if (idx.empty()) return;
    Check(amos_match_list_distances(Handle(), queries, nq, train.descriptors, train.n, off.data(), idx.data(), dist.data()),
          "amos_match_list_distances");
}

void AMOS_VIEW_MATCHER::ListDistances(const uint8_t *train, int nt, const uint8_t *queries, int nq, const std::vector<int> &off,
                               const std::vector<int> &idx, std::vector<uint16_t> &dist)
{
    dist.resize(idx.size());
    if (idx.empty()) return;
    Check(amos_match_list_distances(Handle(), queries, nq, train, nt, off.data(), idx.data(), dist.data()), "amos_match_list_distances");
}

float AMOS_VIEW_MATCHER::RadiusByViewingCos(const float &viewCos)
{
    if (viewCos > 0.998)
        return 2.5;
    else
        return 4.0;
}

// ORBmatcher.cc:1569-1728
int AMOS_VIEW_MATCHER::SearchByProjection(const FeatureGrid &CurrentFrame, const vector<amos_proj_query> &vLastPoints, vector<int> &vnCurMatch,
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
int AMOS_VIEW_MATCHER::SearchByProjection(const FeatureGrid &Fg, const vector<amos_map_query> &vpMapPoints, vector<int> &vnCurMatch,
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
int AMOS_VIEW_MATCHER::SearchByProjection(const FeatureGrid &CurrentFrame, const vector<amos_kf_query> &vKFPoints, vector<int> &vnCurMatch,
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
int AMOS_VIEW_MATCHER::SearchByBoW(const amos_bow_view &KF, const amos_bow_view &F, vector<int> &vnMatchesF)
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

// Candidate lists of a node-by-node search between two feature vectors (the merge of :673-760 / :835-935):
// for every feature of side 1 that `take1` accepts, the features of side 2 in the same node, in list order.
template <typename Take1>
static void BowCandidates(const amos_bow_view &A, const amos_bow_view &B, Take1 take1, vector<int> &q1, vector<int> &off, vector<int> &idx)
{
    q1.clear();
    idx.clear();
    off.assign(1, 0);
    int a = 0, b = 0;
    while (a < A.n_nodes && b < B.n_nodes) {
        if (A.node_ids[a] == B.node_ids[b]) {
            for (int k = A.node_off[a]; k < A.node_off[a + 1]; k++) {
                const int idx1 = A.node_idx[k];
                if (!take1(idx1)) continue;
                q1.push_back(idx1);
                for (int m = B.node_off[b]; m < B.node_off[b + 1]; m++) idx.push_back(B.node_idx[m]);
                off.push_back((int)idx.size());
            }
            a++;
            b++;
        } else if (A.node_ids[a] < B.node_ids[b]) {
            while (a < A.n_nodes && A.node_ids[a] < B.node_ids[b]) a++;
        } else {
            while (b < B.n_nodes && B.node_ids[b] < A.node_ids[a]) b++;
        }
    }
}

// ORBmatcher.cc:656-808
int AMOS_VIEW_MATCHER::SearchByBoW(const amos_bow_view &KF1, const amos_bow_view &KF2, vector<int> &vnMatches12, const bool bBothKeyFrames)
{
    if (!bBothKeyFrames) return SearchByBoW(KF1, KF2, vnMatches12);
    vnMatches12.assign(KF1.n, -1);
    vector<bool> vbMatched2(KF2.n, false);
    vector<int> q1, off, idx;
    BowCandidates(KF1, KF2, [&](int i1) { return !KF1.has_point || KF1.has_point[i1]; }, q1, off, idx);
    const int nq = (int)q1.size();
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) memcpy(&qdesc[(size_t)i * 32], KF1.descriptors + (size_t)q1[i] * 32, 32);
    vector<uint16_t> dist;
    ListDistances(KF2.descriptors, KF2.n, qdesc.data(), nq, off, idx, dist);
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        const int idx1 = q1[i];
        int bestDist1 = 256;
        int bestIdx2 = -1;
        int bestDist2 = 256;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int idx2 = idx[k];
            if (vbMatched2[idx2] || (KF2.has_point && !KF2.has_point[idx2])) continue;  // :704-708
            const int d = dist[k];
            if (d < bestDist1) {
                bestDist2 = bestDist1;
                bestDist1 = d;
                bestIdx2 = idx2;
            } else if (d < bestDist2) {
                bestDist2 = d;
            }
        }
        if (bestDist1 < TH_LOW) {  // strict here, <= in SearchByBoW(KF, F)
            if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                vnMatches12[idx1] = bestIdx2;
                vbMatched2[bestIdx2] = true;
                if (mbCheckOrientation) {
                    float rot = KF1.keys[idx1].angle - KF2.keys[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(idx1);
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
                vnMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// ORBmatcher.cc:188-215 (no fused multiply-add: every product and sum rounds to float as written)
bool AMOS_VIEW_MATCHER::CheckDistEpipolarLine(const amos_keypoint &kp1, const amos_keypoint &kp2, const float F12[9], float sigma2_kp2)
{
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_kp2;
}

// ORBmatcher.cc:810-1018
int AMOS_VIEW_MATCHER::SearchForTriangulation(const amos_bow_view &KF1, const amos_bow_view &KF2, const float F12[9], float ex, float ey,
                                       const vector<float> &mvScaleFactors2, const vector<float> &mvLevelSigma2_2,
                                       vector<pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
{
    auto stereo = [](const amos_bow_view &v, int i) { return v.u_right ? v.u_right[i] >= 0 : false; };
    vector<int> q1, off, idx;
    BowCandidates(KF1, KF2,
                  [&](int i1) {
                      if (KF1.has_point && KF1.has_point[i1]) return false;  // pMP1 exists: nothing to triangulate (:846-849)
                      if (bOnlyStereo && !stereo(KF1, i1)) return false;
                      return true;
                  },
                  q1, off, idx);
    const int nq = (int)q1.size();
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) memcpy(&qdesc[(size_t)i * 32], KF1.descriptors + (size_t)q1[i] * 32, 32);
    vector<uint16_t> dist;
    ListDistances(KF2.descriptors, KF2.n, qdesc.data(), nq, off, idx, dist);
    int nmatches = 0;
    vector<bool> vbMatched2(KF2.n, false);
    vector<int> vMatches12(KF1.n, -1);
    vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < nq; i++) {
        const int idx1 = q1[i];
        const bool bStereo1 = stereo(KF1, idx1);
        const amos_keypoint &kp1 = KF1.keys[idx1];
        int bestDist = TH_LOW;
        int bestIdx2 = -1;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int idx2 = idx[k];
            if (vbMatched2[idx2] || (KF2.has_point && KF2.has_point[idx2])) continue;  // :866-869
            const bool bStereo2 = stereo(KF2, idx2);
            if (bOnlyStereo)
                if (!bStereo2) continue;
            const int d = dist[k];
            if (d > TH_LOW || d > bestDist) continue;
            const amos_keypoint &kp2 = KF2.keys[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2.x;
                const float distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * mvScaleFactors2[kp2.octave]) continue;
            }
            if (CheckDistEpipolarLine(kp1, kp2, F12, mvLevelSigma2_2[kp2.octave])) {
                bestIdx2 = idx2;
                bestDist = d;
            }
        }
        if (bestIdx2 >= 0) {
            const amos_keypoint &kp2 = KF2.keys[bestIdx2];
            vMatches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = true;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = kp1.angle - kp2.angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(idx1);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    vMatchedPairs.clear();
    vMatchedPairs.reserve(nmatches);
    for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
        if (vMatches12[i] < 0) continue;
        vMatchedPairs.push_back(make_pair(i, (size_t)vMatches12[i]));
    }
    return nmatches;
}

// The candidate loop shared by Fuse (:1086-1127, :1252-1272), SearchByProjection(pKF, Scw, ...) (:466-494) and
// SearchBySim3 (:1396-1420, :1476-1500): KeyFrame::GetFeaturesInArea(u, v, radius) has no level filter; the level
// gate  kpLevel < nPredictedLevel-1 || kpLevel > nPredictedLevel  and Fuse's reprojection gate follow per candidate.
void AMOS_VIEW_MATCHER::WindowCandidates(const FeatureGrid &KF, const vector<amos_window_query> &q, const vector<float> &mvScaleFactors, const float th,
                                  const vector<float> *mvInvLevelSigma2, vector<int> &off, vector<int> &idx, vector<uint16_t> &dist)
{
    const amos_frame_view &F = KF.Frame();
    const int nq = (int)q.size();
    off.assign(nq + 1, 0);
    idx.clear();
    vector<uint8_t> qdesc((size_t)nq * 32);
    for (int i = 0; i < nq; i++) {
        const amos_window_query &p = q[i];
        memcpy(&qdesc[(size_t)i * 32], p.desc, 32);
        const int nPredictedLevel = p.level;
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const vector<size_t> vIndices = KF.GetFeaturesInArea(p.u, p.v, radius);
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t i2 = vIndices[k];
            const amos_keypoint &kp = F.keys_un[i2];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (mvInvLevelSigma2) {
                if (F.u_right && F.u_right[i2] >= 0) {  // :1102-1116
                    const float ex = p.u - kp.x;
                    const float ey = p.v - kp.y;
                    const float er = p.ur - F.u_right[i2];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * (*mvInvLevelSigma2)[kpLevel] > 7.8) continue;
                } else {
                    const float ex = p.u - kp.x;
                    const float ey = p.v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * (*mvInvLevelSigma2)[kpLevel] > 5.99) continue;
                }
            }
            idx.push_back((int)i2);
        }
        off[i + 1] = (int)idx.size();
    }
    ListDistances(F, qdesc.data(), nq, off, idx, dist);
}

static int BestOfList(const vector<int> &off, const vector<int> &idx, const vector<uint16_t> &dist, int i, int &bestDist)
{
    bestDist = 256;
    int bestIdx = -1;
    for (int k = off[i]; k < off[i + 1]; k++)
        if (dist[k] < bestDist) {
            bestDist = dist[k];
            bestIdx = idx[k];
        }
    return bestIdx;
}

// ORBmatcher.cc:1020-1177
int AMOS_VIEW_MATCHER::Fuse(const FeatureGrid &KF, const vector<amos_window_query> &vpMapPoints, const vector<float> &mvScaleFactors,
                     const vector<float> &mvInvLevelSigma2, const float th, vector<int> &vnBestIdx)
{
    vector<int> off, idx;
    vector<uint16_t> dist;
    WindowCandidates(KF, vpMapPoints, mvScaleFactors, th, &mvInvLevelSigma2, off, idx, dist);
    int nFused = 0;
    vnBestIdx.assign(vpMapPoints.size(), -1);
    for (int i = 0; i < (int)vpMapPoints.size(); i++) {
        int bestDist;
        const int bestIdx = BestOfList(off, idx, dist, i, bestDist);
        if (bestDist <= TH_LOW) {
            vnBestIdx[i] = bestIdx;
            nFused++;
        }
    }
    return nFused;
}

// ORBmatcher.cc:1179-1312
int AMOS_VIEW_MATCHER::Fuse(const FeatureGrid &KF, const vector<amos_window_query> &vpPoints, const vector<float> &mvScaleFactors, const float th,
                     vector<int> &vnBestIdx)
{
    vector<int> off, idx;
    vector<uint16_t> dist;
    WindowCandidates(KF, vpPoints, mvScaleFactors, th, nullptr, off, idx, dist);
    int nFused = 0;
    vnBestIdx.assign(vpPoints.size(), -1);
    for (int i = 0; i < (int)vpPoints.size(); i++) {
        int bestDist;
        const int bestIdx = BestOfList(off, idx, dist, i, bestDist);
        if (bestDist <= TH_LOW) {
            vnBestIdx[i] = bestIdx;
            nFused++;
        }
    }
    return nFused;
}

// ORBmatcher.cc:388-512
int AMOS_VIEW_MATCHER::SearchByProjection(const FeatureGrid &KF, const vector<amos_window_query> &vpPoints, vector<int> &vnMatched,
                                   const vector<float> &mvScaleFactors, const int th)
{
    vector<int> off, idx;
    vector<uint16_t> dist;
    WindowCandidates(KF, vpPoints, mvScaleFactors, (float)th, nullptr, off, idx, dist);
    int nmatches = 0;
    for (int i = 0; i < (int)vpPoints.size(); i++) {
        int bestDist = 256;
        int bestIdx = -1;
        for (int k = off[i]; k < off[i + 1]; k++) {
            if (vnMatched[idx[k]] != AMOS_MATCH_FREE) continue;  // :470-471
            if (dist[k] < bestDist) {
                bestDist = dist[k];
                bestIdx = idx[k];
            }
        }
        if (bestDist <= TH_LOW) {
            vnMatched[bestIdx] = i;
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher.cc:1314-1565
int AMOS_VIEW_MATCHER::SearchBySim3(const FeatureGrid &KF1, const FeatureGrid &KF2, const vector<amos_window_query> &v1in2,
                             const vector<amos_window_query> &v2in1, const vector<float> &mvScaleFactors1, const vector<float> &mvScaleFactors2,
                             vector<int> &vnMatches12, const float th)
{
    const int N1 = KF1.Frame().n, N2 = KF2.Frame().n;
    vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    vector<int> off, idx;
    vector<uint16_t> dist;
    WindowCandidates(KF2, v1in2, mvScaleFactors2, th, nullptr, off, idx, dist);  // KF1's points searched in KF2
    for (int i = 0; i < (int)v1in2.size(); i++) {
        int bestDist;
        const int bestIdx = BestOfList(off, idx, dist, i, bestDist);
        if (bestDist <= TH_HIGH) vnMatch1[v1in2[i].src] = bestIdx;
    }
    WindowCandidates(KF1, v2in1, mvScaleFactors1, th, nullptr, off, idx, dist);  // KF2's points searched in KF1
    for (int i = 0; i < (int)v2in1.size(); i++) {
        int bestDist;
        const int bestIdx = BestOfList(off, idx, dist, i, bestDist);
        if (bestDist <= TH_HIGH) vnMatch2[v2in1[i].src] = bestIdx;
    }
    vnMatches12.assign(N1, -1);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {  // :1540-1556: keep what both directions agree on
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                vnMatches12[i1] = idx2;
                nFound++;
            }
        }
    }
    return nFound;
}

// ORBmatcher.cc:515-643
int AMOS_VIEW_MATCHER::SearchForInitialization(const amos_frame_view &F1, const FeatureGrid &F2g, vector<cv::Point2f> &vbPrevMatched,
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
void AMOS_VIEW_MATCHER::ComputeThreeMaxima(vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
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
