// amos_match.hip -- 256-bit Hamming distance and the best / second-best reductions that the
// inner loops of ORBmatcher::Search* perform (ORBmatcher.cc:1913-1933 and e.g. :127-148).
//
// Integer work: xor + v_bcnt (popcount) per 32-bit word, keys ordered by (distance, candidate
// position) so that ties resolve exactly as the reference's sequential
//     if(dist<bestDist){...} else if(dist<bestDist2){...}
// loop does (first candidate wins; see DESIGN.md for the proof of equivalence).  No MFMA.
#include "amos_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace amos {

struct Desc {
    uint32_t w[8];
};

__device__ __forceinline__ int hamming256(const Desc &a, const Desc &b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a.w[i] ^ b.w[i]);
    return d;
}

__device__ __forceinline__ Desc load_desc(const uint8_t *p)
{
    Desc d;
    const uint4 lo = reinterpret_cast<const uint4 *>(p)[0], hi = reinterpret_cast<const uint4 *>(p)[1];
    d.w[0] = lo.x; d.w[1] = lo.y; d.w[2] = lo.z; d.w[3] = lo.w;
    d.w[4] = hi.x; d.w[5] = hi.y; d.w[6] = hi.z; d.w[7] = hi.w;
    return d;
}

// popcount(x) + acc in ONE instruction.  The compiler knows v_bcnt_u32_b32's accumulate operand but
// re-associates an 8-term sum into 8 x v_bcnt(.., 0) + 3 x v_add3; the chained form is 8 instructions.
__device__ __forceinline__ int bcnt_acc(uint32_t x, int acc)
{
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// top-2 of unique keys
template <typename K>
__device__ __forceinline__ void top2_push(K &best, K &second, K key)
{
    const K hi = best > key ? best : key;
    best = best < key ? best : key;
    second = second < hi ? second : hi;
}
template <typename K>
__device__ __forceinline__ void top2_merge(K &best, K &second, K ob, K os)
{
    const K hi = best > ob ? best : ob;
    const K lo2 = second < os ? second : os;
    best = best < ob ? best : ob;
    second = hi < lo2 ? hi : lo2;
}

// ---- dense distances: out[i * nt + j].  grid = (ceil(nt/256), ceil(nq/16)), block = 256.
__global__ __launch_bounds__(256) void k_dist_dense(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                   int nt, uint16_t *__restrict__ out)
{
    __shared__ Desc qs[16];
    const int tid = threadIdx.x;
    const int q0 = blockIdx.y * 16;
    if (tid < 16 * 8) {
        const int qi = q0 + (tid >> 3);
        qs[tid >> 3].w[tid & 7] = qi < nq ? reinterpret_cast<const uint32_t *>(q)[(size_t)qi * 8 + (tid & 7)] : 0u;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j >= nt) return;
    const Desc td = load_desc(t + (size_t)j * 32);
    for (int k = 0; k < 16 && q0 + k < nq; k++) out[(size_t)(q0 + k) * nt + j] = (uint16_t)hamming256(qs[k], td);
}

// ---- candidate-list distances (CSR).  One wave per query; grid = ceil(nq/4), block = 256.
__global__ __launch_bounds__(256) void k_list_dist(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                  const int *__restrict__ candOff, const int *__restrict__ candIdx,
                                                  uint16_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    const Desc qd = load_desc(q + (size_t)i * 32);
    const int b = candOff[i], e = candOff[i + 1];
    for (int k = b + lane; k < e; k += 64) out[k] = (uint16_t)hamming256(qd, load_desc(t + (size_t)candIdx[k] * 32));
}

// ---- candidate-list best / second best.  One wave per query.
__global__ __launch_bounds__(256) void k_list_best2(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                   const int *__restrict__ candOff, const int *__restrict__ candIdx,
                                                   int initDist, amos_best2 *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    const Desc qd = load_desc(q + (size_t)i * 32);
    const int b = candOff[i], e = candOff[i + 1];
    const unsigned long long none = ~0ull;
    unsigned long long best = none, second = none;
    for (int k = b + lane; k < e; k += 64) {
        const int d = hamming256(qd, load_desc(t + (size_t)candIdx[k] * 32));
        if (d < initDist) top2_push(best, second, ((unsigned long long)d << 32) | (unsigned)(k - b));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
        top2_merge(best, second, ob, os);
    }
    if (lane == 0) {
        amos_best2 r;
        r.best_idx = best == none ? -1 : candIdx[b + (int)(best & 0xffffffffu)];
        r.best_dist = best == none ? initDist : (int)(best >> 32);
        r.second_idx = second == none ? -1 : candIdx[b + (int)(second & 0xffffffffu)];
        r.second_dist = second == none ? initDist : (int)(second >> 32);
        out[i] = r;
    }
}

// ---- Frame::AssignFeaturesToGrid (Frame.cc:431-461) for a batch: one work-group per frame turns the
// per-keypoint cell numbers (k_rgbd_glue) into CSR lists, ascending keypoint index inside a cell (the
// reference push_back()s in index order).  Counting by LDS atomics, exclusive scan of the 3072
// counts, unordered fill, then each thread insertion-sorts the (tiny) lists of its 12 cells.
constexpr int kGridCells = AMOS_FRAME_GRID_COLS * AMOS_FRAME_GRID_ROWS;

__global__ __launch_bounds__(256) void k_grid_build(const int *__restrict__ cell, const int *__restrict__ counts, int capacity,
                                                   int *__restrict__ cellStart, int *__restrict__ items)
{
    __shared__ int cnt[kGridCells], part[256];
    const int tid = threadIdx.x, frame = blockIdx.x;
    const int n = min(counts[frame], capacity);
    cell += (size_t)frame * capacity;
    items += (size_t)frame * capacity;
    cellStart += (size_t)frame * (kGridCells + 1);
    for (int c = tid; c < kGridCells; c += 256) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = cell[i];
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    constexpr int kPer = kGridCells / 256;  // 12 consecutive cells per thread
    int local[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) { local[k] = cnt[tid * kPer + k]; sum += local[k]; }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        cellStart[tid * kPer + k] = base;
        cnt[tid * kPer + k] = base;  // becomes the fill cursor
        base += local[k];
    }
    if (tid == 255) cellStart[kGridCells] = base;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = cell[i];
        if (c >= 0) items[atomicAdd(&cnt[c], 1)] = i;
    }
    __syncthreads();
    __threadfence_block();
    base = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        for (int a = base + 1; a < base + local[k]; a++) {
            const int v = items[a];
            int b = a - 1;
            while (b >= base && items[b] > v) { items[b + 1] = items[b]; b--; }
            items[b + 1] = v;
        }
        base += local[k];
    }
}

// ---- Frame::GetFeaturesInArea (Frame.cc:894-1003) + the best / second-best loop of
// ORBmatcher::SearchByProjection(F, LastF) (ORBmatcher.cc:1645-1690) for a batch of (query frame,
// train frame) pairs, everything resident.  One thread per query keypoint: window of radius
// th * scaleFactor[octave] around its (projected) position, cells x-major then y, items in insertion
// order, level gate by mode (0: octave-1..octave+1, 1 forward: >= octave, 2 backward: <= octave),
// optional right-coordinate gate; first candidate wins ties (strict <).  The greedy "already matched"
// skip of the reference stays with the caller: the unrestricted best is also the
// restricted best whenever it is free.
struct WindowArgs {
    const amos_keypoint *kps;   // [frames][capacity]
    const uint8_t *desc;        // [frames][capacity][32]
    const int *counts;
    const int *cellStart;       // [frames][3073]
    const int *items;           // [frames][capacity]
    const float *queryUv;       // [pairs][capacity][2] or null: the query keypoint's own position
    const float *queryInvZ;     // [pairs][capacity] or null
    const float *uRight;        // [frames][capacity] or null
    const int *pairsQ, *pairsT;
    float scale[AMOS_MAX_LEVELS];
    float th, mbf, minX, minY, wInv, hInv;
    int capacity, mode, initDist;
};

constexpr int kWindowLanes = 8;  // lanes per query: each takes every 8th grid column of the window

__global__ __launch_bounds__(256) void k_window_best2(const WindowArgs a, amos_best2 *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
    const int i = t / kWindowLanes, sub = t % kWindowLanes;
    const int fq = a.pairsQ[pair], ft = a.pairsT[pair];
    const int nq = min(a.counts[fq], a.capacity);
    const bool active = i < nq;
    const int iq = active ? i : 0;  // idle lanes shadow query 0 so that the group shuffles stay convergent
    const amos_keypoint qk = a.kps[(size_t)fq * a.capacity + iq];
    const Desc qd = load_desc(a.desc + ((size_t)fq * a.capacity + iq) * 32);
    const size_t po = (size_t)pair * a.capacity + iq;
    const float u = a.queryUv ? a.queryUv[2 * po] : qk.x, v = a.queryUv ? a.queryUv[2 * po + 1] : qk.y;
    const int oct = qk.octave;
    const float r = __fmul_rn(a.th, a.scale[oct]);
    const int minLevel = a.mode == 1 ? oct : a.mode == 2 ? 0 : oct - 1;
    const int maxLevel = a.mode == 1 ? -1 : a.mode == 2 ? oct : oct + 1;
    const bool checkLevels = minLevel > 0 || maxLevel >= 0;
    const bool gateRight = a.uRight != nullptr && a.queryInvZ != nullptr;
    const float ur = gateRight ? __fsub_rn(u, __fmul_rn(a.mbf, a.queryInvZ[po])) : 0.f;
    int x0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, a.minX), r), a.wInv));
    int x1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, a.minX), r), a.wInv));
    int y0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, a.minY), r), a.hInv));
    int y1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, a.minY), r), a.hInv));
    // Frame.cc:905-921: the reference returns early when a clamp range is empty; without this a query projected far
    // outside the bounds would index cellStart with y0 >= ROWS
    const bool empty = x0 >= AMOS_FRAME_GRID_COLS || x1 < 0 || y0 >= AMOS_FRAME_GRID_ROWS || y1 < 0;
    x0 = max(x0, 0); y0 = max(y0, 0);
    x1 = empty ? -1 : min(x1, AMOS_FRAME_GRID_COLS - 1); y1 = min(y1, AMOS_FRAME_GRID_ROWS - 1);
    y0 = min(y0, AMOS_FRAME_GRID_ROWS - 1);
    const int *cs = a.cellStart + (size_t)ft * (kGridCells + 1);
    const int *it = a.items + (size_t)ft * a.capacity;
    const amos_keypoint *tk = a.kps + (size_t)ft * a.capacity;
    const uint8_t *td = a.desc + (size_t)ft * a.capacity * 32;
    const float *tr = gateRight ? a.uRight + (size_t)ft * a.capacity : nullptr;
    // The reference walks the cells x-major, then y, then insertion order = ascending CSR position, and its
    // strict-< updates keep the FIRST of equal distances: a min-reduction over keys (dist << 16 | CSR position)
    // gives the same best and second best whatever the evaluation order.
    unsigned best = 0xffffffffu, second = 0xffffffffu;
    for (int ix = x0 + sub; ix <= x1; ix += kWindowLanes) {
        // cells (ix, y0..y1) are consecutive in the CSR: one item range per column
        const int b = cs[ix * AMOS_FRAME_GRID_ROWS + y0], e = y1 >= y0 ? cs[ix * AMOS_FRAME_GRID_ROWS + y1 + 1] : b;
        for (int j = b; j < e; j++) {
            const int idx = it[j];
            const amos_keypoint k = tk[idx];
            if (checkLevels && (k.octave < minLevel || (maxLevel >= 0 && k.octave > maxLevel))) continue;
            if (!(fabsf(__fsub_rn(k.x, u)) < r && fabsf(__fsub_rn(k.y, v)) < r)) continue;
            if (gateRight) {
                const float tt = tr[idx];
                if (tt > 0 && fabsf(__fsub_rn(ur, tt)) > r) continue;
            }
            const int d = hamming256(qd, load_desc(td + (size_t)idx * 32));
            if (d < a.initDist) top2_push(best, second, ((unsigned)d << 16) | (unsigned)j);
        }
    }
#pragma unroll
    for (int off = kWindowLanes / 2; off > 0; off >>= 1) {
        const unsigned ob = __shfl_xor(best, off, kWindowLanes), os = __shfl_xor(second, off, kWindowLanes);
        top2_merge(best, second, ob, os);
    }
    if (active && sub == 0) {
        amos_best2 res;
        res.best_idx = best == 0xffffffffu ? -1 : it[best & 0xffffu];
        res.best_dist = best == 0xffffffffu ? a.initDist : (int)(best >> 16);
        res.second_idx = second == 0xffffffffu ? -1 : it[second & 0xffffu];
        res.second_dist = second == 0xffffffffu ? a.initDist : (int)(second >> 16);
        out[po] = res;
    }
}

// ---- brute-force best / second best over ALL train descriptors, for a batch of (query set,
// train set) pairs.  Block = 64 queries x 4 waves; wave w scans the w-th quarter of the train set
// from LDS tiles (every lane reads the same descriptor: LDS broadcast), the four partial top-2s are
// merged through LDS.  grid = (ceil(maxQueries/64), nPairs).

template <bool kGate>
__global__ __launch_bounds__(256) void k_bf_best2(const uint8_t *__restrict__ descBaseQ, const uint8_t *__restrict__ descBaseT,
                                                 size_t frameStrideBytes, const int *__restrict__ counts,
                                                 const int *__restrict__ pairsQ, const int *__restrict__ pairsT,
                                                 int nqFixed, int ntFixed, int capacity, int initDist,
                                                 amos_best2 *__restrict__ out)
{
    __shared__ unsigned mergeB[4][64], mergeS[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar loop bounds
    const int pair = blockIdx.y;
    const int fq = pairsQ ? pairsQ[pair] : 0, ft = pairsT ? pairsT[pair] : 0;
    const int nq = counts ? min(counts[fq], capacity) : nqFixed;
    const int nt = counts ? min(counts[ft], capacity) : ntFixed;
    const int qi = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= nq) return;  // whole block idle (uniform)
    const uint8_t *qb = descBaseQ + (size_t)fq * frameStrideBytes;
    const uint8_t *tb = descBaseT + (size_t)ft * frameStrideBytes;
    Desc qd;
    if (qi < nq) qd = load_desc(qb + (size_t)qi * 32);
    else
        for (int k = 0; k < 8; k++) qd.w[k] = 0;
    const int per = (nt + 3) >> 2;  // contiguous quarter per wave
    const int j0 = wave * per, j1 = min(j0 + per, nt);
    unsigned best = 0xffffffffu, second = 0xffffffffu;
    // Every lane of the wave compares against the SAME train descriptor: its address is wave-uniform, so the
    // 32 bytes arrive through the scalar data cache (s_load_dwordx8) straight into SGPR operands of the
    // v_xor -- no LDS tile, no vector registers for the train side.
    const uint4 *tv = reinterpret_cast<const uint4 *>(tb);
    // key = dist << 16 | train index.  The init_dist gate ("only dist < init_dist can win") is applied
    // once at the end: dropping every key >= init_dist << 16 from the ungated top-2 gives the gated top-2.
    auto key_of = [&](const uint4 &lo, const uint4 &hi, int jj) -> unsigned {
        int d = __popc(qd.w[0] ^ lo.x);
        d = bcnt_acc(qd.w[1] ^ lo.y, d);
        d = bcnt_acc(qd.w[2] ^ lo.z, d);
        d = bcnt_acc(qd.w[3] ^ lo.w, d);
        d = bcnt_acc(qd.w[4] ^ hi.x, d);
        d = bcnt_acc(qd.w[5] ^ hi.y, d);
        d = bcnt_acc(qd.w[6] ^ hi.z, d);
        d = bcnt_acc(qd.w[7] ^ hi.w, d);
        return ((unsigned)d << 16) | (unsigned)jj;
    };
    // two new keys against (best <= second) in three instructions: the smallest of {best, k1, k2} is the new
    // best, the median of the three or the old second (whichever is smaller) the new second
    auto push2 = [&](unsigned k1, unsigned k2) {
        unsigned mid;
        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(mid) : "v"(best), "v"(k1), "v"(k2));
        best = min(best, min(k1, k2));  // v_min3_u32
        second = min(second, mid);
    };
    int jj = j0;
    for (; jj + 4 <= j1; jj += 4) {  // four descriptors (eight scalar loads) in flight per trip
        const uint4 a0 = tv[2 * jj], a1 = tv[2 * jj + 1], b0 = tv[2 * jj + 2], b1 = tv[2 * jj + 3];
        const uint4 c0 = tv[2 * jj + 4], c1 = tv[2 * jj + 5], d0 = tv[2 * jj + 6], d1 = tv[2 * jj + 7];
        push2(key_of(a0, a1, jj), key_of(b0, b1, jj + 1));
        push2(key_of(c0, c1, jj + 2), key_of(d0, d1, jj + 3));
    }
    for (; jj < j1; jj++) top2_push(best, second, key_of(tv[2 * jj], tv[2 * jj + 1], jj));
    mergeB[wave][lane] = best;
    mergeS[wave][lane] = second;
    __syncthreads();
    if (wave == 0 && qi < nq) {
#pragma unroll
        for (int w = 1; w < 4; w++) top2_merge(best, second, mergeB[w][lane], mergeS[w][lane]);
        if (kGate) {
            const unsigned limit = (unsigned)max(initDist, 0) << 16;  // keys of distances >= init_dist
            if (best >= limit) best = 0xffffffffu;
            if (second >= limit) second = 0xffffffffu;
        }
        amos_best2 r;
        r.best_idx = best == 0xffffffffu ? -1 : (int)(best & 0xffff);
        r.best_dist = best == 0xffffffffu ? initDist : (int)(best >> 16);
        r.second_idx = second == 0xffffffffu ? -1 : (int)(second & 0xffff);
        r.second_dist = second == 0xffffffffu ? initDist : (int)(second >> 16);
        out[(size_t)pair * capacity + qi] = r;
    }
}

// ---- the same reduction on the matrix cores (exact integer arithmetic).
// For bit vectors  ham(q, t) = |q| + sum_k t_k * (1 - 2 q_k):  with the train bits as 0/1 bytes and the query bits
// as -+1 bytes (set bit = -1) the sum is an i8 GEMM over K = 256 (v_mfma_i32_32x32x32_i8, i32 accumulate) and |q|
// is a per-query constant, so the ORDER of the candidates of one query is the order of acc = sum and the key
//     ((512 + acc) << 16) | train index
// min-reduces exactly like (dist << 16 | index) of the popcount kernel (first candidate wins ties).  Against
// k_bf_best2's 18.5 vector instructions per 64 pairs (16 of them xor + popcount) this needs about 4: 2 per
// operand dword to spread bits into bytes ((x >> r) & 0x01010101, amortised over the four query tiles of a wave),
// 1 per pair for the key (v_lshl_add_u32 on the accumulator) and 1.5 per pair for the running best two.
//
// Layout: block = 128 queries x 4 waves; a wave holds the B operand (queries, +-1) of four 32-query tiles in
// registers for its whole life (128 VGPRs) and walks the train tiles wave, wave + 4, ...; per tile 32 MFMAs
// (4 query tiles x K = 256 in 8 steps).  The k index is permuted the same way on both operands (lane half h owns
// descriptor dwords 4h .. 4h + 3; register r of a step holds bits r, r + 8, r + 16, r + 24 of a dword), which a
// dot product does not see.  C/D: lane = query column, the 16 accumulators = train rows
// (i & 3) + 8 (i >> 2) + 4 h  (ascending in i: ascending train index).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// kWaves = waves per block = ways the train tiles of one 128-query block are split (tile = wave, wave + kWaves, ...).
// One wave per block amortises the query-side setup over every train tile (batched launches: enough blocks to fill
// the chip anyway); four waves shorten the critical path of a single pair.
//
// The wave's loop is software-pipelined in half tiles so that the matrix cores and the vector units work at the same
// time (two co-resident waves running the same phases in step do not overlap): while the 16 MFMAs of query tiles
// 2, 3 run, the best-two update of tiles 0, 1 and the bit spreading of the NEXT train tile issue between them, then
// the 16 MFMAs of tiles 0, 1 of the next train tile run over the update of tiles 2, 3.
// kQ = independent 128-query groups per block (each with its own kWaves waves): single-wave work-groups are only ever placed
// on half of a CU's SIMDs (measured: 1 024 one-wave blocks run in two rounds of 512), multi-wave blocks use all four.
template <bool kGate, int kWaves, int kQ>
__global__ __launch_bounds__(64 * kWaves * kQ) void k_bf_best2_mfma(const uint8_t *__restrict__ descBaseQ, const uint8_t *__restrict__ descBaseT,
                                                             size_t frameStrideBytes, const int *__restrict__ counts,
                                                             const int *__restrict__ pairsQ, const int *__restrict__ pairsT,
                                                             int nqFixed, int ntFixed, int capacity, int initDist,
                                                             amos_best2 *__restrict__ out)
{
    __shared__ unsigned mBq[kQ][2 * kWaves][128], mSq[kQ][2 * kWaves][128];
    const int lane = threadIdx.x & 63, waveAll = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = waveAll / kWaves, wave = waveAll - qg * kWaves, tid = threadIdx.x - qg * 64 * kWaves;  // query group, train-split wave, thread of the group
    unsigned (*mB)[128] = mBq[qg], (*mS)[128] = mSq[qg];
    const int r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.y;
    const int fq = pairsQ ? pairsQ[pair] : 0, ft = pairsT ? pairsT[pair] : 0;
    const int nq = counts ? min(counts[fq], capacity) : nqFixed;
    const int nt = counts ? min(counts[ft], capacity) : ntFixed;
    const int q0 = (blockIdx.x * kQ + qg) * 128;
    if (blockIdx.x * kQ * 128 >= nq) return;  // whole block idle (uniform)
    const bool groupActive = q0 < nq;          // wave-uniform; idle groups of a live block only keep the barrier company
    const uint8_t *qb = descBaseQ + (size_t)fq * frameStrideBytes;
    const uint8_t *tb = descBaseT + (size_t)ft * frameStrideBytes;
    constexpr unsigned M = 0x01010101u;
    if (groupActive) {
    // B operand: -+1 bytes of the lane's four queries (zero for queries beyond nq)
    v4i Bq[4][4][2];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int q = q0 + 32 * c + r;
        uint4 y4 = uint4{0, 0, 0, 0};
        if (q < nq) y4 = *reinterpret_cast<const uint4 *>(qb + (size_t)q * 32 + 16 * h);
        const unsigned y[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                const unsigned m = (y[d] >> rr) & M;                 // 1 where the bit is set
                const unsigned v = ((m << 8) - m) | (m ^ M);         // bit set: 0xff = -1, clear: 0x01 = +1 (the accumulator is -dot)
                Bq[c][d][rr >> 2][rr & 3] = q < nq ? (int)v : 0;
            }
    }
    unsigned best[4], second[4];
#pragma unroll
    for (int c = 0; c < 4; c++) best[c] = second[c] = 0xffffffffu;
    // (512 << 16) + train index of accumulator i in the wave's current tile, WITHOUT the lane half's 4 h: every key of
    // a lane carries the same 4 h, so it is added once at the end -- which makes these offsets wave-uniform (SGPRs,
    // bumped by scalar adds) and the key one v_lshl_add_u32 with a scalar operand.
    int idxoff[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        idxoff[i] = (512 << 16) + wave * 32 + (i & 3) + 8 * (i >> 2);
        asm volatile("" : "+s"(idxoff[i]));
    }
    auto push2 = [](unsigned &b, unsigned &s, unsigned k1, unsigned k2) {
        unsigned mid;
        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(mid) : "v"(b), "v"(k1), "v"(k2));
        b = min(b, min(k1, k2));  // v_min3_u32
        s = min(s, mid);
    };
    auto spread1 = [&](const unsigned x, v4i (&Ad)[2]) {  // 0/1 bytes of one descriptor dword: register r' holds bits r', r' + 8, ...
        Ad[0] = v4i{(int)(x & M), (int)((x >> 1) & M), (int)((x >> 2) & M), (int)((x >> 3) & M)};
        Ad[1] = v4i{(int)((x >> 4) & M), (int)((x >> 5) & M), (int)((x >> 6) & M), (int)((x >> 7) & M)};
    };
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto mfma_step = [&](const v4i (&Ad)[2], const int c0, const int d, v16i &accA, v16i &accB) {  // query tiles c0, c0 + 1, k-steps 2 d, 2 d + 1
#pragma unroll
        for (int k = 0; k < 2; k++) {
            accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ad[k], Bq[c0][d][k], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ad[k], Bq[c0 + 1][d][k], accB, 0, 0, 0);
        }
    };
    auto update = [&](const v16i &acc, unsigned &b, unsigned &s) {  // 16 keys of one query into its running best two
#pragma unroll
        for (int i = 0; i < 16; i += 2)
            push2(b, s, ((unsigned)acc[i] << 16) + (unsigned)idxoff[i], ((unsigned)acc[i + 1] << 16) + (unsigned)idxoff[i + 1]);  // v_lshl_add_u32
    };
    auto load_tile = [&](int tile) { return *reinterpret_cast<const uint4 *>(tb + (size_t)(tile * 32 + r) * 32 + 16 * h); };

    const int nFull = nt >> 5;                       // tiles whose 32 rows are all train descriptors
    const int nMine = nFull > wave ? (nFull - wave + kWaves - 1) / kWaves : 0;
    if (nMine > 0) {
        // The loop is software-pipelined in half tiles: phase P multiplies query tiles 2, 3 of the current train tile
        // while the best-two update of tiles 0, 1 issues between the MFMAs; phase Q spreads the NEXT train tile dword by
        // dword into the same operand registers (the current tile's are dead by then), multiplies its query tiles 0, 1
        // and updates tiles 2, 3 of the current one.  One operand buffer: under 256 registers, two waves per SIMD.
        v4i A[4][2];
        v16i acc0 = zero16, acc1 = zero16, acc2, acc3;
        {
            const uint4 a4 = load_tile(wave);
            const unsigned a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                spread1(a[d], A[d]);
                mfma_step(A[d], 0, d, acc0, acc1);
            }
        }
        for (int it = 0; it < nMine; it++) {
            // the next tile (the last trip re-reads its own: those MFMAs are never consumed), used in phase Q
            const uint4 n4 = load_tile(wave + min(it + 1, nMine - 1) * kWaves);
            const unsigned nx[4] = {n4.x, n4.y, n4.z, n4.w};
            acc2 = zero16;
            acc3 = zero16;
#pragma unroll
            for (int d = 0; d < 4; d++) mfma_step(A[d], 2, d, acc2, acc3);
            update(acc0, best[0], second[0]);
            update(acc1, best[1], second[1]);
#pragma unroll
            for (int k = 0; k < 16; k++) {  // phase P: 1 MFMA : 5 vector instructions
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
            acc0 = zero16;
            acc1 = zero16;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                spread1(nx[d], A[d]);
                mfma_step(A[d], 0, d, acc0, acc1);
            }
            update(acc2, best[2], second[2]);
            update(acc3, best[3], second[3]);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                idxoff[i] += 32 * kWaves;
                asm volatile("" : "+s"(idxoff[i]));  // sixteen live SGPRs, not expressions of the trip count folded into every key
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {  // phase Q: 1 MFMA : 9 vector instructions
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        }
    }
    if ((nt & 31) && (nFull % kWaves) == wave) {  // the partial last tile: rows beyond nt never win
        const int t = nFull * 32 + r;
        uint4 a4 = uint4{0, 0, 0, 0};
        if (t < nt) a4 = *reinterpret_cast<const uint4 *>(tb + (size_t)t * 32 + 16 * h);
        const unsigned a[4] = {a4.x, a4.y, a4.z, a4.w};
        v4i A[4][2];
        v16i acc[4] = {zero16, zero16, zero16, zero16};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            spread1(a[d], A[d]);
            mfma_step(A[d], 0, d, acc[0], acc[1]);
            mfma_step(A[d], 2, d, acc[2], acc[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int row0 = nFull * 32 + (i & 3) + 8 * (i >> 2), row1 = row0 + 1;  // accumulators i, i + 1 (without 4 h)
                unsigned k1 = ((unsigned)acc[c][i] << 16) + (unsigned)((512 << 16) + row0);
                unsigned k2 = ((unsigned)acc[c][i + 1] << 16) + (unsigned)((512 << 16) + row1);
                if (row0 + 4 * h >= nt) k1 = 0xffffffffu;
                if (row1 + 4 * h >= nt) k2 = 0xffffffffu;
                push2(best[c], second[c], k1, k2);
            }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {  // the lane half's share of the train index
        if (best[c] != 0xffffffffu) best[c] += 4 * h;
        if (second[c] != 0xffffffffu) second[c] += 4 * h;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        mB[wave * 2 + h][32 * c + r] = best[c];
        mS[wave * 2 + h][32 * c + r] = second[c];
    }
    }  // groupActive
    __syncthreads();
    if (!groupActive) return;
    for (int qq = tid; qq < 128; qq += 64 * kWaves) {
        const int qi = q0 + qq;
        if (qi >= nq) break;
        unsigned b = mB[0][qq], s2 = mS[0][qq];
#pragma unroll
        for (int w = 1; w < 2 * kWaves; w++) top2_merge(b, s2, mB[w][qq], mS[w][qq]);
        const Desc qd = load_desc(qb + (size_t)qi * 32);
        int pq = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) pq += __popc(qd.w[k]);
        // key >> 16 = 512 + acc and ham = |q| + acc
        int bd = (int)(b >> 16) - 512 + pq, sd = (int)(s2 >> 16) - 512 + pq;
        bool hb = b != 0xffffffffu, hs = s2 != 0xffffffffu;
        if (kGate) {
            hb = hb && bd < initDist;
            hs = hs && sd < initDist;
        }
        amos_best2 res;
        res.best_idx = hb ? (int)(b & 0xffff) : -1;
        res.best_dist = hb ? bd : initDist;
        res.second_idx = hs ? (int)(s2 & 0xffff) : -1;
        res.second_dist = hs ? sd : initDist;
        out[(size_t)pair * capacity + qi] = res;
    }
}

}  // namespace amos

using namespace amos;

struct amos_match {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // device scratch of the host-pointer entry points: the inputs of the call in flight (pointers into dArena), the grow-only result buffer
    uint8_t *dQ = nullptr, *dT = nullptr;
    int *dOff = nullptr, *dIdx = nullptr;
    void *dOut = nullptr;
    size_t capOut = 0;
    // pinned host staging of the host-buffer calls: the caller's (pageable) arrays are copied here and travel as true asynchronous DMA
    // transfers; results land here behind the kernel and are copied out after the ONE synchronisation of the call (a hipMemcpyAsync on
    // pageable memory is a blocking staged copy of its own: six of them were most of a 0.26 ms list-distance call)
    uint8_t *hStage = nullptr;
    uint8_t *dArena = nullptr;  // device mirror of the staging buffer's input part: the inputs of a call travel as ONE transfer (dQ / dT / dOff / dIdx point into it)
    size_t capStage = 0, stageUsed = 0;
    int bfKernel = 0;  // brute-force best-2: 0 = choose by size, 1 = xor + popcount kernel, 2 = i8 MFMA kernel
};

// the MFMA kernel works on 128-query x 32-train tiles: below a few tiles' worth of work the popcount kernel wins
static bool use_mfma(const amos_match *m, int nq, int nt)
{
    if (m->bfKernel == 1) return false;
    if (m->bfKernel == 2) return true;
    return nq >= 96 && nt >= 64;
}

template <bool kGate>
static void launch_bf(amos_match *m, bool mfma, dim3 gridPop, dim3 gridMfma, const uint8_t *dq, const uint8_t *dt, size_t stride, const int *counts,
                      const int *pq, const int *pt, int nq, int nt, int capacity, int initDist, amos_best2 *out)
{
    static const int forced = getenv("AMOS_MM_WAVES") ? atoi(getenv("AMOS_MM_WAVES")) : 0;  // experiment switch
    // experiment switch: AMOS_MM_WAVES = 10 * (query groups per block) + (train-split waves per group), e.g. 21, 41, 14
    const int waves = forced ? forced : ((size_t)gridMfma.x * gridMfma.y >= 512 ? 21 : 14);
    const int cap128 = (int)gridMfma.x;  // 128-query groups per pair
    if (mfma && waves == 21)   // batched launches: two independent one-wave groups per block
        hipLaunchKernelGGL((k_bf_best2_mfma<kGate, 1, 2>), dim3((cap128 + 1) / 2, gridMfma.y), dim3(128), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
    else if (mfma && waves == 41)
        hipLaunchKernelGGL((k_bf_best2_mfma<kGate, 1, 4>), dim3((cap128 + 3) / 4, gridMfma.y), dim3(256), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
    else if (mfma && waves == 11)
        hipLaunchKernelGGL((k_bf_best2_mfma<kGate, 1, 1>), gridMfma, dim3(64), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
    else if (mfma && waves == 12)
        hipLaunchKernelGGL((k_bf_best2_mfma<kGate, 2, 1>), gridMfma, dim3(128), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
    else if (mfma)             // a single pair: four waves split the train tiles of each 128-query group
        hipLaunchKernelGGL((k_bf_best2_mfma<kGate, 4, 1>), gridMfma, dim3(256), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
    else
        hipLaunchKernelGGL(k_bf_best2<kGate>, gridPop, dim3(256), 0, m->stream, dq, dt, stride, counts, pq, pt, nq, nt, capacity, initDist, out);
}

template <typename T>
static int grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return AMOS_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t n = std::max<size_t>(need + need / 2, 256);
    AMOS_HIP_CHECK(hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return AMOS_OK;
}

// every host-buffer call ends with a stream synchronisation, so the staging buffer is free at the start of the next one
static int stage_begin(amos_match *m, size_t bytes)
{
    m->stageUsed = 0;
    bytes += 1024;  // alignment slack of the pieces
    if (bytes <= m->capStage) return AMOS_OK;
    if (m->hStage || m->dArena) (void)hipStreamSynchronize(m->stream);  // (a call that failed half way may have left a transfer in flight)
    if (m->hStage) (void)hipHostFree(m->hStage);
    if (m->dArena) (void)hipFree(m->dArena);
    m->hStage = m->dArena = nullptr;
    m->capStage = 0;
    const size_t n = std::max<size_t>(bytes + bytes / 2, 1 << 16);
    if (hipHostMalloc((void **)&m->hStage, n, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&m->dArena, n) != hipSuccess) {
        (void)hipGetLastError();
        set_error("matcher staging of %zu bytes (pinned host + device) could not be allocated", n);
        return AMOS_ERR_DEVICE;
    }
    m->capStage = n;
    return AMOS_OK;
}

static size_t stage_take(amos_match *m, size_t bytes)
{
    const size_t o = m->stageUsed;
    m->stageUsed += (bytes + 63) & ~(size_t)63;
    return o;  // (stage_begin sized the buffers for the sum of the call's pieces)
}

// host array -> staging; returns where it will sit on the device once stage_flush has run
template <typename T>
static T *stage_input(amos_match *m, const void *src, size_t bytes)
{
    const size_t o = stage_take(m, bytes);
    if (bytes) std::memcpy(m->hStage + o, src, bytes);
    return reinterpret_cast<T *>(m->dArena + o);
}

static int stage_flush(amos_match *m)
{
    if (m->stageUsed) AMOS_HIP_CHECK(hipMemcpyAsync(m->dArena, m->hStage, m->stageUsed, hipMemcpyHostToDevice, m->stream));
    return AMOS_OK;
}

// device -> staging (behind the inputs), one synchronisation, staging -> `out`
static int stage_d2h_sync(amos_match *m, void *out, const void *src, size_t bytes)
{
    uint8_t *p = m->hStage + stage_take(m, bytes);
    AMOS_HIP_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, m->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    std::memcpy(out, p, bytes);
    return AMOS_OK;
}

static int upload_sets(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt)
{
    m->dQ = stage_input<uint8_t>(m, q, (size_t)nq * 32);
    m->dT = stage_input<uint8_t>(m, t, (size_t)nt * 32);
    return AMOS_OK;
}

static int upload_lists(amos_match *m, int nq, int nt, const int32_t *cand_off, const int32_t *cand_idx, int *total)
{
    if (cand_off[0] != 0) { set_error("cand_off[0] must be 0"); return AMOS_ERR_INVALID; }
    for (int i = 0; i < nq; i++)
        if (cand_off[i + 1] < cand_off[i]) { set_error("cand_off not monotonic at %d", i); return AMOS_ERR_INVALID; }
    const int n = cand_off[nq];
    for (int k = 0; k < n; k++)
        if (cand_idx[k] < 0 || cand_idx[k] >= nt) { set_error("cand_idx[%d] = %d outside [0,%d)", k, cand_idx[k], nt); return AMOS_ERR_INVALID; }
    m->dOff = stage_input<int>(m, cand_off, sizeof(int) * ((size_t)nq + 1));
    m->dIdx = stage_input<int>(m, cand_idx, sizeof(int) * (size_t)n);
    *total = n;
    return AMOS_OK;
}

static int grow_out(amos_match *m, size_t bytes)
{
    uint8_t *p = (uint8_t *)m->dOut;
    int rc = grow(&p, &m->capOut, bytes);
    m->dOut = p;
    return rc;
}

extern "C" {

int amos_match_create(int device, void *stream, amos_match **out)
{
    if (!out) { set_error("amos_match_create: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_match *m = new amos_match();
    m->device = device;
    if (stream) m->stream = (hipStream_t)stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete m; return AMOS_ERR_DEVICE; }
        m->ownStream = true;
    }
    *out = m;
    return AMOS_OK;
}

void amos_match_destroy(amos_match *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    void *ptrs[] = {m->dArena, m->dOut};  // (dQ / dT / dOff / dIdx point into the arena)
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (m->hStage) (void)hipHostFree(m->hStage);
    if (m->ownStream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int amos_match_sync(amos_match *m)
{
    if (!m) return AMOS_ERR_INVALID;
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    return AMOS_OK;
}

void *amos_match_stream(amos_match *m) { return m ? (void *)m->stream : nullptr; }

int amos_match_set_bruteforce_kernel(amos_match *m, int mode)
{
    if (!m || mode < 0 || mode > 2) { set_error("amos_match_set_bruteforce_kernel: mode 0 (by size), 1 (popcount) or 2 (MFMA)"); return AMOS_ERR_INVALID; }
    m->bfKernel = mode;
    return AMOS_OK;
}

int amos_match_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *out)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || (!out && nq > 0 && nt > 0)) { set_error("amos_match_distances: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0 || nt == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    const size_t bytes = (size_t)nq * nt * sizeof(uint16_t);
    int rc = stage_begin(m, ((size_t)nq + nt) * 32 + bytes);
    if (rc != AMOS_OK) return rc;
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, bytes);
    if (rc != AMOS_OK) return rc;
    rc = stage_flush(m);
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_dist_dense, dim3((nt + 255) / 256, (nq + 15) / 16), dim3(256), 0, m->stream, m->dQ, nq, m->dT, nt, (uint16_t *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    return stage_d2h_sync(m, out, m->dOut, bytes);
}

int amos_match_list_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *cand_off,
                              const int32_t *cand_idx, uint16_t *out)
{
    if (!m || nq < 0 || nt < 0 || !cand_off || (nq > 0 && !q)) { set_error("amos_match_list_distances: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int total = 0;
    const size_t nCand = cand_off[nq] > 0 ? (size_t)cand_off[nq] : 0;  // (validated by upload_lists)
    int rc = stage_begin(m, ((size_t)nq + nt) * 32 + ((size_t)nq + 1) * 4 + nCand * 6);
    if (rc != AMOS_OK) return rc;
    rc = upload_lists(m, nq, nt, cand_off, cand_idx, &total);
    if (rc != AMOS_OK) return rc;
    if (total == 0) return AMOS_OK;
    if (!t || !out || !cand_idx) { set_error("amos_match_list_distances: null buffer"); return AMOS_ERR_INVALID; }
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)total * sizeof(uint16_t));
    if (rc != AMOS_OK) return rc;
    rc = stage_flush(m);
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_list_dist, dim3((nq + 3) / 4), dim3(256), 0, m->stream, m->dQ, nq, m->dT, m->dOff, m->dIdx, (uint16_t *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    return stage_d2h_sync(m, out, m->dOut, (size_t)total * sizeof(uint16_t));
}

int amos_match_list_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *cand_off,
                          const int32_t *cand_idx, int init_dist, amos_best2 *out)
{
    if (!m || nq < 0 || nt < 0 || !cand_off || (nq > 0 && (!q || !out))) { set_error("amos_match_list_best2: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int total = 0;
    const size_t nCand = cand_off[nq] > 0 ? (size_t)cand_off[nq] : 0;
    int rc = stage_begin(m, ((size_t)nq + nt) * 32 + ((size_t)nq + 1) * 4 + nCand * 4 + (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    rc = upload_lists(m, nq, nt, cand_off, cand_idx, &total);
    if (rc != AMOS_OK) return rc;
    if (total > 0 && (!t || !cand_idx)) { set_error("amos_match_list_best2: null buffer"); return AMOS_ERR_INVALID; }
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    rc = stage_flush(m);
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_list_best2, dim3((nq + 3) / 4), dim3(256), 0, m->stream, m->dQ, nq, m->dT, m->dOff, m->dIdx, init_dist, (amos_best2 *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    return stage_d2h_sync(m, out, m->dOut, (size_t)nq * sizeof(amos_best2));
}

int amos_match_bruteforce_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int init_dist,
                                amos_best2 *out)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) { set_error("amos_match_bruteforce_best2: invalid argument"); return AMOS_ERR_INVALID; }
    if (nt > 65535) { set_error("amos_match_bruteforce_best2: at most 65535 train descriptors"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int rc = stage_begin(m, ((size_t)nq + nt) * 32 + (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    rc = stage_flush(m);
    if (rc != AMOS_OK) return rc;
    {
        const bool mfma = use_mfma(m, nq, nt);
        const dim3 gp((nq + 63) / 64, 1), gm((nq + 127) / 128, 1);
        if (init_dist > 256)
            launch_bf<false>(m, mfma, gp, gm, m->dQ, m->dT, 0, nullptr, nullptr, nullptr, nq, nt, nq, init_dist, (amos_best2 *)m->dOut);
        else
            launch_bf<true>(m, mfma, gp, gm, m->dQ, m->dT, 0, nullptr, nullptr, nullptr, nq, nt, nq, init_dist, (amos_best2 *)m->dOut);
    }
    AMOS_HIP_CHECK(hipGetLastError());
    return stage_d2h_sync(m, out, m->dOut, (size_t)nq * sizeof(amos_best2));
}

int amos_match_bruteforce_best2_batch_device(amos_match *m, const uint8_t *d_desc, size_t frame_stride_bytes,
                                             const int32_t *d_counts, const int32_t *d_pairs_q, const int32_t *d_pairs_t,
                                             int n_pairs, int capacity, int init_dist, amos_best2 *d_out)
{
    if (!m || !d_desc || !d_counts || !d_pairs_q || !d_pairs_t || !d_out || n_pairs < 1 || capacity < 1 || capacity > 65535) {
        set_error("amos_match_bruteforce_best2_batch_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    {
        const bool mfma = use_mfma(m, capacity, capacity);  // the counts live on the device: decide by the capacity
        const dim3 gp((capacity + 63) / 64, n_pairs), gm((capacity + 127) / 128, n_pairs);
        if (init_dist > 256)
            launch_bf<false>(m, mfma, gp, gm, d_desc, d_desc, frame_stride_bytes, d_counts, d_pairs_q, d_pairs_t, 0, 0, capacity, init_dist, d_out);
        else
            launch_bf<true>(m, mfma, gp, gm, d_desc, d_desc, frame_stride_bytes, d_counts, d_pairs_q, d_pairs_t, 0, 0, capacity, init_dist, d_out);
    }
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_frame_grid_build_batch_device(amos_match *m, const int32_t *d_grid_cell, const int32_t *d_counts, int n_frames, int capacity,
                                       int32_t *d_cell_start, int32_t *d_items)
{
    if (!m || !d_grid_cell || !d_counts || !d_cell_start || !d_items || n_frames < 1 || capacity < 1) {
        set_error("amos_frame_grid_build_batch_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_grid_build, dim3(n_frames), dim3(256), 0, m->stream, d_grid_cell, d_counts, capacity, d_cell_start, d_items);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_match_window_best2_batch_device(amos_match *m, const amos_window_search *w, amos_best2 *d_out)
{
    if (!m || !w || !d_out || !w->d_kps || !w->d_desc || !w->d_counts || !w->d_cell_start || !w->d_items || !w->d_pairs_q ||
        !w->d_pairs_t || w->n_pairs < 1 || w->capacity < 1 || w->capacity > 65535 || w->n_levels < 1 || w->n_levels > AMOS_MAX_LEVELS || !w->scale_factors ||
        !(w->max_x > w->min_x) || !(w->max_y > w->min_y) || w->mode < 0 || w->mode > 2 || !(w->th > 0)) {
        set_error("amos_match_window_best2_batch_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    WindowArgs a;
    a.kps = w->d_kps; a.desc = w->d_desc; a.counts = w->d_counts; a.cellStart = w->d_cell_start; a.items = w->d_items;
    a.queryUv = w->d_query_uv; a.queryInvZ = w->d_query_invz; a.uRight = w->d_u_right; a.pairsQ = w->d_pairs_q; a.pairsT = w->d_pairs_t;
    for (int l = 0; l < AMOS_MAX_LEVELS; l++) a.scale[l] = l < w->n_levels ? w->scale_factors[l] : 0.f;
    a.th = w->th; a.mbf = w->mbf; a.minX = w->min_x; a.minY = w->min_y;
    a.wInv = static_cast<float>(AMOS_FRAME_GRID_COLS) / static_cast<float>(w->max_x - w->min_x);  // Frame.cc:302-303
    a.hInv = static_cast<float>(AMOS_FRAME_GRID_ROWS) / static_cast<float>(w->max_y - w->min_y);
    a.capacity = w->capacity; a.mode = w->mode; a.initDist = w->init_dist;
    hipLaunchKernelGGL(k_window_best2, dim3((w->capacity * kWindowLanes + 255) / 256, w->n_pairs), dim3(256), 0, m->stream, a, d_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
