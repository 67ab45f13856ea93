// amos_match.hip -- 256-bit Hamming distance and the best / second-best reductions that the
// inner loops of ORBmatcher::Search* perform (ORBmatcher.cc:1913-1933 and e.g. :127-148).
//
// Integer work: xor + v_bcnt (popcount) per 32-bit word, keys ordered by (distance, candidate
// position) so that ties resolve exactly as the reference's sequential
//     if(dist<bestDist){...} else if(dist<bestDist2){...}
// loop does (first candidate wins; see DESIGN.md for the proof of equivalence).  No MFMA.
#include "amos_common.h"

#include <algorithm>
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
    auto eval = [&](const uint4 &lo, const uint4 &hi, int jj) {
        int d = __popc(qd.w[0] ^ lo.x);
        d = bcnt_acc(qd.w[1] ^ lo.y, d);
        d = bcnt_acc(qd.w[2] ^ lo.z, d);
        d = bcnt_acc(qd.w[3] ^ lo.w, d);
        d = bcnt_acc(qd.w[4] ^ hi.x, d);
        d = bcnt_acc(qd.w[5] ^ hi.y, d);
        d = bcnt_acc(qd.w[6] ^ hi.z, d);
        d = bcnt_acc(qd.w[7] ^ hi.w, d);
        unsigned key = ((unsigned)d << 16) | (unsigned)jj;
        if (kGate) key = d < initDist ? key : 0xffffffffu;  // initDist >= 257 can never reject
        top2_push(best, second, key);
    };
    int jj = j0;
    for (; jj + 4 <= j1; jj += 4) {  // four descriptors (eight scalar loads) in flight per trip
        const uint4 a0 = tv[2 * jj], a1 = tv[2 * jj + 1], b0 = tv[2 * jj + 2], b1 = tv[2 * jj + 3];
        const uint4 c0 = tv[2 * jj + 4], c1 = tv[2 * jj + 5], d0 = tv[2 * jj + 6], d1 = tv[2 * jj + 7];
        eval(a0, a1, jj);
        eval(b0, b1, jj + 1);
        eval(c0, c1, jj + 2);
        eval(d0, d1, jj + 3);
    }
    for (; jj < j1; jj++) eval(tv[2 * jj], tv[2 * jj + 1], jj);
    mergeB[wave][lane] = best;
    mergeS[wave][lane] = second;
    __syncthreads();
    if (wave == 0 && qi < nq) {
#pragma unroll
        for (int w = 1; w < 4; w++) top2_merge(best, second, mergeB[w][lane], mergeS[w][lane]);
        amos_best2 r;
        r.best_idx = best == 0xffffffffu ? -1 : (int)(best & 0xffff);
        r.best_dist = best == 0xffffffffu ? initDist : (int)(best >> 16);
        r.second_idx = second == 0xffffffffu ? -1 : (int)(second & 0xffff);
        r.second_dist = second == 0xffffffffu ? initDist : (int)(second >> 16);
        out[(size_t)pair * capacity + qi] = r;
    }
}

}  // namespace amos

using namespace amos;

struct amos_match {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // grow-only device scratch for the host-pointer entry points
    uint8_t *dQ = nullptr, *dT = nullptr;
    size_t capQ = 0, capT = 0;
    int *dOff = nullptr, *dIdx = nullptr;
    size_t capOff = 0, capIdx = 0;
    void *dOut = nullptr;
    size_t capOut = 0;
};

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

static int upload_sets(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt)
{
    int rc = grow(&m->dQ, &m->capQ, (size_t)nq * 32 + 32);
    if (rc != AMOS_OK) return rc;
    rc = grow(&m->dT, &m->capT, (size_t)nt * 32 + 32);
    if (rc != AMOS_OK) return rc;
    if (nq > 0) AMOS_HIP_CHECK(hipMemcpyAsync(m->dQ, q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    if (nt > 0) AMOS_HIP_CHECK(hipMemcpyAsync(m->dT, t, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
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
    int rc = grow(&m->dOff, &m->capOff, (size_t)nq + 1);
    if (rc != AMOS_OK) return rc;
    rc = grow(&m->dIdx, &m->capIdx, (size_t)n + 1);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipMemcpyAsync(m->dOff, cand_off, sizeof(int) * ((size_t)nq + 1), hipMemcpyHostToDevice, m->stream));
    if (n > 0) AMOS_HIP_CHECK(hipMemcpyAsync(m->dIdx, cand_idx, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, m->stream));
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
    void *ptrs[] = {m->dQ, m->dT, m->dOff, m->dIdx, m->dOut};
    for (void *p : ptrs) if (p) (void)hipFree(p);
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

int amos_match_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *out)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || (!out && nq > 0 && nt > 0)) { set_error("amos_match_distances: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0 || nt == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    const size_t bytes = (size_t)nq * nt * sizeof(uint16_t);
    rc = grow_out(m, bytes);
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_dist_dense, dim3((nt + 255) / 256, (nq + 15) / 16), dim3(256), 0, m->stream, m->dQ, nq, m->dT, nt, (uint16_t *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    AMOS_HIP_CHECK(hipMemcpyAsync(out, m->dOut, bytes, hipMemcpyDeviceToHost, m->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    return AMOS_OK;
}

int amos_match_list_distances(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *cand_off,
                              const int32_t *cand_idx, uint16_t *out)
{
    if (!m || nq < 0 || nt < 0 || !cand_off || (nq > 0 && !q)) { set_error("amos_match_list_distances: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int total = 0;
    int rc = upload_lists(m, nq, nt, cand_off, cand_idx, &total);
    if (rc != AMOS_OK) return rc;
    if (total == 0) return AMOS_OK;
    if (!t || !out || !cand_idx) { set_error("amos_match_list_distances: null buffer"); return AMOS_ERR_INVALID; }
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)total * sizeof(uint16_t));
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_list_dist, dim3((nq + 3) / 4), dim3(256), 0, m->stream, m->dQ, nq, m->dT, m->dOff, m->dIdx, (uint16_t *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    AMOS_HIP_CHECK(hipMemcpyAsync(out, m->dOut, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, m->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    return AMOS_OK;
}

int amos_match_list_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *cand_off,
                          const int32_t *cand_idx, int init_dist, amos_best2 *out)
{
    if (!m || nq < 0 || nt < 0 || !cand_off || (nq > 0 && (!q || !out))) { set_error("amos_match_list_best2: invalid argument"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int total = 0;
    int rc = upload_lists(m, nq, nt, cand_off, cand_idx, &total);
    if (rc != AMOS_OK) return rc;
    if (total > 0 && (!t || !cand_idx)) { set_error("amos_match_list_best2: null buffer"); return AMOS_ERR_INVALID; }
    rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    hipLaunchKernelGGL(k_list_best2, dim3((nq + 3) / 4), dim3(256), 0, m->stream, m->dQ, nq, m->dT, m->dOff, m->dIdx, init_dist, (amos_best2 *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    AMOS_HIP_CHECK(hipMemcpyAsync(out, m->dOut, (size_t)nq * sizeof(amos_best2), hipMemcpyDeviceToHost, m->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    return AMOS_OK;
}

int amos_match_bruteforce_best2(amos_match *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int init_dist,
                                amos_best2 *out)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) { set_error("amos_match_bruteforce_best2: invalid argument"); return AMOS_ERR_INVALID; }
    if (nt > 65535) { set_error("amos_match_bruteforce_best2: at most 65535 train descriptors"); return AMOS_ERR_INVALID; }
    if (nq == 0) return AMOS_OK;
    AMOS_HIP_CHECK(hipSetDevice(m->device));
    int rc = upload_sets(m, q, nq, t, nt);
    if (rc != AMOS_OK) return rc;
    rc = grow_out(m, (size_t)nq * sizeof(amos_best2));
    if (rc != AMOS_OK) return rc;
    if (init_dist > 256)
        hipLaunchKernelGGL(k_bf_best2<false>, dim3((nq + 63) / 64, 1), dim3(256), 0, m->stream, m->dQ, m->dT, (size_t)0, (const int *)nullptr,
                           (const int *)nullptr, (const int *)nullptr, nq, nt, nq, init_dist, (amos_best2 *)m->dOut);
    else
        hipLaunchKernelGGL(k_bf_best2<true>, dim3((nq + 63) / 64, 1), dim3(256), 0, m->stream, m->dQ, m->dT, (size_t)0, (const int *)nullptr,
                           (const int *)nullptr, (const int *)nullptr, nq, nt, nq, init_dist, (amos_best2 *)m->dOut);
    AMOS_HIP_CHECK(hipGetLastError());
    AMOS_HIP_CHECK(hipMemcpyAsync(out, m->dOut, (size_t)nq * sizeof(amos_best2), hipMemcpyDeviceToHost, m->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(m->stream));
    return AMOS_OK;
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
    if (init_dist > 256)
        hipLaunchKernelGGL(k_bf_best2<false>, dim3((capacity + 63) / 64, n_pairs), dim3(256), 0, m->stream, d_desc, d_desc, frame_stride_bytes,
                           d_counts, d_pairs_q, d_pairs_t, 0, 0, capacity, init_dist, d_out);
    else
        hipLaunchKernelGGL(k_bf_best2<true>, dim3((capacity + 63) / 64, n_pairs), dim3(256), 0, m->stream, d_desc, d_desc, frame_stride_bytes,
                           d_counts, d_pairs_q, d_pairs_t, 0, 0, capacity, init_dist, d_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
