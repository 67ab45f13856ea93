// amos_stem.hip -- the stem of the mask network's backbone (a15; backbone.py:77-80,129-132: conv1 7 x 7 stride 2 padding 3, 3 -> 64
// channels, bn1 folded into it, ReLU, max_pool2d 3 x 3 stride 2 padding 1) as ONE kernel on the fp32 MFMA units.
//
// What it replaces in a pass: a layout copy of the network input (planar -> channels-last), the library's zero-fill + implicit GEMM
// (1.08 ms per 64 frames, its 275 x 275 x 64 output is 1.24 GB) and the bias + ReLU + max-pool pass over that output (0.35 ms): the
// convolution's output never reaches memory here -- a work-group keeps the accumulators of a 13 x 47 block of convolution pixels in
// registers, hands them over through LDS and writes the 6 x 23 pooled pixels they cover.  HBM traffic per frame: 3.6 MB in, 4.9 MB out.
//
// GEMM view: rows = convolution pixels, columns = 32 of the 64 output channels (blockIdx.x & 1 picks the half), k = (dy, dx, c) of the
// 7 x 7 x 3 window.  v_mfma_f32_32x32x2_f32 takes k = 2 s + h with h = lane >> 5; the sum over k is order-free, so the k of step s is
// CHOSEN: the input patch lies in LDS as [row][column][channel], a window row is 21 consecutive floats, and lane half 0 takes floats
// 0 .. 10 of it, half 1 floats 11 .. 21 (float 21, outside the window, meets a zero weight and is replaced by zero on the way in): the
// address of every A operand is "pixel base + 44 h bytes + a compile-time immediate", one ds_read_b32 per MFMA, 77 steps for the 147
// products.  The B operands (77 weights per lane) stay in registers for the whole work-group.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

// AMOS_STEM_EXP: timing experiments of tools/stem_bench.hip (bit 0: no MFMA phase, bit 1: no hand-over / pooling, bit 2: no patch load) --
// wrong results, so the switch exists only in that program's own compilation of this file
#if defined(AMOS_STEM_EXP) && !defined(AMOS_STEM_BENCH)
#error "AMOS_STEM_EXP is for tools/stem_bench.hip only"
#endif
#ifndef AMOS_STEM_EXP
#define AMOS_STEM_EXP 0
#endif

namespace amos {

constexpr int kStemPH = 6, kStemPW = 23;                            // pooled pixels of a work-group (138 = 6 x 23: the network's size divides)
constexpr int kStemCR = 2 * kStemPH + 1, kStemCC = 2 * kStemPW + 1;  // 13 x 47 convolution pixels under them
constexpr int kStemPix = kStemCR * kStemCC;                         // 611
constexpr int kStemTiles = (kStemPix + 31) / 32;                    // 20 MFMA row tiles
constexpr int kStemWaves = 4, kStemTilesPerWave = kStemTiles / kStemWaves;
constexpr int kStemIR = 2 * kStemCR + 5, kStemIC = 2 * kStemCC + 5;  // 31 x 99 input pixels under those
constexpr int kStemPitch = kStemIC * 3;                             // floats per patch row
constexpr int kStemPatch = kStemIR * kStemPitch;                    // 9 207 floats (+ 1: the last pixel's float 21)
constexpr int kStemHalfK = 11, kStemSteps = 7 * kStemHalfK;         // MFMA steps: 7 window rows x 11
constexpr int kStemLdsFloats = kStemPix * 32;                       // the hand-over image [pixel][32 channels] reuses the patch's LDS
static_assert(kStemTiles % kStemWaves == 0 && kStemLdsFloats > kStemPatch, "tile split / LDS reuse");

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StemArgs {
    const float *x;
    long long sb, sc, sy, sx;  // element strides of the input: frame, channel, row, column (planar or channels-last alike)
    const float *wp, *bias;
    float *y;                  // [batch][poolH][poolW][64]
    int H, W, convH, convW, poolH, poolW, tilesX;
};

// packed[(half * 77 + s) * 64 + lane] = w[n = 32 half + (lane & 31)][c][dy][dx]  with s = 11 dy + j, jj = j + 11 (lane >> 5),
// (dx, c) = (jj / 3, jj % 3), zero for jj = 21
__global__ __launch_bounds__(256) void k_stem_pack(const float *__restrict__ w, long long sn, long long sc, long long sy, long long sx, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * kStemSteps * 64) return;
    const int lane = i & 63, s = (i >> 6) % kStemSteps, half = i / (64 * kStemSteps);
    const int dy = s / kStemHalfK, jj = s % kStemHalfK + kStemHalfK * (lane >> 5), n = 32 * half + (lane & 31);
    out[i] = jj > 20 ? 0.f : w[n * sn + (jj % 3) * sc + dy * sy + (jj / 3) * sx];
}

// (Two work-groups share a CU and run their three phases in lock-step -- 1.10 ms per 64 frames of which the multiply phases are 0.80,
// tools/stem_bench.hip.  Giving every second arrival at a CU a higher wave priority (a counter per CU indexed by the hardware id
// registers, s_setprio) so that neighbours alternate made it SLOWER, 1.15 ms: the arbiter starves the other group's load / pool
// instructions too.  Starting the work-groups of the CUs' second slots 3 - 20 us late (so that neighbours are out of phase from the first
// round on -- the groups of ids 256 .. 511, or whichever group finds another one resident on its CU by a per-CU counter) changed nothing: 1.10 -
// 1.11 ms.  Neither is kept.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_stem_conv_pool(const StemArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, hi = lane >> 5;
    const int half = blockIdx.x & 1, tile = blockIdx.x >> 1, b = blockIdx.y;
    const int ty = tile / a.tilesX, tx = tile - ty * a.tilesX;
    const int py0 = ty * kStemPH, px0 = tx * kStemPW;  // first pooled pixel; convolution pixel (2 py0 - 1, 2 px0 - 1) is local (0, 0)
    const int iy0 = 4 * py0 - 5, ix0 = 4 * px0 - 5;    // ... and input pixel (iy0, ix0) is patch (0, 0): 2 (2 py0 - 1) - 3
    // the weights of this channel half: one coalesced load per step, in registers until the end
    float bw[kStemSteps];
    {
        const float *wp = a.wp + (size_t)half * kStemSteps * 64 + lane;
#pragma unroll
        for (int s = 0; s < kStemSteps; s++) bw[s] = wp[s * 64];
    }
    // the input patch -> LDS [row][column][channel]; zeros outside the image (the convolution's padding).  Every load of the thread is
    // issued before the first LDS write (addresses clamped into the image, the value selected afterwards): one memory latency per
    // work-group instead of one per sweep -- the two work-groups of a CU start together and would otherwise sit in this phase together
    {
        const float *xb = a.x + (size_t)b * a.sb;
        constexpr int kSweeps = (kStemIR * kStemIC + 255) / 256;  // 12
        float v[kSweeps][3];
#pragma unroll
        for (int i = 0; i < ((AMOS_STEM_EXP & 4) ? 0 : kSweeps); i++) {
            const int e = min(t + 256 * i, kStemIR * kStemIC - 1);
            const int y = e / kStemIC, x = e - y * kStemIC, iy = iy0 + y, ix = ix0 + x;
            const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const float *p = xb + (long long)min(max(iy, 0), a.H - 1) * a.sy + (long long)min(max(ix, 0), a.W - 1) * a.sx;
            const float l0 = p[0], l1 = p[a.sc], l2 = p[2 * a.sc];
            v[i][0] = in ? l0 : 0.f; v[i][1] = in ? l1 : 0.f; v[i][2] = in ? l2 : 0.f;
        }
#pragma unroll
        for (int i = 0; i < ((AMOS_STEM_EXP & 4) ? 0 : kSweeps); i++) {
            const int e = t + 256 * i;
            if (e < kStemIR * kStemIC) {
                const int y = e / kStemIC, x = e - y * kStemIC;
                float *d = &lds[y * kStemPitch + 3 * x];
                d[0] = v[i][0]; d[1] = v[i][1]; d[2] = v[i][2];
            }
        }
        if (t == 0) lds[kStemPatch] = 0.f;
    }
    // (the compiler otherwise sinks the weight loads into the MFMA phase, one window row's worth in front of its first use: seven exposed
    // memory latencies per work-group, measured 31 instead of 19 us for a work-group alone)
#pragma unroll
    for (int s = 0; s < kStemSteps; s++) asm volatile("" : "+v"(bw[s]));
    __syncthreads();
    int base[kStemTilesPerWave];
#pragma unroll
    for (int i = 0; i < kStemTilesPerWave; i++) {
        const int p = min((wave + kStemWaves * i) * 32 + (lane & 31), kStemPix - 1);  // rows past the block repeat its last pixel; not stored
        const int cy = p / kStemCC, cx = p - cy * kStemCC;
        base[i] = 2 * cy * kStemPitch + 6 * cx + kStemHalfK * hi;
    }
    f32x16 acc[kStemTilesPerWave];
#pragma unroll
    for (int i = 0; i < kStemTilesPerWave; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
#pragma unroll
    for (int dy = 0; dy < ((AMOS_STEM_EXP & 1) ? 0 : 7); dy++)
#pragma unroll
        for (int j = 0; j < kStemHalfK; j++) {
            float av[kStemTilesPerWave];
#pragma unroll
            for (int i = 0; i < kStemTilesPerWave; i++) {
                av[i] = lds[base[i] + dy * kStemPitch + j];
                if (j == kStemHalfK - 1) av[i] = hi ? 0.f : av[i];  // float 21 of the window row belongs to the next pixel (its weight is zero; a NaN there must not leak)
            }
#pragma unroll
            for (int i = 0; i < kStemTilesPerWave; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[dy * kStemHalfK + j], acc[i], 0, 0, 0);
        }
    if (AMOS_STEM_EXP & 2) {  // (keeps the accumulators and weights alive)
        float sum = 0.f;
        for (int i = 0; i < kStemTilesPerWave; i++)
            for (int r = 0; r < 16; r++) sum += acc[i][r];
        for (int q = 0; q < kStemSteps; q++) sum += bw[q];
        if (sum == 12345.678f) a.y[t] = sum;
        return;
    }
    // hand-over: accumulator register r of a tile is row (r & 3) + 8 (r >> 2) + 4 hi, column lane & 31 -> LDS [pixel][channel]
    __syncthreads();  // nobody reads the patch any more
#pragma unroll
    for (int i = 0; i < kStemTilesPerWave; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (wave + kStemWaves * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < kStemPix) lds[row * 32 + (lane & 31)] = acc[i][r];
        }
    __syncthreads();
    // max-pool 3 x 3 stride 2 (padding = -inf, as F.max_pool2d pads), then bias + ReLU once: relu(x + b) and rounding are monotone, so
    // max_i relu(x_i + b) == relu(max_i x_i + b) bit for bit.  A NaN in the window wins, as in torch.
    const int g = t & 7;  // four channels
    const float4 bv = *reinterpret_cast<const float4 *>(a.bias + 32 * half + 4 * g);
    const float ninf = -__builtin_inff();
    for (int q = t >> 3; q < kStemPH * kStemPW; q += 32) {
        const int pr = q / kStemPW, pc = q - pr * kStemPW, py = py0 + pr, px = px0 + pc;
        if (py >= a.poolH || px >= a.poolW) continue;
        float4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            if ((unsigned)(2 * py - 1 + dy) >= (unsigned)a.convH) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                if ((unsigned)(2 * px - 1 + dx) >= (unsigned)a.convW) continue;
                const float4 v = *reinterpret_cast<const float4 *>(&lds[((2 * pr + dy) * kStemCC + 2 * pc + dx) * 32 + 4 * g]);
                m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x; m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
                m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z; m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
            }
        }
        float4 r = {__fadd_rn(m.x, bv.x), __fadd_rn(m.y, bv.y), __fadd_rn(m.z, bv.z), __fadd_rn(m.w, bv.w)};
        r.x = r.x < 0.f ? 0.f : r.x; r.y = r.y < 0.f ? 0.f : r.y; r.z = r.z < 0.f ? 0.f : r.z; r.w = r.w < 0.f ? 0.f : r.w;  // (a NaN stays)
        *reinterpret_cast<float4 *>(a.y + (((size_t)b * a.poolH + py) * a.poolW + px) * 64 + 32 * half + 4 * g) = r;
    }
}

static DeviceOnce g_stem_lds_once;

}  // namespace amos

using namespace amos;

extern "C" {

int amos_mask_stem_weight_floats(void) { return 2 * kStemSteps * 64; }

int amos_mask_stem_weights_device(void *stream, const float *d_w, long long sn, long long sc, long long sy, long long sx, float *d_packed)
{
    if (!d_w || !d_packed) { set_error("amos_mask_stem_weights_device: invalid argument"); return AMOS_ERR_INVALID; }
    hipLaunchKernelGGL(k_stem_pack, dim3((2 * kStemSteps * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_w, sn, sc, sy, sx, d_packed);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_stem_device(void *stream, const float *d_x, long long sb, long long sc, long long sy, long long sx, const float *d_packed, const float *d_bias,
                          float *d_y, int batch, int height, int width)
{
    if (!d_x || !d_packed || !d_bias || !d_y || batch < 1 || batch > 65535 || height < 1 || width < 1 || ((uintptr_t)d_y | (uintptr_t)d_bias) % 16 != 0) {
        set_error("amos_mask_stem_device: invalid argument (1 <= batch <= 65535, 16-byte aligned bias and output)");
        return AMOS_ERR_INVALID;
    }
    StemArgs a;
    a.x = d_x; a.sb = sb; a.sc = sc; a.sy = sy; a.sx = sx; a.wp = d_packed; a.bias = d_bias; a.y = d_y;
    a.H = height; a.W = width;
    a.convH = (height - 1) / 2 + 1; a.convW = (width - 1) / 2 + 1;  // (h + 6 - 7) / 2 + 1
    a.poolH = (a.convH - 1) / 2 + 1; a.poolW = (a.convW - 1) / 2 + 1;  // (h + 2 - 3) / 2 + 1
    a.tilesX = (a.poolW + kStemPW - 1) / kStemPW;
    const long long tiles = (long long)a.tilesX * ((a.poolH + kStemPH - 1) / kStemPH);
    if (2 * tiles > 0x7fffffffLL) { set_error("amos_mask_stem_device: image too large"); return AMOS_ERR_INVALID; }
    const int ldsBytes = kStemLdsFloats * (int)sizeof(float);
    AMOS_HIP_CHECK(set_max_dynamic_lds(g_stem_lds_once, reinterpret_cast<const void *>(k_stem_conv_pool), ldsBytes, (hipStream_t)stream));
    hipLaunchKernelGGL(k_stem_conv_pool, dim3((unsigned)(2 * tiles), batch), dim3(256), ldsBytes, (hipStream_t)stream, a);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
