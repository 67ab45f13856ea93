// amos_mask_pre.hip -- the mask pass's pre-processing chain on the device: from the raw BGR
// frame to the network's input tensor in three small kernels, numerically what the reference does in three
// places:
//   A  yolact::evalImage (yolact.cc:220, 385-451): cv::resize(BGR u8 -> W480 x H640) [sic, swapped], u8 / 255.0
//      as CHW float, and eval_image's "* 255" (yolact_interface.py:862-864)            -> float HWC 640 x 480
//   B  eval_image's cv2.resize(float32, (640, 480)) (yolact_interface.py:865)          -> float HWC 480 x 640
//   C  FastBaseTransform (utils/augmentations.py:616-657): bilinear to 550 x 550 (align_corners=False),
//      (x - MEANS) / STD in BGR order, channels swapped to RGB                         -> float CHW 3 x 550 x 550
// The two OpenCV resizes use host-built tap tables (8-bit: 11-bit fixed point; float32: float
// weights), the u8 -> float step is a 256-entry table made on the host with the reference's double division.
// PyTorch is not involved: the network consumes d_out directly.
#include "amos_common.h"

#include <cmath>
#include <vector>

namespace amos {

constexpr int kNetSize = 550;            // cfg.max_size
constexpr int kMidW = kMaskMidW, kMidH = kMaskMidH;  // yolact.cc:220  cv::Size(480, 640)
constexpr int kBackW = 640, kBackH = 480;  // yolact_interface.py:865

struct FltTap { int s0, s1; float f0, f1; };  // float cv::resize: source indices, (1 - f) and f

// A: grid = (ceil(480 * 640 / 256), frames); one thread per intermediate pixel, three channels
__global__ __launch_bounds__(256) void k_mask_pre_a(const uint8_t *__restrict__ bgr, int srcW, int srcH, const FixTap *__restrict__ tx,
                                                   const FixTap *__restrict__ ty, const float *__restrict__ lut, float *__restrict__ mid)
{
    const int p = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (p >= kMidW * kMidH) return;
    const int y = p / kMidW, x = p - y * kMidW;
    const FixTap ax = tx[x], ay = ty[y];
    const uint8_t *r0 = bgr + ((size_t)frame * srcH + ay.s0) * srcW * 3, *r1 = bgr + ((size_t)frame * srcH + ay.s1) * srcW * 3;
    float *o = mid + ((size_t)frame * kMidH * kMidW + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
        const int h1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
        const int v = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = lut[v];  // float(double(v) / 255.0) * 255.0f
    }
}

// B: grid = (ceil(640 * 480 / 256), frames); cv::resize float INTER_LINEAR, horizontal then vertical, no FMA
__global__ __launch_bounds__(256) void k_mask_pre_b(const float *__restrict__ mid, const FltTap *__restrict__ tx, const FltTap *__restrict__ ty,
                                                   float *__restrict__ back)
{
    const int p = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (p >= kBackW * kBackH) return;
    const int y = p / kBackW, x = p - y * kBackW;
    const FltTap ax = tx[x], ay = ty[y];
    const float *r0 = mid + ((size_t)frame * kMidH + ay.s0) * kMidW * 3, *r1 = mid + ((size_t)frame * kMidH + ay.s1) * kMidW * 3;
    float *o = back + ((size_t)frame * kBackH * kBackW + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float h0 = __fadd_rn(__fmul_rn(r0[ax.s0 * 3 + c], ax.f0), __fmul_rn(r0[ax.s1 * 3 + c], ax.f1));
        const float h1 = __fadd_rn(__fmul_rn(r1[ax.s0 * 3 + c], ax.f0), __fmul_rn(r1[ax.s1 * 3 + c], ax.f1));
        o[c] = __fadd_rn(__fmul_rn(h0, ay.f0), __fmul_rn(h1, ay.f1));
    }
}

// C: grid = (ceil(550 * 550 / 256), frames); torch.nn.functional.interpolate(bilinear, align_corners=False),
// normalisation and the BGR -> RGB swap; writes planar CHW
__global__ __launch_bounds__(256) void k_mask_pre_c(const float *__restrict__ back, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (p >= kNetSize * kNetSize) return;
    const int oy = p / kNetSize, ox = p - oy * kNetSize;
    const float sh = (float)kBackH / (float)kNetSize, sw = (float)kBackW / (float)kNetSize;
    const float hr = fmaxf(__fsub_rn(__fmul_rn(sh, __fadd_rn((float)oy, 0.5f)), 0.5f), 0.f);
    const float wr = fmaxf(__fsub_rn(__fmul_rn(sw, __fadd_rn((float)ox, 0.5f)), 0.5f), 0.f);
    const int h1 = (int)hr, w1 = (int)wr;
    const int h1p = h1 < kBackH - 1 ? 1 : 0, w1p = w1 < kBackW - 1 ? 1 : 0;
    const float hl1 = __fsub_rn(hr, (float)h1), hl0 = __fsub_rn(1.f, hl1);
    const float wl1 = __fsub_rn(wr, (float)w1), wl0 = __fsub_rn(1.f, wl1);
    const float *b = back + (size_t)frame * kBackH * kBackW * 3;
    const float *p00 = b + ((size_t)h1 * kBackW + w1) * 3, *p01 = p00 + w1p * 3, *p10 = p00 + (size_t)h1p * kBackW * 3, *p11 = p10 + w1p * 3;
    const float mean[3] = {103.94f, 116.78f, 123.68f}, stdv[3] = {57.38f, 57.12f, 58.40f};  // BGR, data/config.py:28-29
    float *o = out + (size_t)frame * 3 * kNetSize * kNetSize + p;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float top = __fadd_rn(__fmul_rn(wl0, p00[c]), __fmul_rn(wl1, p01[c]));
        const float bot = __fadd_rn(__fmul_rn(wl0, p10[c]), __fmul_rn(wl1, p11[c]));
        const float v = __fadd_rn(__fmul_rn(hl0, top), __fmul_rn(hl1, bot));
        o[(size_t)(2 - c) * kNetSize * kNetSize] = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);  // RGB plane order
    }
}


// ---- epilogue of a convolution of the mask network, fused: y = act(y + bias[c] (+ residual)), in place, on an NHWC
// (channels-last) float tensor.  MIOpen's convolutions take no bias and PyTorch adds it, the residual and the ReLU as
// one elementwise pass each over the activations (three reads + three writes of the tensor per bottleneck output);
// this is one read (+ the residual) and one write.  Sums in the order (y + b) + r, like the unfused ops: same bits.
// grid = ceil(n / 4 / 256), block = 256; kVec: channels % 4 == 0 and 16-byte aligned pointers; kPow2: channels is a
// power of two (every ResNet / FPN width), so the channel of an element is a mask, not a 64-bit modulo.
template <bool kVec, bool kPow2>
__global__ __launch_bounds__(256) void k_bias_act(float *__restrict__ y, const float *__restrict__ bias, const float *__restrict__ res, size_t n,
                                                 int channels, int relu)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (kVec) {
        const unsigned c = kPow2 ? ((unsigned)i & (unsigned)(channels - 1)) : (unsigned)(i % (size_t)channels);
        float4 v = *reinterpret_cast<const float4 *>(y + i);
        const float4 b = *reinterpret_cast<const float4 *>(bias + c);
        v.x = __fadd_rn(v.x, b.x); v.y = __fadd_rn(v.y, b.y); v.z = __fadd_rn(v.z, b.z); v.w = __fadd_rn(v.w, b.w);
        if (res) {
            const float4 r = *reinterpret_cast<const float4 *>(res + i);
            v.x = __fadd_rn(v.x, r.x); v.y = __fadd_rn(v.y, r.y); v.z = __fadd_rn(v.z, r.z); v.w = __fadd_rn(v.w, r.w);
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4 *>(y + i) = v;
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) {
            float v = __fadd_rn(y[k], bias[k % (size_t)channels]);
            if (res) v = __fadd_rn(v, res[k]);
            y[k] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// The same resize for an exact x 2 enlargement (the prototype network's upsampling step: 256 channels, 69 -> 138, 1.25 GB written per 64
// frames): a thread makes the 2 x 2 output block around one source pixel from the 3 x 3 source pixels it touches -- 2.25 sixteen-byte
// reads per output instead of 4 (the one-output-per-thread kernel above moves 10 TB/s from L2 to L1 for this layer and is bound by that).
// Per output the same taps, weights and order of operations as k_bilinear_nhwc: the same bits.  A block whose two rows (columns) do not
// share their middle source row (column) -- the first and last ones, where the source index is clamped -- takes its taps one by one.
// grid = (ceil(ceil(outH / 2) * ceil(outW / 2) * c4 / 256), 1, n): the blocks of an image are numbered row by row, so a narrow image (the
// channel-blocked layout is images of 8 channels: 138 threads per block row) still fills its work-groups
template <bool kRelu>
__global__ __launch_bounds__(256) void k_bilinear_nhwc_x2(const float *__restrict__ x, float *__restrict__ y, int inH, int inW, int outH, int outW, int c4,
                                                         float scaleH, float scaleW)
{
    const int t = blockIdx.x * 256 + threadIdx.x, bw = (outW + 1) / 2, bh = (outH + 1) / 2, perRow = bw * c4;
    if (t >= bh * perRow) return;
    const int by = t / perRow, tr = t - by * perRow;
    const int bx = tr / c4, q = tr - bx * c4, n = blockIdx.z;
    int yy[2][2], xx[2][2];
    float wy[2][2], wx[2][2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        float fy = __fsub_rn(__fmul_rn(scaleH, __fadd_rn((float)(2 * by + e), 0.5f)), 0.5f);
        float fx = __fsub_rn(__fmul_rn(scaleW, __fadd_rn((float)(2 * bx + e), 0.5f)), 0.5f);
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        yy[e][0] = min((int)fy, inH - 1);  // (only a row / column past the output, never stored, can exceed the source)
        xx[e][0] = min((int)fx, inW - 1);
        yy[e][1] = yy[e][0] + (yy[e][0] < inH - 1 ? 1 : 0);
        xx[e][1] = xx[e][0] + (xx[e][0] < inW - 1 ? 1 : 0);
        wy[e][1] = __fsub_rn(fy, (float)(int)fy); wy[e][0] = __fsub_rn(1.f, wy[e][1]);
        wx[e][1] = __fsub_rn(fx, (float)(int)fx); wx[e][0] = __fsub_rn(1.f, wx[e][1]);
    }
    const float4 *src = reinterpret_cast<const float4 *>(x) + (size_t)n * inH * inW * c4;
    auto at = [&](int r, int c) { return src[((size_t)r * inW + c) * c4 + q]; };
    float4 s[4][4];  // [row tap: a0 a1 b0 b1][column tap: a0 a1 b0 b1]
    const bool shareY = yy[1][0] == yy[0][1], shareX = xx[1][0] == xx[0][1];
    const int ry[4] = {yy[0][0], yy[0][1], yy[1][0], yy[1][1]}, cx[4] = {xx[0][0], xx[0][1], xx[1][0], xx[1][1]};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r == 2 && shareY) {
#pragma unroll
            for (int c = 0; c < 4; c++) s[2][c] = s[1][c];
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c == 2 && shareX) s[r][2] = s[r][1];
            else s[r][c] = at(ry[r], cx[c]);
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(y);
#pragma unroll
    for (int ey = 0; ey < 2; ey++)
#pragma unroll
        for (int ex = 0; ex < 2; ex++) {
            const int oy = 2 * by + ey, ox = 2 * bx + ex;
            if (oy >= outH || ox >= outW) continue;
            const float hy = wy[ey][0], ly = wy[ey][1], hx = wx[ex][0], lx = wx[ex][1];
            const float4 a = s[2 * ey][2 * ex], b = s[2 * ey][2 * ex + 1], c = s[2 * ey + 1][2 * ex], d = s[2 * ey + 1][2 * ex + 1];
            auto mix = [&](float p, float r, float u, float v2) {
                const float v = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, p), __fmul_rn(lx, r))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, u), __fmul_rn(lx, v2))));
                return kRelu ? (v < 0.f ? 0.f : v) : v;
            };
            dst[(((size_t)n * outH + oy) * outW + ox) * c4 + q] = make_float4(mix(a.x, b.x, c.x, d.x), mix(a.y, b.y, c.y, d.y), mix(a.z, b.z, c.z, d.z), mix(a.w, b.w, c.w, d.w));
        }
}

// ---- the stem's tail: bias + ReLU + max_pool2d(3, stride 2, padding 1) of the 7 x 7 convolution's raw output in one pass
// (backbone.py ResNetBackbone.forward: conv1 -> bn1 (folded) -> relu -> maxpool).  relu(x + b) is monotone in x, and rounding is
// monotone, so max over the window of relu(x_i + b) == relu(max_i(x_i) + b) bit for bit: the kernel takes the maximum of the raw
// values (padding = -inf, as F.max_pool2d pads) and applies bias and ReLU once.  Replaces the in-place bias/ReLU pass over the
// 275 x 275 x 64 tensor (read + write) and the pooling kernel's read of it with one read.
// grid = (ceil(outW * C / 4 / 256), outH, N), block = 256; a thread makes four channels of one pooled pixel.
__global__ __launch_bounds__(256) void k_bias_relu_maxpool(const float *__restrict__ x, const float *__restrict__ bias, float *__restrict__ y, int inH, int inW,
                                                          int outH, int outW, int c4)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= outW * c4) return;
    const int ox = t / c4, q = t - ox * c4, oy = blockIdx.y, n = blockIdx.z;
    const float4 *src = reinterpret_cast<const float4 *>(x) + (size_t)n * inH * inW * c4;
    const float ninf = -__builtin_inff();
    float4 m = {ninf, ninf, ninf, ninf};
    bool nan = false;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
        const int iy = 2 * oy + dy;
        if ((unsigned)iy >= (unsigned)inH) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int ix = 2 * ox + dx;
            if ((unsigned)ix >= (unsigned)inW) continue;
            const float4 v = src[((size_t)iy * inW + ix) * c4 + q];
            nan = nan || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
            m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y; m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
        }
    }
    const float4 b = reinterpret_cast<const float4 *>(bias)[q];
    float4 r = {fmaxf(__fadd_rn(m.x, b.x), 0.f), fmaxf(__fadd_rn(m.y, b.y), 0.f), fmaxf(__fadd_rn(m.z, b.z), 0.f), fmaxf(__fadd_rn(m.w, b.w), 0.f)};
    if (nan) {  // rare path: per component, a NaN in the window wins (as in max_pool2d of the relu'd tensor)
        float *rp = &r.x;
        for (int e = 0; e < 4; e++) {
            bool cn = false;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                    if ((unsigned)iy < (unsigned)inH && (unsigned)ix < (unsigned)inW) {
                        const float v = (&src[((size_t)iy * inW + ix) * c4 + q].x)[e];
                        cn = cn || v != v;
                    }
                }
            if (cn) rp[e] = __builtin_nanf("");
        }
    }
    reinterpret_cast<float4 *>(y)[(((size_t)n * outH + oy) * outW + ox) * c4 + q] = r;
}

// ---- bilinear resize of an NHWC float tensor (F.interpolate(..., mode="bilinear", align_corners=False): the FPN's
// top-down path and the prototype network's x2 step, yolact.py:318-329, config mask_proto_net).  PyTorch's channels-last
// kernel for this took 13 % of the mask pass (5 ms per call at 32 frames); here a thread makes four channels of one
// output pixel with 16-byte loads / stores.  Source index and weights as PyTorch computes them in float32:
// src = scale * (dst + 0.5) - 0.5, clamped at 0, scale = 1 / scale_factor when given, else in / out.
// grid = (ceil(outW * C / 4 / 256), outH, N), block = 256.
template <bool kRelu>
__global__ __launch_bounds__(256) void k_bilinear_nhwc(const float *__restrict__ x, float *__restrict__ y, int inH, int inW, int outH, int outW, int c4,
                                                      float scaleH, float scaleW)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= outW * c4) return;
    const int ox = t / c4, q = t - ox * c4, oy = blockIdx.y, n = blockIdx.z;
    float fy = __fsub_rn(__fmul_rn(scaleH, __fadd_rn((float)oy, 0.5f)), 0.5f), fx = __fsub_rn(__fmul_rn(scaleW, __fadd_rn((float)ox, 0.5f)), 0.5f);
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < inH - 1 ? 1 : 0), x1 = x0 + (x0 < inW - 1 ? 1 : 0);
    const float ly = __fsub_rn(fy, (float)y0), lx = __fsub_rn(fx, (float)x0), hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
    const float4 *src = reinterpret_cast<const float4 *>(x) + (size_t)n * inH * inW * c4;
    const float4 a = src[((size_t)y0 * inW + x0) * c4 + q], b = src[((size_t)y0 * inW + x1) * c4 + q];
    const float4 c = src[((size_t)y1 * inW + x0) * c4 + q], d = src[((size_t)y1 * inW + x1) * c4 + q];
    auto mix = [&](float p, float r, float s, float u) {  // h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
        const float v = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, p), __fmul_rn(lx, r))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, s), __fmul_rn(lx, u))));
        return kRelu ? (v < 0.f ? 0.f : v) : v;  // a NaN stays a NaN, as through torch.relu
    };
    reinterpret_cast<float4 *>(y)[(((size_t)n * outH + oy) * outW + ox) * c4 + q] =
        make_float4(mix(a.x, b.x, c.x, d.x), mix(a.y, b.y, c.y, d.y), mix(a.z, b.z, c.z, d.z), mix(a.w, b.w, c.w, d.w));
}

// Fast NMS suppression term (layers/functions/detection.py:103-170 fast_nms as mask/detect.py restates it): for every class list of
// k score-sorted boxes, out[j] = max over i < j of IoU(box i, box j) (the column maxima of the upper-triangular IoU matrix; 0 for
// j = 0).  PyTorch spends nine elementwise passes over the [lists][k][k] matrix on this (410 MB at 32 frames x 80 classes x 200);
// here a work-group keeps one list's boxes in LDS and every thread walks its column.  box_utils.jaccard's arithmetic in its order:
// intersection = clamp(min(x2) - max(x1), 0) * clamp(min(y2) - max(y1), 0); IoU = inter / ((area_i + area_j) - inter), correctly
// rounded; a NaN (0 / 0) wins the maximum as it does in torch.max.
// grid = lists, block = 256 (k <= 256).
__global__ __launch_bounds__(256) void k_nms_column_max(const float4 *__restrict__ boxes, float *__restrict__ out, int k)
{
    __shared__ float4 sb[256];
    __shared__ float sarea[256];
    const int j = threadIdx.x;
    const float4 *b = boxes + (size_t)blockIdx.x * k;
    float4 me = {0.f, 0.f, 0.f, 0.f};
    float myArea = 0.f;
    if (j < k) {
        me = b[j];
        myArea = __fmul_rn(__fsub_rn(me.z, me.x), __fsub_rn(me.w, me.y));
        sb[j] = me;
        sarea[j] = myArea;
    }
    __syncthreads();
    if (j >= k) return;
    float best = 0.f;
    bool nan = false;
    for (int i = 0; i < j; i++) {
        const float4 o = sb[i];  // the same address for every lane: a broadcast
        float w = __fsub_rn(fminf(o.z, me.z), fmaxf(o.x, me.x)), h = __fsub_rn(fminf(o.w, me.w), fmaxf(o.y, me.y));
        w = w < 0.f ? 0.f : w;
        h = h < 0.f ? 0.f : h;
        const float inter = __fmul_rn(w, h);
        const float uni = __fsub_rn(__fadd_rn(sarea[i], myArea), inter);
        const float iou = (float)((double)inter / (double)uni);  // == the correctly rounded float quotient
        nan = nan || iou != iou;
        best = iou > best ? iou : best;
    }
    out[(size_t)blockIdx.x * k + j] = nan ? __builtin_nanf("") : best;
}

}  // namespace amos

using namespace amos;

struct amos_mask_pre {
    int device = 0, width = 0, height = 0, maxBatch = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    FixTap *dFixX = nullptr, *dFixY = nullptr;
    FltTap *dFltX = nullptr, *dFltY = nullptr;
    float *dLut = nullptr, *dMid = nullptr, *dBack = nullptr;
    int *dFirstX = nullptr, *dFirstY = nullptr;  // inverse of the stage-A tap tables (fused import, amos_orb.hip)
};

// cv::resize INTER_LINEAR source index and fraction per destination index (double -> float as OpenCV does);
// the horizontal pass clamps the fraction at both ends, the vertical pass only clips the indices.
static void axis_taps(int srcN, int dstN, bool clampFraction, std::vector<int> &s0, std::vector<int> &s1, std::vector<float> &f)
{
    const double scale = 1.0 / ((double)dstN / (double)srcN);
    s0.resize(dstN); s1.resize(dstN); f.resize(dstN);
    for (int d = 0; d < dstN; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fx);
        fx -= (float)s;
        if (clampFraction) {
            if (s < 0) { fx = 0; s = 0; }
            if (s >= srcN - 1) { fx = 0; s = srcN - 1; }
        }
        s0[d] = std::min(std::max(s, 0), srcN - 1);
        s1[d] = std::min(std::max(s + 1, 0), srcN - 1);
        f[d] = fx;
    }
}

static int round_even(float v) { return (int)std::nearbyintf(v); }

namespace amos {
int mask_pre_stage_a(amos_mask_pre *p, MaskPreStageA *out)
{
    if (!p || !out) { set_error("mask pre-processing handle missing"); return AMOS_ERR_INVALID; }
    *out = MaskPreStageA{p->dFixX, p->dFixY, p->dFirstX, p->dFirstY, p->dLut, p->dMid, p->width, p->height, p->maxBatch};
    return AMOS_OK;
}
int mask_pre_finish(amos_mask_pre *p, hipStream_t stream, int n_frames, float *d_out)
{
    hipLaunchKernelGGL(k_mask_pre_b, dim3((kBackW * kBackH + 255) / 256, n_frames), dim3(256), 0, stream, p->dMid, p->dFltX, p->dFltY, p->dBack);
    hipLaunchKernelGGL(k_mask_pre_c, dim3((kNetSize * kNetSize + 255) / 256, n_frames), dim3(256), 0, stream, p->dBack, d_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}
}  // namespace amos

extern "C" {

int amos_mask_pre_create(int device, void *stream, int width, int height, int max_batch, amos_mask_pre **out)
{
    if (!out || width < 2 || height < 2 || max_batch < 1) { set_error("amos_mask_pre_create: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_mask_pre *p = new amos_mask_pre();
    p->device = device; p->width = width; p->height = height; p->maxBatch = max_batch;
    if (stream) p->stream = (hipStream_t)stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete p; return AMOS_ERR_DEVICE; }
        p->ownStream = true;
    }
    std::vector<int> s0, s1;
    std::vector<float> f;
    auto fix = [&](int srcN, int dstN, bool clamp, FixTap **dst, int **first) -> hipError_t {
        axis_taps(srcN, dstN, clamp, s0, s1, f);
        std::vector<FixTap> t(dstN);
        for (int d = 0; d < dstN; d++) t[d] = FixTap{s0[d], s1[d], round_even((1.f - f[d]) * 2048.f), round_even(f[d] * 2048.f)};
        // first tap indices are non-decreasing: first[X] = the first destination index whose first tap is >= X
        std::vector<int> inv(srcN + 1);
        for (int X = 0, d = 0; X <= srcN; X++) {
            while (d < dstN && s0[d] < X) d++;
            inv[X] = d;
        }
        hipError_t e = hipMalloc((void **)dst, sizeof(FixTap) * dstN);
        if (e == hipSuccess) e = hipMemcpy(*dst, t.data(), sizeof(FixTap) * dstN, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)first, sizeof(int) * inv.size());
        if (e == hipSuccess) e = hipMemcpy(*first, inv.data(), sizeof(int) * inv.size(), hipMemcpyHostToDevice);
        return e;
    };
    auto flt = [&](int srcN, int dstN, bool clamp, FltTap **dst) -> hipError_t {
        axis_taps(srcN, dstN, clamp, s0, s1, f);
        std::vector<FltTap> t(dstN);
        for (int d = 0; d < dstN; d++) t[d] = FltTap{s0[d], s1[d], 1.0f - f[d], f[d]};
        hipError_t e = hipMalloc((void **)dst, sizeof(FltTap) * dstN);
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, t.data(), sizeof(FltTap) * dstN, hipMemcpyHostToDevice);
    };
    float lut[256];
    for (int v = 0; v < 256; v++) lut[v] = (float)((double)v / 255.0) * 255.0f;  // yolact.cc:424-431, yolact_interface.py:864
    hipError_t e = fix(width, kMidW, true, &p->dFixX, &p->dFirstX);
    if (e == hipSuccess) e = fix(height, kMidH, false, &p->dFixY, &p->dFirstY);
    if (e == hipSuccess) e = flt(kMidW, kBackW, true, &p->dFltX);
    if (e == hipSuccess) e = flt(kMidH, kBackH, false, &p->dFltY);
    if (e == hipSuccess) e = hipMalloc((void **)&p->dLut, sizeof(lut));
    if (e == hipSuccess) e = hipMemcpy(p->dLut, lut, sizeof(lut), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&p->dMid, sizeof(float) * 3 * kMidW * kMidH * (size_t)max_batch);
    if (e == hipSuccess) e = hipMalloc((void **)&p->dBack, sizeof(float) * 3 * kBackW * kBackH * (size_t)max_batch);
    if (e != hipSuccess) {
        set_error("amos_mask_pre_create: %s", hipGetErrorString(e));
        amos_mask_pre_destroy(p);
        return AMOS_ERR_DEVICE;
    }
    *out = p;
    return AMOS_OK;
}

void amos_mask_pre_destroy(amos_mask_pre *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void *ptrs[] = {p->dFixX, p->dFixY, p->dFltX, p->dFltY, p->dLut, p->dMid, p->dBack, p->dFirstX, p->dFirstY};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (p->ownStream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

void *amos_mask_pre_stream(amos_mask_pre *p) { return p ? (void *)p->stream : nullptr; }

int amos_mask_preprocess_batch_device(amos_mask_pre *p, const uint8_t *d_bgr, int n_frames, float *d_out)
{
    if (!p || !d_bgr || !d_out || n_frames < 1) { set_error("amos_mask_preprocess_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_frames > p->maxBatch) { set_error("amos_mask_preprocess_batch_device: %d frames, handle made for %d", n_frames, p->maxBatch); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipSetDevice(p->device));
    hipLaunchKernelGGL(k_mask_pre_a, dim3((kMidW * kMidH + 255) / 256, n_frames), dim3(256), 0, p->stream, d_bgr, p->width, p->height, p->dFixX,
                       p->dFixY, p->dLut, p->dMid);
    hipLaunchKernelGGL(k_mask_pre_b, dim3((kBackW * kBackH + 255) / 256, n_frames), dim3(256), 0, p->stream, p->dMid, p->dFltX, p->dFltY, p->dBack);
    hipLaunchKernelGGL(k_mask_pre_c, dim3((kNetSize * kNetSize + 255) / 256, n_frames), dim3(256), 0, p->stream, p->dBack, d_out);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}


int amos_mask_bias_act_device(void *stream, float *d_y, const float *d_bias, const float *d_residual, size_t n, int channels, int relu)
{
    if (!d_y || !d_bias || channels < 1 || n % (size_t)channels != 0) { set_error("amos_mask_bias_act_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n == 0) return AMOS_OK;
    const bool vec = channels % 4 == 0 && ((uintptr_t)d_y | (uintptr_t)d_bias | (uintptr_t)d_residual) % 16 == 0;
    const dim3 grid((unsigned)((n / 4 + 256) / 256)), block(256);
    const bool pow2 = (channels & (channels - 1)) == 0;
    if (vec && pow2) hipLaunchKernelGGL((k_bias_act<true, true>), grid, block, 0, (hipStream_t)stream, d_y, d_bias, d_residual, n, channels, relu);
    else if (vec) hipLaunchKernelGGL((k_bias_act<true, false>), grid, block, 0, (hipStream_t)stream, d_y, d_bias, d_residual, n, channels, relu);
    else hipLaunchKernelGGL((k_bias_act<false, false>), grid, block, 0, (hipStream_t)stream, d_y, d_bias, d_residual, n, channels, relu);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}


int amos_mask_bias_relu_maxpool_device(void *stream, const float *d_x, const float *d_bias, float *d_y, int n, int in_h, int in_w, int channels)
{
    if (!d_x || !d_bias || !d_y || n < 1 || in_h < 1 || in_w < 1 || channels < 4 || channels % 4 != 0 ||
        ((uintptr_t)d_x | (uintptr_t)d_y | (uintptr_t)d_bias) % 16 != 0 || n > 65535 || in_h > 2 * 65535) {
        set_error("amos_mask_bias_relu_maxpool_device: invalid argument (channels % 4 == 0, 16-byte aligned tensors)");
        return AMOS_ERR_INVALID;
    }
    const int oh = (in_h + 2 - 3) / 2 + 1, ow = (in_w + 2 - 3) / 2 + 1, c4 = channels / 4;
    hipLaunchKernelGGL(k_bias_relu_maxpool, dim3((ow * c4 + 255) / 256, oh, n), dim3(256), 0, (hipStream_t)stream, d_x, d_bias, d_y, in_h, in_w, oh, ow, c4);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}


static std::atomic<int> g_bilinear_x2{1};
int amos_mask_bilinear_x2_mode(int mode)
{
    const int before = g_bilinear_x2.load();
    if (mode == 0 || mode == 1) g_bilinear_x2.store(mode);
    return before;
}

int amos_mask_bilinear_nhwc_act_device(void *stream, const float *d_x, float *d_y, int n, int in_h, int in_w, int out_h, int out_w, int channels,
                                       float scale_h, float scale_w, int relu)
{
    if (!d_x || !d_y || n < 1 || in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || channels < 4 || channels % 4 != 0 ||
        ((uintptr_t)d_x | (uintptr_t)d_y) % 16 != 0 || out_h > 65535 || n > 65535) {
        set_error("amos_mask_bilinear_nhwc_device: invalid argument (channels % 4 == 0, 16-byte aligned tensors)");
        return AMOS_ERR_INVALID;
    }
    const int c4 = channels / 4;
    if (g_bilinear_x2 && out_h == 2 * in_h && out_w == 2 * in_w && scale_h == 0.5f && scale_w == 0.5f) {  // an exact x 2 enlargement: 2 x 2 outputs per thread
        const dim3 grid2((unsigned)(((long long)((out_h + 1) / 2) * ((out_w + 1) / 2) * c4 + 255) / 256), 1, n);
        if (relu) hipLaunchKernelGGL(k_bilinear_nhwc_x2<true>, grid2, dim3(256), 0, (hipStream_t)stream, d_x, d_y, in_h, in_w, out_h, out_w, c4, scale_h, scale_w);
        else hipLaunchKernelGGL(k_bilinear_nhwc_x2<false>, grid2, dim3(256), 0, (hipStream_t)stream, d_x, d_y, in_h, in_w, out_h, out_w, c4, scale_h, scale_w);
        AMOS_HIP_CHECK(hipGetLastError());
        return AMOS_OK;
    }
    const dim3 grid((out_w * c4 + 255) / 256, out_h, n), block(256);
    if (relu) hipLaunchKernelGGL(k_bilinear_nhwc<true>, grid, block, 0, (hipStream_t)stream, d_x, d_y, in_h, in_w, out_h, out_w, c4, scale_h, scale_w);
    else hipLaunchKernelGGL(k_bilinear_nhwc<false>, grid, block, 0, (hipStream_t)stream, d_x, d_y, in_h, in_w, out_h, out_w, c4, scale_h, scale_w);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_bilinear_nhwc_device(void *stream, const float *d_x, float *d_y, int n, int in_h, int in_w, int out_h, int out_w, int channels,
                                   float scale_h, float scale_w)
{
    return amos_mask_bilinear_nhwc_act_device(stream, d_x, d_y, n, in_h, in_w, out_h, out_w, channels, scale_h, scale_w, 0);
}

int amos_mask_nms_column_max_device(void *stream, const float *d_boxes, float *d_out, int n_lists, int k)
{
    if (!d_boxes || !d_out || n_lists < 0 || k < 1 || k > 256 || (uintptr_t)d_boxes % 16 != 0) {
        set_error("amos_mask_nms_column_max_device: invalid argument (1 <= k <= 256, 16-byte aligned boxes)");
        return AMOS_ERR_INVALID;
    }
    if (n_lists == 0) return AMOS_OK;
    hipLaunchKernelGGL(k_nms_column_max, dim3(n_lists), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(d_boxes), d_out, k);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"
