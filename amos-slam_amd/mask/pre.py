"""Pre-processing chain of the mask pass, exactly as the reference runs it:

  C++ (yolact.cc:220, 385-451): cv::resize(BGR u8 480x640x3 -> W480 x H640)  [sic: swapped sizes],
                                CHW float32 = u8 / 255.0
  Python (yolact_interface.py:862-866): HWC, * 255, cv2.resize(float32 -> W640 x H480)
  FastBaseTransform (utils/augmentations.py:616-657): bilinear to 550x550 (align_corners=False),
                                (x - MEANS) / STD in BGR order, channels swapped to RGB

All of it runs as torch ops on the GPU here (one pass over the frame; the reference does the two
resizes on the CPU).  The two OpenCV resizes are restated from OpenCV 4.5's algorithms (8-bit:
11-bit fixed point as SURVEY A.1; float32: float weights): parity with OpenCV itself is unpinned,
see DESIGN.md.
"""
import numpy as np
import torch
import torch.nn.functional as F

MEANS = (103.94, 116.78, 123.68)  # BGR, data/config.py:28
STD = (57.38, 57.12, 58.40)


def _axis_taps(src_n, dst_n, clamp_fraction):
    """cv::resize INTER_LINEAR source index and fraction per destination index (double -> float as OpenCV)."""
    scale = 1.0 / (np.float64(dst_n) / np.float64(src_n))
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:  # horizontal pass: fraction reset at both ends
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= src_n - 1
        f[hi], s[hi] = 0, src_n - 1
    s0 = np.clip(s, 0, src_n - 1)
    s1 = np.clip(s + 1, 0, src_n - 1)
    return s0, s1, f


def resize_u8_cv(img, dst_w, dst_h):
    """cv::resize(INTER_LINEAR) of an 8-bit [..., H, W, C] tensor: 11-bit fixed-point weights,
    ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2 >> 2."""
    h, w = img.shape[-3], img.shape[-2]
    dev = img.device
    x0, x1, fx = _axis_taps(w, dst_w, True)
    y0, y1, fy = _axis_taps(h, dst_h, False)

    def fixed(f):
        a1 = np.rint(f * np.float32(2048)).astype(np.int32)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
        return torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)

    a0, a1 = fixed(fx)
    b0, b1 = fixed(fy)
    x0, x1, y0, y1 = (torch.from_numpy(v).to(dev) for v in (x0, x1, y0, y1))
    s = img.to(torch.int32)                                       # all intermediates fit 32 bits
    rows0, rows1 = s.index_select(-3, y0), s.index_select(-3, y1)  # [..., dh, w, c]
    a0, a1 = a0.view(-1, 1), a1.view(-1, 1)
    h0 = rows0.index_select(-2, x0) * a0 + rows0.index_select(-2, x1) * a1  # [..., dh, dw, c]
    h1 = rows1.index_select(-2, x0) * a0 + rows1.index_select(-2, x1) * a1
    b0, b1 = b0.view(-1, 1, 1), b1.view(-1, 1, 1)
    out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    return out.to(torch.uint8)


def resize_f32_cv(img, dst_w, dst_h):
    """cv::resize(INTER_LINEAR) of a float32 [..., H, W, C] tensor: float weights, horizontal then vertical."""
    h, w = img.shape[-3], img.shape[-2]
    dev = img.device
    x0, x1, fx = _axis_taps(w, dst_w, True)
    y0, y1, fy = _axis_taps(h, dst_h, False)
    fx_t = torch.from_numpy(fx).to(dev).view(-1, 1)
    fy_t = torch.from_numpy(fy).to(dev).view(-1, 1, 1)
    x0, x1, y0, y1 = (torch.from_numpy(v).to(dev) for v in (x0, x1, y0, y1))
    rows0, rows1 = img.index_select(-3, y0), img.index_select(-3, y1)
    h0 = rows0.index_select(-2, x0) * (1.0 - fx_t) + rows0.index_select(-2, x1) * fx_t
    h1 = rows1.index_select(-2, x0) * (1.0 - fx_t) + rows1.index_select(-2, x1) * fx_t
    return h0 * (1.0 - fy_t) + h1 * fy_t


def cxx_marshalling(bgr_u8):
    """What yolact::evalImage hands to Python: the frame ([H, W, 3] or a batch [B, H, W, 3]) resized to
    W480 x H640 and converted to CHW float32 in [0, 1] (division by the double 255.0, rounded to float32)."""
    small = resize_u8_cv(bgr_u8, 480, 640)
    chw = (small.to(torch.float64) / 255.0).to(torch.float32)
    return chw.movedim(-1, -3).contiguous()


def fast_base_transform(img_hwc_f32):
    """[H, W, 3] (or [B, H, W, 3]) float32 BGR in 0..255 -> [B, 3, 550, 550] normalised RGB."""
    dev = img_hwc_f32.device
    if img_hwc_f32.dim() == 3:
        img_hwc_f32 = img_hwc_f32.unsqueeze(0)
    x = img_hwc_f32.permute(0, 3, 1, 2).contiguous()
    x = F.interpolate(x, (550, 550), mode="bilinear", align_corners=False)
    mean = torch.tensor(MEANS, dtype=torch.float32, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32, device=dev).view(1, 3, 1, 1)
    x = (x - mean) / std
    return x[:, (2, 1, 0), :, :].contiguous()
