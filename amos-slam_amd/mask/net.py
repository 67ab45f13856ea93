"""YOLACT-ResNet50-FPN (the reference's `yolact_resnet50` configuration, data/config.py:741-812)
as one plain nn.Module, inference only.

Architecture: ResNet-50 trunk -> C3,C4,C5 -> FPN with 256 features and two
stride-2 conv downsamples (P3..P7) -> protonet on P3 (3x conv3x3, bilinear x2, conv3x3, conv1x1 ->
32 prototypes at 138x138, ReLU) and ONE prediction head shared by the five levels (conv3x3 "upfeature"
then 3x3 convs for 3 anchors x {4 box, 81 class, 32 coefficient} values; tanh on coefficients).
Module and parameter names follow the reference's state dict (backbone.layers.N.M.convK,
fpn.lat_layers.N, proto_net.{0,2,4,8,10}, prediction_layers.0.{upfeature.0,bbox_layer,conf_layer,
mask_layer}, semantic_seg_conv) so `yolact_resnet50_54_800000.pth` loads unchanged.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_CLASSES = 81        # 80 COCO classes + background, data/config.py:660
MASK_DIM = 32           # prototypes, config mask_proto_net last entry
FPN_FEATURES = 256      # fpn_base.num_features
CONF_THRESH = 0.05      # Detect's class-confidence threshold (detect.py CONF_THRESH, layers/functions/detection.py:27)
MAX_SIZE = 550          # yolact_base_config.max_size
PRED_SCALES = (24, 48, 96, 192, 384)   # one scale per pyramid level
ASPECT_RATIOS = (1.0, 0.5, 2.0)        # pred_aspect_ratios; anchors are squares (use_square_anchors)


def _gemm_conv(conv, x):
    """Should this convolution run on the project's fp32 MFMA (implicit) GEMM (amos_mask_conv_device: bias, residual and ReLU in its
    epilogue) rather than MIOpen + the epilogue pass?  Measured per shape on MI355X at 32 frames (tools/conv1x1_probe.py,
    tools/conv_gemm_probe.py).  1 x 1: the GEMM wins 1.03 - 1.56 x on every layer whose launch has >= 600
    work-groups (of 128 x 128 outputs, or of 128 x 64 where the library switches to those); MIOpen's assembly kernels keep the
    2048 -> 256 lateral at 18 x 18 (0.88 x: 324 groups leave CUs idle).  3 x 3: 1.02 - 1.04 x on the layers with >= 1024 work-groups, MIOpen ahead on the smaller ones (0.7 - 0.9 x).
    AMOS_MASK_CONV1X1=0 / 1 and AMOS_MASK_CONV3X3=0 / 2 force a side (experiments, tests)."""
    k = conv.kernel_size
    if k[0] != k[1] or conv.padding[0] != conv.padding[1] or conv.dilation != (1, 1) or conv.groups != 1 or conv.stride[0] != conv.stride[1]:
        return False
    from .. import mask_conv_supported
    if not mask_conv_supported(conv.in_channels, conv.out_channels, k[0], k[1], conv.stride[0], conv.padding[0]):
        return False
    s = conv.stride[0]
    m = x.shape[0] * ((x.shape[2] + 2 * conv.padding[0] - k[0]) // s + 1) * ((x.shape[3] + 2 * conv.padding[1] - k[1]) // s + 1)
    groups = ((m + 127) // 128) * (conv.out_channels // (128 if conv.out_channels % 128 == 0 else 64))
    if groups < 1024 and (k, conv.padding) == ((1, 1), (0, 0)):
        groups = ((m + 127) // 128) * (conv.out_channels // 64)  # the library then runs 128 x 64 tiles (amos_mask_conv_device)
    if k != (1, 1) or conv.padding != (0, 0):
        mode3 = os.environ.get("AMOS_MASK_CONV3X3", "1")  # "0" never, "1" by the rule, "2" wherever the kernel applies (tests)
        if groups < 1024 and s >= 2:
            # strided layers (the stride-1 ones belong to the Winograd kernel): the 128 x 64 tiles the library then runs tie with MIOpen + the
            # epilogue pass from 1 024 work-groups on (256 ch, 69 -> 35 at 32 frames: 0.417 against 0.422 ms, tools/conv_gemm_probe.py)
            groups = ((m + 127) // 128) * (conv.out_channels // 64)
        # (one frame per pass: the 64-channel 3 x 3 layers at 138 x 138 -- 149 groups of 18 k-stages, too few tiles for the Winograd rule --
        # run 25.9 us here against 32.1 for the library kernel + the bias pass, tools/r5_small_gemm_probe.py)
        small = s == 1 and conv.in_channels <= 64 and groups >= 128
        return mode3 != "0" and (groups >= 1024 or small or mode3 == "2") and conv.weight.is_contiguous(memory_format=torch.channels_last)
    mode = os.environ.get("AMOS_MASK_CONV1X1", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    # small launches (one frame per pass; tools/r5_small_gemm_probe.py, 128 x 64 tiles): with at most 512 input channels (16 k-stages) the
    # GEMM with its fused epilogue beats the library kernel + the bias pass from ~100 work-groups on (64 -> 256 + residual at 138 x 138:
    # 16.6 against 24.4 us, 256 -> 1024 + residual at 35 x 35: 13.7 / 17.0, 512 -> 256 at 69 x 69: 21.6 / 25.1); deeper k with so few
    # groups stays with the library (1024 -> 256 at 35 x 35, 40 groups: 35.9 / 19.7 -- and 27 us cut along k, amos_mask_conv_ws_device)
    return groups >= 600 or (groups >= 100 and conv.in_channels <= 512)


def winograd_rule(cin, cout, kernel, stride, padding, dilation, groups, batch, height, width):
    """Does a convolution of this shape run as Winograd (F(2 x 4, 3 x 3) or F(2 x 2, 3 x 3): winograd_family) on the fp32 MFMA units
    (amos_mask_winograd24_conv_device / amos_mask_winograd_conv_device)?  3 x 3, stride 1, pad 1, channel counts the kernel takes, an
    input below 2 GiB, and a launch of at least 256 work-groups of 256 output pixels x 64 channels (one per CU).  Measured on MI355X at 32 frames (tools/winograd_probe.py): 1.4 - 1.8 x the direct implicit GEMM on every
    such layer of the network.  AMOS_MASK_WINOGRAD=0 never, 1 by this rule (default), 2 wherever the kernel applies (tests).
    (bench.py asks the same function which layers to count at the Winograd form's multiplies -- 24 per 2 x 4 outputs for F(2 x 4), 16 per
    2 x 2 for F(2 x 2) -- instead of the direct convolution's 9 per output.)"""
    mode = os.environ.get("AMOS_MASK_WINOGRAD", "1")
    if mode == "0" or tuple(kernel) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    from .. import mask_winograd_supported
    if not mask_winograd_supported(cin, cout) or batch * height * width * cin * 4 >= 2 ** 31 - 4096:
        return False
    # work-groups of 256 output pixels (64 tiles of 2 x 2; F(2 x 4)'s 32 tiles of 2 x 4 give the same count within a few per cent, and the
    # same layers run in either family: the threshold was measured per layer, tools/winograd_probe.py)
    n_groups = (batch * ((height + 1) // 2) * ((width + 1) // 2) + 63) // 64
    wg64 = n_groups * (cout // 64)
    if mode == "2" or wg64 >= 256:
        return True
    if winograd_family() != "24":   # (the small-launch form exists for the F(2 x 4) kernel only)
        return False
    # Small launches (one frame per pass).  A work-group's time is its stage count x the latency of a stage (1.85 us with 64 output channels per
    # group, 1.4 us with 32: amos_mask_winograd24_conv_device takes 32 while 64-channel groups are at most 100), whatever the layer's size as long
    # as every group has a CU; the library's direct kernel + the bias pass cost ~10 us + the direct FLOPs at ~85 TFLOP/s.  Measured at one
    # frame (tools/r5_small_gemm_probe.py): 64 ch at 138 x 138 15 against 32 us, 128 ch at 69 x 69 23 / 31, 256 ch at 69 x 69 45 / 68,
    # 256 -> 384 at 69 x 69 62 / 91; the 256-channel layers at 35 x 35 and below lose (44 / 30) and stay with the library.
    narrow = wg64 <= 100
    wgs = n_groups * (cout // 32) if narrow else wg64
    t_winograd = (cin / 8.0) * (1.4 if narrow else 1.85) * ((wgs + 255) // 256)
    t_library = 10.0 + 18.0 * cin * cout * batch * height * width / 85e6
    return t_winograd < 0.9 * t_library


def _winograd_conv(conv, x):
    return winograd_rule(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups,
                         x.shape[0], x.shape[2], x.shape[3])


def winograd_family():
    """Which Winograd form the eligible layers run in: "24" = F(2 x 4, 3 x 3) (amos_mask_winograd24_conv_device: 24 multiplies per 2 x 4
    outputs, the default) or "22" = F(2 x 2, 3 x 3) (amos_mask_winograd_conv_device: 16 per 2 x 2).  AMOS_MASK_WINOGRAD_F selects (A/B runs, tests)."""
    return "22" if os.environ.get("AMOS_MASK_WINOGRAD_F", "24") == "22" else "24"


def _winograd_fns():
    """(family, positions, weight transform, convolution) of the Winograd form in force"""
    if winograd_family() == "22":
        from .. import mask_winograd_weights, mask_winograd_conv
        return "22", 16, mask_winograd_weights, mask_winograd_conv
    from .. import mask_winograd24_weights, mask_winograd24_conv
    return "24", 24, mask_winograd24_weights, mask_winograd24_conv


def _winograd_weight(conv):
    """The layer's transformed weight G g G^T (16 or 24 x cin x cout floats in the kernel's staging layout), made on first use and kept on the
    module; remade when the weight tensor changes (another storage or an in-place update).  The cached tensor is shared by every caller
    on every stream, so the transform is a SYNCHRONOUS step: the making stream is drained before the tensor is published (callers on
    other streams would otherwise read it with nothing ordering them after the transform kernel).  MaskEngine.prepare() transforms all
    eligible layers up front (prepare_winograd_weights), so in steady state this is a dictionary hit."""
    w = conv.weight
    family, positions, make_weights, _ = _winograd_fns()
    key = (w.data_ptr(), w._version, str(w.device), family)
    cached = getattr(conv, "_amos_winograd", None)
    if cached is None or cached[0] != key:
        if torch.cuda.is_current_stream_capturing():
            # a tensor allocated and filled inside a capture belongs to that graph's private pool and is filled only on replay: it must
            # never be published on the module for eager passes or other graphs to read
            raise RuntimeError("a Winograd weight would be created inside a HIP-graph capture: call MaskEngine.prepare() (or "
                               "prepare_winograd_weights) before capture_graph / frame_session, and again after changing weights or AMOS_MASK_WINOGRAD_F")
        wl = w.detach().contiguous(memory_format=torch.channels_last)  # [cout][3][3][cin] in memory
        u = torch.empty(positions * conv.in_channels * conv.out_channels, dtype=torch.float32, device=w.device)
        stream = torch.cuda.current_stream(w.device)
        make_weights(stream.cuda_stream, wl.data_ptr(), u.data_ptr(), conv.in_channels, conv.out_channels)
        stream.synchronize()
        cached = (key, u)
        object.__setattr__(conv, "_amos_winograd", cached)
    return cached[1]


def stem_kernel_enabled():
    """AMOS_MASK_STEM=library: the stem as the library's convolution + the project's bias / ReLU / max-pool pass (A/B runs, tests); default: the
    project's one-kernel stem (amos_mask_stem_device)."""
    return os.environ.get("AMOS_MASK_STEM", "own") != "library"


def _stem_weight(conv):
    """The stem's 7 x 7 weight in amos_mask_stem_device's layout, made on first use and kept on the module like a Winograd weight (same
    rules: synchronous, never inside a graph capture; MaskEngine.prepare() makes it up front)."""
    from .. import mask_stem_weight_floats, mask_stem_weights
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), tuple(w.stride()))
    cached = getattr(conv, "_amos_stem", None)
    if cached is None or cached[0] != key:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the stem's packed weight would be created inside a HIP-graph capture: call MaskEngine.prepare() (or "
                               "prepare_stem_weight) before capture_graph / frame_session, and again after changing weights")
        packed = torch.empty(mask_stem_weight_floats(), dtype=torch.float32, device=w.device)
        stream = torch.cuda.current_stream(w.device)
        mask_stem_weights(stream.cuda_stream, w.data_ptr(), w.stride(), packed.data_ptr())
        stream.synchronize()
        cached = (key, packed)
        object.__setattr__(conv, "_amos_stem", cached)
    return cached[1]


def _stem_eligible(conv):
    return (conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.in_channels == 3 and conv.out_channels == 64 and conv.bias is not None and conv.weight.is_cuda
            and conv.weight.dtype == torch.float32)


def prepare_winograd_weights(module):
    """Transforms the weights of every convolution of `module` the Winograd kernel can take (3 x 3, stride 1, pad 1, supported channel
    counts -- whether a launch then USES the kernel still depends on its size, winograd_rule) and waits for them: after this no forward
    on any stream creates a shared tensor.  Returns the number of layers transformed."""
    from .. import mask_winograd_supported
    convs = [m for m in module.modules() if isinstance(m, nn.Conv2d)]
    merged = [getattr(m, "merged", None) for m in module.modules()]
    n = 0
    for conv in convs + [m for m in merged if isinstance(m, nn.Conv2d)]:
        if (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.weight.is_cuda and conv.weight.dtype == torch.float32 and mask_winograd_supported(conv.in_channels, conv.out_channels)):
            _winograd_weight(conv)
            n += 1
    return n


def prepare_stem_weight(module):
    """Packs the stem weight of every ResNet50Trunk of `module` for amos_mask_stem_device (and waits): after this no forward creates it."""
    n = 0
    for m in module.modules():
        if isinstance(m, ResNet50Trunk) and _stem_eligible(m.conv1) and stem_kernel_enabled():
            _stem_weight(m.conv1)
            n += 1
    return n


class Blocked:
    """A float32 activation in the channel-blocked layout [b][c / 8][h][w][8] (`data`), with its logical NCHW `shape`.  The F(2 x 4) Winograd
    kernel reads 8 input channels of every pixel of its patch per stage: blocked, those are contiguous whole cache lines used once;
    channels-last, a quarter of every line four times (amos_mask_winograd24_conv_layout_device; same bits, 1.5 - 3.6 % faster per layer at 64
    frames).  Only a chain of layers that ALL run on that kernel can hand such a tensor on: the pyramid's level 0 -> the prototype network
    and level 0's prediction head (YolactR50.forward decides, blocked_chain_for)."""
    __slots__ = ("data", "shape")
    is_cuda = True
    dtype = torch.float32

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)

    @property
    def device(self):
        return self.data.device


def _winograd24_layout(conv, x, bias, residual, relu, out_blocked):
    """One F(2 x 4) launch with either layout on either side; x: Blocked or a channels-last tensor."""
    from .. import mask_winograd24_conv_layout
    in_blocked = isinstance(x, Blocked)
    b, _, h, w = x.shape
    dev = x.device
    if out_blocked:
        y = Blocked(torch.empty((b, conv.out_channels // 8, h, w, 8), device=dev, dtype=torch.float32), (b, conv.out_channels, h, w))
    else:
        y = torch.empty((b, conv.out_channels, h, w), device=dev, dtype=torch.float32, memory_format=torch.channels_last)
    mask_winograd24_conv_layout(torch.cuda.current_stream(dev).cuda_stream, (x.data if in_blocked else x).data_ptr(), _winograd_weight(conv).data_ptr(),
                                bias.data_ptr() if bias is not None else None, residual.data_ptr() if residual is not None else None,
                                (y.data if out_blocked else y).data_ptr(), b, h, w, conv.in_channels, conv.out_channels, relu, in_blocked, out_blocked)
    return y


def blocked_chain_for(p3_in, convs_69, convs_138):
    """May level 0 of the pyramid (made from p3_in, [b, 256, 69, 69]) travel channel-blocked through the prototype network and its prediction head?
    Every layer on the way must run on the F(2 x 4) Winograd kernel at this launch size (the wide form: at least 8 frames), float32, on the GPU.
    AMOS_MASK_BLOCKED_CHAIN=0: never (A/B runs, tests)."""
    if os.environ.get("AMOS_MASK_BLOCKED_CHAIN", "1") == "0" or winograd_family() != "24" or torch.is_autocast_enabled():
        return False
    if not (p3_in.is_cuda and p3_in.dtype == torch.float32 and p3_in.shape[0] >= 8 and p3_in.is_contiguous(memory_format=torch.channels_last)):
        return False
    b, _, h, w = p3_in.shape
    ok = all(c.bias is not None and c.weight.dtype == torch.float32 and c.in_channels % 8 == 0 and c.out_channels % 8 == 0 and
             winograd_rule(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.groups, b, h, w) for c in convs_69)
    return ok and all(c.bias is not None and c.weight.dtype == torch.float32 and
                      winograd_rule(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.groups, b, 2 * h, 2 * w) for c in convs_138)


def conv_raw(conv, x):
    """The convolution alone (no bias): Winograd / implicit GEMM of this project where they apply, else the library."""
    cl = torch.channels_last
    if isinstance(x, Blocked):
        return _winograd24_layout(conv, x, None, None, False, False)
    if x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and x.is_contiguous(memory_format=cl) and conv.weight.dtype == torch.float32:
        if _winograd_conv(conv, x):
            b, _, h, w = x.shape
            y = torch.empty((b, conv.out_channels, h, w), device=x.device, dtype=torch.float32, memory_format=cl)
            _winograd_fns()[3](torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), _winograd_weight(conv).data_ptr(), None, None, y.data_ptr(),
                               b, h, w, conv.in_channels, conv.out_channels, False)
            return y
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def conv_bias_act(conv, x, relu, residual=None, out_blocked=False):
    """conv -> + bias -> (+ residual) -> (ReLU).  On the GPU with float32 channels-last activations the bias, the residual
    and the ReLU are ONE in-place pass by a HIP kernel of this project (amos_mask_bias_act_device) behind MIOpen's
    convolution instead of PyTorch's three elementwise passes; the summation order is the same, so are the bits.  The large
    1 x 1 convolutions go to this project's MFMA GEMM with that epilogue fused (_gemm_conv; float32 rounding apart from MIOpen).
    Anywhere else (CPU tests, autocast) the plain torch ops run.  x Blocked / out_blocked: a link of the channel-blocked chain (Blocked)."""
    if isinstance(x, Blocked) or out_blocked:
        if residual is not None:
            raise ValueError("conv_bias_act: no residual in the channel-blocked chain")
        return _winograd24_layout(conv, x, conv.bias, None, relu, out_blocked)
    if x.is_cuda and conv.bias is not None and x.dtype == torch.float32 and not torch.is_autocast_enabled():
        cl = torch.channels_last
        if _winograd_conv(conv, x) and x.is_contiguous(memory_format=cl) and conv.weight.dtype == torch.float32 and (residual is None or (
                residual.dtype == torch.float32 and residual.is_contiguous(memory_format=cl))):
            b, _, h, w = x.shape
            y = torch.empty((b, conv.out_channels, h, w), device=x.device, dtype=torch.float32, memory_format=cl)
            if residual is not None and residual.shape != y.shape:
                raise ValueError("conv_bias_act: residual shape %s, output shape %s" % (tuple(residual.shape), tuple(y.shape)))
            _winograd_fns()[3](torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), _winograd_weight(conv).data_ptr(), conv.bias.data_ptr(),
                               residual.data_ptr() if residual is not None else None, y.data_ptr(), b, h, w, conv.in_channels, conv.out_channels, relu)
            return y
        if _gemm_conv(conv, x) and x.is_contiguous(memory_format=cl) and conv.weight.dtype == torch.float32 and (
                conv.weight.is_contiguous(memory_format=cl) or (conv.kernel_size == (1, 1) and conv.weight.is_contiguous())) and (residual is None or (
                residual.dtype == torch.float32 and residual.is_contiguous(memory_format=cl))):
            from .. import mask_conv
            b, _, h, w = x.shape
            s, k, p = conv.stride[0], conv.kernel_size[0], conv.padding[0]
            y = torch.empty((b, conv.out_channels, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1), device=x.device, dtype=torch.float32, memory_format=cl)
            if residual is not None and residual.shape != y.shape:
                raise ValueError("conv_bias_act: residual shape %s, output shape %s" % (tuple(residual.shape), tuple(y.shape)))
            mask_conv(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr(),
                      residual.data_ptr() if residual is not None else None, y.data_ptr(), b, h, w, conv.in_channels, conv.out_channels, k, k, s, p, relu)
            return y
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if y.dtype == torch.float32 and y.is_contiguous(memory_format=cl) and (residual is None or (
                residual.shape == y.shape and residual.dtype == torch.float32 and residual.is_contiguous(memory_format=cl))):
            from .. import mask_bias_act
            mask_bias_act(torch.cuda.current_stream(x.device).cuda_stream, y.data_ptr(), conv.bias.data_ptr(),
                          residual.data_ptr() if residual is not None else None, y.numel(), y.shape[1], relu)
            return y
        y = y + conv.bias.view(1, -1, 1, 1)
    else:
        y = conv(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def bilinear(x, size=None, scale_factor=None, relu=False):
    """F.interpolate(x, mode="bilinear", align_corners=False), then a ReLU when asked.  Float32 channels-last tensors on the GPU
    go through this project's HIP kernel (amos_mask_bilinear_nhwc_act_device: PyTorch's channels-last kernel was 13 % of the mask
    pass); the source index and the weights are computed as PyTorch computes them."""
    def geometry(h, w):  # output size and PyTorch's area_pixel_compute_scale, in float32 as PyTorch computes it
        if size is not None:
            oh, ow = int(size[0]), int(size[1])
            return oh, ow, float(torch.tensor(h, dtype=torch.float32) / oh), float(torch.tensor(w, dtype=torch.float32) / ow)
        s = float(torch.tensor(1.0, dtype=torch.float32) / scale_factor)
        return int(h * scale_factor), int(w * scale_factor), s, s

    if isinstance(x, Blocked):  # [b][c / 8][h][w][8] is b x c / 8 channels-last images of 8 channels: the same kernel, the same taps
        n, c, h, w = x.shape
        oh, ow, sh, sw = geometry(h, w)
        y = Blocked(torch.empty((n, c // 8, oh, ow, 8), dtype=torch.float32, device=x.device), (n, c, oh, ow))
        from .. import mask_bilinear_nhwc
        mask_bilinear_nhwc(torch.cuda.current_stream(x.device).cuda_stream, x.data.data_ptr(), y.data.data_ptr(), n * (c // 8), h, w, oh, ow, 8, sh, sw, relu)
        return y
    if x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last):
        n, c, h, w = x.shape
        oh, ow, sh, sw = geometry(h, w)
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        from .. import mask_bilinear_nhwc
        mask_bilinear_nhwc(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), y.data_ptr(), n, h, w, oh, ow, c, sh, sw, relu)
        return y
    if size is not None:
        y = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    else:
        y = F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False)
    return F.relu(y) if relu else y


def _fold(conv, bn):
    """conv followed by an inference-mode batch norm == one conv with scaled weights and a bias:
    w' = w * g / sqrt(var + eps),  b' = beta - mean * g / sqrt(var + eps)   (folded in float64, stored float32)."""
    with torch.no_grad():
        scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        w = (conv.weight.double() * scale.view(-1, 1, 1, 1)).to(conv.weight.dtype)
        b = bn.bias.double() - bn.running_mean.double() * scale
        if conv.bias is not None:
            b = b + conv.bias.double() * scale
        fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, bias=True)
        fused = fused.to(device=conv.weight.device, dtype=conv.weight.dtype)
        fused.weight.copy_(w)
        fused.bias.copy_(b.to(conv.weight.dtype))
    return fused


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, project=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        if self.conv1.bias is not None:  # batch norms folded: conv -> fused (bias, residual, ReLU)
            y = conv_bias_act(self.conv1, x, True)
            y = conv_bias_act(self.conv2, y, True)
            identity = x if self.downsample is None else conv_bias_act(self.downsample[0], x, False)
            return conv_bias_act(self.conv3, y, True, residual=identity)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))

    def fold_batch_norms(self):
        for k in (1, 2, 3):
            setattr(self, f"conv{k}", _fold(getattr(self, f"conv{k}"), getattr(self, f"bn{k}")))
            setattr(self, f"bn{k}", nn.Identity())
        if self.downsample is not None:
            self.downsample = nn.Sequential(_fold(self.downsample[0], self.downsample[1]), nn.Identity())


class ResNet50Trunk(nn.Module):
    """backbone.py:60-125 with args ([3, 4, 6, 3],); returns the four stage outputs."""

    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layers = nn.ModuleList()
        self.channels = []
        inplanes = 64
        for stage, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks)):
            stride = 1 if stage == 0 else 2
            seq = [Bottleneck(inplanes, planes, stride, project=True)]
            inplanes = planes * 4
            seq += [Bottleneck(inplanes, planes) for _ in range(n - 1)]
            self.layers.append(nn.Sequential(*seq))
            self.channels.append(inplanes)

    def fold_batch_norms(self):
        self.conv1 = _fold(self.conv1, self.bn1)
        self.bn1 = nn.Identity()
        for layer in self.layers:
            for block in layer:
                block.fold_batch_norms()

    def forward(self, x):
        cl = torch.channels_last
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and x.dim() == 4 and x.shape[1] == 3 and x.shape[0] <= 65535
                and _stem_eligible(self.conv1) and stem_kernel_enabled()):
            # convolution + bias + ReLU + max-pool as ONE kernel of this project: the input through its strides (planar as the pre-processing
            # writes it, or channels-last), the 275 x 275 x 64 convolution output never in memory
            from .. import mask_stem
            b, _, h, w = x.shape
            ch, cw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            y = torch.empty((b, 64, (ch - 1) // 2 + 1, (cw - 1) // 2 + 1), dtype=torch.float32, device=x.device, memory_format=cl)
            mask_stem(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), x.stride(), _stem_weight(self.conv1).data_ptr(), self.conv1.bias.data_ptr(),
                      y.data_ptr(), b, h, w)
            x = y
        elif self.conv1.bias is not None and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
            # raw convolution, then bias + ReLU + max-pool in one pass of a HIP kernel (bit-identical to the separate passes)
            raw = F.conv2d(x, self.conv1.weight, None, self.conv1.stride, self.conv1.padding)
            if raw.is_contiguous(memory_format=cl) and raw.shape[1] % 4 == 0:
                from .. import mask_bias_relu_maxpool
                b, c, h, w = raw.shape
                y = torch.empty((b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device, memory_format=cl)
                mask_bias_relu_maxpool(torch.cuda.current_stream(x.device).cuda_stream, raw.data_ptr(), self.conv1.bias.data_ptr(), y.data_ptr(), b, h, w, c)
                x = y
            else:
                x = F.max_pool2d(F.relu(raw + self.conv1.bias.view(1, -1, 1, 1)), 3, stride=2, padding=1)
        else:
            x = conv_bias_act(self.conv1, x, True) if self.conv1.bias is not None else F.relu(self.bn1(self.conv1(x)))
            x = F.max_pool2d(x, 3, stride=2, padding=1)
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return outs


_SIDE_STREAMS = {}


class Branches:
    """Independent parts of a SMALL pass on side streams.  One frame leaves most of the chip idle in every launch after the first stages (the
    35 x 35 and smaller layers run 5 - 80 work-groups on 256 CUs) while the pass is a chain of ~150 launches; what does not depend on each other
    -- the prototype network against the prediction head's five levels, the pyramid's three output convolutions -- can run side by side.  Inside a
    HIP-graph capture the side streams become parallel branches of the graph (mask/interface.py FrameSession).  Same kernels on the same
    operands: the same bits as the one-stream pass.  Measured on MI355X, one frame, graph replay: 2.11 -> 1.98 - 2.02 ms with ONE side stream
    (the default: the pyramid's levels 1 - 4 and their heads there, level 0 + its head + the prototype network on the main stream); with two
    or three side streams 2.23 - 2.27 ms -- every further branch of a HIP graph costs more than it hides (AMOS_MASK_BRANCH_STREAMS).

    Rules kept here: a side stream starts behind everything queued on the main stream so far (`side(i)`, or continues its own work with
    fork=False); every tensor that crosses streams is held in `keep` until `join()`, so no block returns to an allocator pool while another
    stream's queued kernels still read it; `join()` puts the main stream behind all side streams."""

    def __init__(self, device, n=1):
        self.device = device
        self.main = torch.cuda.current_stream(device)
        key = (str(device), self.main.cuda_stream)  # side streams belong to ONE main stream: two lanes never meet on (or capture) the same one
        if key not in _SIDE_STREAMS:
            if len(_SIDE_STREAMS) >= 64:  # (a host that keeps making main streams: do not keep a side stream for every one ever seen)
                _SIDE_STREAMS.clear()
            _SIDE_STREAMS[key] = []
        while len(_SIDE_STREAMS[key]) < n:
            _SIDE_STREAMS[key].append(torch.cuda.Stream(device))
        self.streams = _SIDE_STREAMS[key][:n]
        self.used = set()
        self.keep = []

    def side(self, i, fork=True):
        """Context of side stream i (with fewer streams than roles: i modulo their number; None -> the main stream itself)."""
        if i is None:
            return torch.cuda.stream(self.main)
        i %= len(self.streams)
        s = self.streams[i]
        if fork or i not in self.used:
            s.wait_stream(self.main)
        self.used.add(i)
        return torch.cuda.stream(s)

    def hold(self, *tensors):
        self.keep.extend(tensors)

    def join(self):
        for i in sorted(self.used):
            self.main.wait_stream(self.streams[i])
        self.used.clear()
        self.keep = []


def branches_for(x):
    """AMOS_MASK_BRANCHES: 0 never, 1 always (on a GPU), default: passes of at most AMOS_MASK_BRANCH_MAX_BATCH (16) frames -- measured as one
    HIP graph (tools/r5_small_batch_branches.py): 1 frame 2.11 -> 1.98 ms, 2 frames 3.00 -> 2.78, 4 frames 4.39 -> 4.12, 8 frames 6.93 -> 6.62,
    16 frames 11.73 -> 11.48; at the bench's 64 frames per pass the side stream changes nothing (1 545 / 1 543 frames/s)."""
    mode = os.environ.get("AMOS_MASK_BRANCHES", "auto")
    if mode == "0" or not x.is_cuda:
        return None
    if mode != "1" and x.shape[0] > int(os.environ.get("AMOS_MASK_BRANCH_MAX_BATCH", "16")):
        return None
    return Branches(x.device, int(os.environ.get("AMOS_MASK_BRANCH_STREAMS", "1")))


class FeaturePyramid(nn.Module):
    """yolact.py:265-355 with fpn = {256 features, bilinear, 2 conv downsamples, pad, relu on pred layers}."""

    def __init__(self, in_channels):
        super().__init__()
        self.lat_layers = nn.ModuleList([nn.Conv2d(c, FPN_FEATURES, 1) for c in reversed(in_channels)])
        self.pred_layers = nn.ModuleList([nn.Conv2d(FPN_FEATURES, FPN_FEATURES, 3, padding=1) for _ in in_channels])
        self.downsample_layers = nn.ModuleList([nn.Conv2d(FPN_FEATURES, FPN_FEATURES, 3, padding=1, stride=2) for _ in range(2)])

    def forward(self, feats, br=None, p3_blocked=False):
        """p3_blocked: level 0 leaves channel-blocked (Blocked; YolactR50.forward asked blocked_chain_for).  br (Branches): the output convolutions of the deeper levels and the two extra levels run on side streams (level 2 + P6 + P7 on side
        0, level 1 on side 1, level 0 stays on the caller's stream); the caller joins."""
        n = len(feats)
        merged = [None] * n
        top = None
        for k, lat in enumerate(self.lat_layers):  # deepest level first
            j = n - 1 - k
            # lateral convolution + bias + the upsampled level above (the residual of the fused epilogue: same order of sums)
            up = bilinear(top, size=feats[j].shape[2:]) if top is not None else None
            merged[j] = top = conv_bias_act(lat, feats[j], False, residual=up)
        outs = [None] * n
        if br is not None and n == 3:
            br.hold(*merged)
            with br.side(0):
                outs[2] = conv_bias_act(self.pred_layers[0], merged[2], True)
                extra = []
                for down in self.downsample_layers:
                    extra.append(down(extra[-1] if extra else outs[2]))
            with br.side(1, fork=len(br.streams) > 1):  # (one side stream: it continues there, no second dependency on the main stream)
                outs[1] = conv_bias_act(self.pred_layers[1], merged[1], True)
            outs[0] = conv_bias_act(self.pred_layers[2], merged[0], True, out_blocked=p3_blocked)
            outs += extra
            br.hold(*outs)
            return outs
        for k, pred in enumerate(self.pred_layers):
            j = n - 1 - k
            outs[j] = conv_bias_act(pred, merged[j], True, out_blocked=p3_blocked and j == 0)
        for down in self.downsample_layers:
            outs.append(down(outs[-1]))
        return outs


class SharedHead(nn.Module):
    """prediction_layers.0 of the reference (yolact.py:47-201): every level runs these weights."""

    def __init__(self, num_priors=len(ASPECT_RATIOS)):
        super().__init__()
        self.upfeature = nn.Sequential(nn.Conv2d(FPN_FEATURES, FPN_FEATURES, 3, padding=1), nn.ReLU(inplace=True))
        self.bbox_layer = nn.Conv2d(FPN_FEATURES, num_priors * 4, 3, padding=1)
        self.conf_layer = nn.Conv2d(FPN_FEATURES, num_priors * NUM_CLASSES, 3, padding=1)
        self.mask_layer = nn.Conv2d(FPN_FEATURES, num_priors * MASK_DIM, 3, padding=1)

    def merge_output_layers(self):
        """Inference form: the three output convolutions read the same tensor, so they run as ONE 3 x 3 convolution with
        12 + 243 + 96 (+ 33 zero filters: 384 = a multiple of 64 for this project's convolution kernels) output channels -- one pass over the
        upfeature tensor and one launch instead of three, with better output-channel tiles than 12 or 243 give (measured
        on MI355X, 32 frames at 69 x 69: 2.09 ms against 0.40 + 1.63 + 0.78).  Same sums per channel; MIOpen may pick
        another solver for the wider layer, hence float32 rounding.  Built from the loaded weights: call after
        load_state_dict.  Idempotent."""
        if getattr(self, "merged", None) is None:
            layers = (self.bbox_layer, self.conf_layer, self.mask_layer)
            n = sum(l.out_channels for l in layers)
            pad = (-n) % 64  # 351 -> 384: a multiple of 64, which the project's Winograd / GEMM kernels take (zero filters: 9 % more work than 352)
            merged = nn.Conv2d(FPN_FEATURES, n + pad, 3, padding=1).to(device=self.bbox_layer.weight.device, dtype=self.bbox_layer.weight.dtype)
            with torch.no_grad():
                merged.weight.zero_()
                merged.bias.zero_()
                merged.weight[:n].copy_(torch.cat([l.weight for l in layers], 0))
                merged.bias[:n].copy_(torch.cat([l.bias for l in layers], 0))
            # not a registered sub-module: the state dict keeps the reference's keys
            object.__setattr__(self, "merged", merged.requires_grad_(False))
        return self

    def fused_applies(self, x0):
        merged = getattr(self, "merged", None)
        return not (merged is None or not x0.is_cuda or x0.dtype != torch.float32 or torch.is_autocast_enabled() or os.environ.get("AMOS_MASK_FUSED_HEAD", "1") == "0")

    def fused_outputs(self, pyramid, n_priors, br=None, out=None, scores=None):
        """All levels' outputs, concatenated, softmax / tanh applied: (loc [B, P, 4], conf [B, P, 81], coef [B, P, 32]), or None when the
        fused path does not apply (no merged layer, not float32 channels-last on the GPU).  Per level: the upfeature convolution, the
        merged output convolution WITHOUT its bias, then ONE HIP kernel (amos_mask_head_outputs_device) that adds the bias, takes the
        softmax and the tanh and writes the level's priors into the three concatenated tensors -- instead of a bias pass, three
        strided reshape copies per level, three concatenations, a softmax and a tanh pass.
        br (Branches): the levels run on the side streams that made their inputs (FeaturePyramid.forward: levels 2 - 4 on side 0, level 1 on
        side 1) and level 0 on side 2, so the caller's stream is free for the prototype network; `out` = the three tensors, allocated by the
        caller BEFORE any side stream started (they are written there and read on the caller's stream after its join).
        scores: a [B, classes, P] tensor (alloc_scores) -> Detect's class scores come out of the same kernel (amos_mask_head_outputs_scores_device:
        what amos_mask_class_scores_device would make of conf, bit for bit); with out[1] None the softmax tensor itself is not written."""
        x0 = pyramid[0]
        if not self.fused_applies(x0):
            return None
        merged = self.merged
        from .. import mask_head_outputs_scores
        b, dev = x0.shape[0], x0.device
        n_anchor = self.bbox_layer.out_channels // 4
        if out is None:
            if br is not None:
                raise ValueError("fused_outputs: with side streams the caller allocates the outputs")
            out = self.alloc_outputs(b, n_priors, dev)
        loc, conf, coef = out
        offs, off = [], 0
        for x in pyramid:
            offs.append(off)
            off += x.shape[2] * x.shape[3] * n_anchor
        if off != n_priors:
            raise ValueError("fused_outputs: %d priors written, %d expected" % (off, n_priors))

        def level(i):
            x = pyramid[i]
            u = conv_bias_act(self.upfeature[0], x, True, out_blocked=isinstance(x, Blocked))
            raw = conv_raw(merged, u)
            if not raw.is_contiguous(memory_format=torch.channels_last):
                raw = raw.contiguous(memory_format=torch.channels_last)
            cells = raw.shape[2] * raw.shape[3]
            mask_head_outputs_scores(torch.cuda.current_stream(dev).cuda_stream, raw.data_ptr(), merged.bias.data_ptr(), loc.data_ptr(),
                                     conf.data_ptr() if conf is not None else None, coef.data_ptr(), scores.data_ptr() if scores is not None else None,
                                     CONF_THRESH, b, cells, raw.shape[1], n_anchor, NUM_CLASSES, MASK_DIM, n_priors, offs[i])

        if br is not None and len(pyramid) == 5:
            with br.side(2 if len(br.streams) >= 3 else None):  # behind the caller's stream: level 0 was made there
                level(0)
            with br.side(1, fork=False):
                level(1)
            with br.side(0, fork=False):
                for i in (2, 3, 4):
                    level(i)
        else:
            for i in range(len(pyramid)):
                level(i)
        return loc, conf, coef

    def alloc_outputs(self, b, n_priors, dev, conf=True):
        return (torch.empty((b, n_priors, 4), dtype=torch.float32, device=dev),
                torch.empty((b, n_priors, NUM_CLASSES), dtype=torch.float32, device=dev) if conf else None,
                torch.empty((b, n_priors, MASK_DIM), dtype=torch.float32, device=dev))

    def alloc_scores(self, b, n_priors, dev):
        return torch.empty((b, NUM_CLASSES - 1, n_priors), dtype=torch.float32, device=dev)

    def forward(self, x):
        b = x.shape[0]
        x = conv_bias_act(self.upfeature[0], x, True)
        merged = getattr(self, "merged", None)
        if merged is not None:
            y = conv_bias_act(merged, x, False).permute(0, 2, 3, 1)  # [b, h, w, 384]: channels last in memory, so this is a view
            n0, n1, n2 = self.bbox_layer.out_channels, self.conf_layer.out_channels, self.mask_layer.out_channels
            loc = y[..., :n0].reshape(b, -1, 4)
            conf = y[..., n0:n0 + n1].reshape(b, -1, NUM_CLASSES)
            coef = torch.tanh(y[..., n0 + n1:n0 + n1 + n2].reshape(b, -1, MASK_DIM))
            return loc, conf, coef
        loc = self.bbox_layer(x).permute(0, 2, 3, 1).reshape(b, -1, 4)
        conf = self.conf_layer(x).permute(0, 2, 3, 1).reshape(b, -1, NUM_CLASSES)
        coef = torch.tanh(self.mask_layer(x).permute(0, 2, 3, 1).reshape(b, -1, MASK_DIM))
        return loc, conf, coef


class _NoParams(nn.Module):
    """Levels 1..4 own no weights (they point at level 0): keeps the ModuleList length of the reference."""


class _Upsample2x(nn.Module):
    def forward(self, x):
        return bilinear(x, scale_factor=2)


def build_priors(conv_sizes, device="cpu"):
    """Anchor boxes [cx, cy, w, h] (relative), yolact.py:203-247: for every cell (row-major), every
    aspect ratio: w = scale * sqrt(ar) / 550, h = w (square anchors kept for weight compatibility)."""
    rows = []
    for (h, w), scale in zip(conv_sizes, PRED_SCALES):
        for j in range(h):
            for i in range(w):
                x, y = (i + 0.5) / w, (j + 0.5) / h
                for ar in ASPECT_RATIOS:
                    side = scale * math.sqrt(ar) / MAX_SIZE
                    rows.append((x, y, side, side))
    return torch.tensor(rows, dtype=torch.float32, device=device)


def _outputs(loc, conf, cls, coef, priors, proto):
    out = {"loc": loc, "mask": coef, "priors": priors, "proto": proto}
    if conf is not None:
        out["conf"] = conf
    if cls is not None:
        out["cls"] = cls
    return out


class YolactR50(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = ResNet50Trunk()
        self.fpn = FeaturePyramid(self.backbone.channels[1:])
        c = FPN_FEATURES
        self.proto_net = nn.Sequential(
            nn.Conv2d(c, c, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, c, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c, c, 3, padding=1), nn.ReLU(inplace=True), _Upsample2x(), nn.ReLU(inplace=True),
            nn.Conv2d(c, c, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, MASK_DIM, 1))
        self.prediction_layers = nn.ModuleList([SharedHead()] + [_NoParams() for _ in range(4)])
        self.semantic_seg_conv = nn.Conv2d(c, NUM_CLASSES - 1, 1)  # training-only head; kept so the .pth loads strictly
        self._prior_cache = {}

    def load_weights(self, path, map_location="cpu"):
        """Accepts the reference's checkpoints (yolact.py:477-490: legacy `backbone.layer*` and surplus
        `fpn.downsample_layers.N` entries are dropped)."""
        sd = torch.load(path, map_location=map_location)
        for key in list(sd.keys()):
            if key.startswith("backbone.layer") and not key.startswith("backbone.layers"):
                del sd[key]
            elif key.startswith("fpn.downsample_layers.") and int(key.split(".")[2]) >= 2:
                del sd[key]
        return self.load_state_dict(sd, strict=False)

    def fold_batch_norms(self):
        """Inference form: every batch norm of the trunk becomes part of its convolution (53 fewer elementwise
        passes over the activations).  After this the module no longer matches the checkpoint's key layout:
        load the weights FIRST.  Idempotent."""
        if not getattr(self, "_folded", False):
            self.backbone.fold_batch_norms()
            self._folded = True
        return self

    def merge_head_outputs(self):
        self.prediction_layers[0].merge_output_layers()
        return self

    def _apply(self, fn, *args, **kwargs):  # .to(device / memory format) after the merge carries the merged layer along
        super()._apply(fn, *args, **kwargs)
        merged = getattr(self.prediction_layers[0], "merged", None)
        if merged is not None:
            merged._apply(fn, *args, **kwargs)
        return self

    def forward(self, x, scores_only=False):
        """x: [B, 3, 550, 550] normalised RGB.  Returns raw network outputs (before Detect):
        loc [B, P, 4], conf [B, P, 81] (softmax), mask [B, P, 32] (tanh), priors [P, 4], proto [B, 138, 138, 32] (ReLU).
        scores_only (the detector's own passes, where the fused head applies): instead of conf, "cls" [B, 80, P] -- Detect's thresholded,
        transposed class scores, written by the head's output kernel (fused_outputs); mask/post.py takes either."""
        feats = self.backbone(x)[1:]
        head = self.prediction_layers[0]
        pn = self.proto_net  # conv, relu, conv, relu, conv, relu, upsample, relu, conv, relu, conv (+ the final ReLU)

        def prototypes(p3):
            blk = isinstance(p3, Blocked)  # the channel-blocked chain: blocked through the three 69 x 69 layers and the resize, channels-last out of the 138 x 138 layer
            p = conv_bias_act(pn[4], conv_bias_act(pn[2], conv_bias_act(pn[0], p3, True, out_blocked=blk), True, out_blocked=blk), True, out_blocked=blk)
            p = conv_bias_act(pn[10], conv_bias_act(pn[8], bilinear(p, scale_factor=2, relu=True), True), True)  # pn[6] (upsample) + pn[7] (ReLU) in one pass
            return p.permute(0, 2, 3, 1).contiguous()

        def priors_of(pyramid):
            sizes = tuple(tuple(p.shape[2:]) for p in pyramid)
            key = (sizes, str(x.device))
            if key not in self._prior_cache:
                self._prior_cache[key] = build_priors(sizes, x.device)
            return self._prior_cache[key]

        # level 0 of the pyramid channel-blocked through the prototype network and its head, where every layer on the way is an F(2 x 4) launch
        chain = (len(feats) == 3 and head.fused_applies(feats[0]) and getattr(head, "merged", None) is not None and
                 blocked_chain_for(feats[0], (self.fpn.pred_layers[2], pn[0], pn[2], pn[4], head.upfeature[0], head.merged), (pn[8],)))
        br = branches_for(x) if head.fused_applies(feats[0]) and len(feats) == 3 else None
        if br is not None:
            # a small pass: the pyramid's side levels and the prediction head on side streams, the prototype network on this one (Branches)
            sizes = [tuple(f.shape[2:]) for f in feats]
            for _ in range(2):  # the two extra levels: 3 x 3, stride 2, padding 1
                sizes.append(((sizes[-1][0] - 1) // 2 + 1, (sizes[-1][1] - 1) // 2 + 1))
            n_priors = sum(h * w for h, w in sizes) * (head.bbox_layer.out_channels // 4)
            out = head.alloc_outputs(x.shape[0], n_priors, x.device, conf=not scores_only)  # before any side stream starts
            cls = head.alloc_scores(x.shape[0], n_priors, x.device) if scores_only else None
            pyramid = self.fpn(feats, br, chain)
            priors = priors_of(pyramid)
            loc, conf, coef = head.fused_outputs(pyramid, priors.shape[0], br, out, cls)
            proto = prototypes(pyramid[0])
            br.join()
            return _outputs(loc, conf, cls, coef, priors, proto)
        pyramid = self.fpn(feats, None, chain)
        proto = prototypes(pyramid[0])
        priors = priors_of(pyramid)
        if head.fused_applies(pyramid[0]):
            out = head.alloc_outputs(x.shape[0], priors.shape[0], x.device, conf=not scores_only)
            cls = head.alloc_scores(x.shape[0], priors.shape[0], x.device) if scores_only else None
            loc, conf, coef = head.fused_outputs(pyramid, priors.shape[0], None, out, cls)
            return _outputs(loc, conf, cls, coef, priors, proto)
        locs, confs, coefs = zip(*(head(p) for p in pyramid))
        return {"loc": torch.cat(locs, 1), "conf": F.softmax(torch.cat(confs, 1), -1), "mask": torch.cat(coefs, 1),
                "priors": priors, "proto": proto}
