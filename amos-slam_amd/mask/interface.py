"""MaskEngine: the Python side of the reference's `yolact` C++ class (yolact_interface.py:834-884).

`eval_chw(image)` takes what the reference's C++ sends (CHW float32 in [0,1], 3 x 640 x 480) and
returns the 8-bit person mask (480 x 640, values 0 / 255, overlaps wrapping modulo 256);
`eval_bgr(frame)` takes the raw BGR frame and performs the C++ marshalling on the GPU as well.
"""
import os

import torch

from .detect import detect, detect_batch
from .net import YolactR50, _stem_eligible, stem_kernel_enabled
from .post import person_mask, person_mask_batch, person_masks_fused
from .pre import cxx_marshalling, fast_base_transform, resize_f32_cv


class MaskEngine:
    def __init__(self, weight_path=None, device=None, seed=0, conv_dtype=None):
        """conv_dtype: None = float32 everywhere (the default, what every parity statement refers to), or
        torch.float16 / torch.bfloat16 to run the network's convolutions under autocast (fp32 accumulation; the
        detection and mask post-processing stay float32).  The reference runs its cuDNN convolutions in TF32 on the
        GPUs it targets (10-bit mantissa, PyTorch's default), so float16 is the same precision class; it is an
        explicit opt-in because its mask IoU against the reference cannot be pinned without the trained weights."""
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("MaskEngine needs a GPU (PyTorch-ROCm); pass device='cpu' explicitly for CPU tests")
            device = "cuda:0"
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.net = YolactR50()
        self.has_weights = False
        if weight_path:
            self.net.load_weights(weight_path)
            self.has_weights = True
        self.net.eval().to(self.device)
        # NHWC activations/weights: MIOpen's fp32 convolutions run the 550x550 ResNet-50-FPN forward 29 % faster
        # than NCHW on MI355X (22.2 vs 31.2 ms per 16 frames, tools/mask_prof.py); results agree to 3e-7.
        self.conv_dtype = conv_dtype
        self.use_hip_pre = True   # GPU: pre-processing by the HIP kernels; False = the torch restatement (tests)
        self.channels_last = self.device.type == "cuda"
        if self.channels_last:
            self.net.to(memory_format=torch.channels_last)
            torch.backends.cudnn.benchmark = True  # MIOpen solver search on first use of a shape
        self._hip_pre = {}  # (height, width) -> MaskPreprocessor, made on first use (GPU only)

    def prepare(self):
        """Inference form of the network: batch norms folded into the convolutions (same function up to float32
        rounding of the folded weights; tests/test_mask.py holds it to the golden tensors and IoU >= 1 - 1e-3); on the GPU
        also the prediction head's three output convolutions merged into one."""
        self.net.fold_batch_norms()
        if self.device.type == "cuda" and os.environ.get("AMOS_MASK_MERGE_HEADS", "1") != "0":  # (the switch is for A/B runs)
            self.net.merge_head_outputs()  # the prediction head's three output convolutions as one (net.py SharedHead.merge_output_layers)
        if self.channels_last:
            self.net.to(memory_format=torch.channels_last)
        if self.device.type == "cuda" and os.environ.get("AMOS_MASK_WINOGRAD", "1") != "0":
            # the Winograd-transformed weights are shared by every stream that later runs a forward: made (and waited for) here
            from .net import prepare_winograd_weights
            with torch.cuda.device(self.device):
                prepare_winograd_weights(self.net)
        if self.device.type == "cuda":
            from .net import prepare_stem_weight
            with torch.cuda.device(self.device):
                prepare_stem_weight(self.net)  # the stem's 7 x 7 weight in the layout of amos_mask_stem_device
        return self

    def _preprocess_hip(self, frames):
        """The whole pre-processing chain in three HIP kernels (libamos_frontend.so, amos_mask_pre_*): returns
        the [b, 3, 550, 550] network input for [b, H, W, 3] uint8 frames on this engine's GPU."""
        from .. import MaskPreprocessor
        b, h, w = frames.shape[:3]
        pre = self._hip_pre.get((h, w))
        if pre is None or pre.max_batch < b:
            pre = MaskPreprocessor(w, h, max(b, 16), device=self.device.index or 0, stream=None)
            self._hip_pre[(h, w)] = pre
        out = torch.empty((b, 3, 550, 550), dtype=torch.float32, device=self.device)
        frames = frames.contiguous()
        cur = torch.cuda.current_stream(self.device)
        # the kernels run on the preprocessor's own stream: order them after the frames' producer and
        # make the consumer (the network, on torch's stream) wait for them
        ev_in = torch.cuda.Event()
        ev_in.record(cur)
        ext = torch.cuda.ExternalStream(pre.stream_ptr, device=self.device)
        ext.wait_event(ev_in)
        pre.run(frames.data_ptr(), b, out.data_ptr())
        ev_out = torch.cuda.Event()
        ev_out.record(ext)
        cur.wait_event(ev_out)
        return out

    def _masks_of(self, x, width, height):
        """(masks, found) of a network input: the detector's own pass.  With AMOS_MASK_HEAD_SCORES=1, where the fused post-processing will take
        the outputs, the head's kernel writes Detect's class scores directly and no softmax tensor (YolactR50.forward(scores_only=True)): 0.4 GB
        less traffic per 64 frames and one launch less, but the kernel's extra steps cost what they save (0.55 ms against 0.39 + 0.16 at 64
        frames; 107 against 80 + 9 us at one frame), so the default stays the softmax tensor and the separate class-score pass."""
        scores_only = (self.device.type == "cuda" and self.conv_dtype is None and x.dtype == torch.float32
                       and os.environ.get("AMOS_MASK_HEAD_SCORES", "0") == "1" and os.environ.get("AMOS_MASK_FUSED_POST", "1") != "0")
        return self._person_masks(self._forward(x, scores_only), width, height)

    def _forward(self, x, scores_only=False):
        # (the project's stem kernel reads the input through its strides: no layout copy of the network input for it)
        if self.channels_last and not (self.conv_dtype is None and x.dtype == torch.float32 and stem_kernel_enabled() and _stem_eligible(self.net.backbone.conv1)):
            x = x.contiguous(memory_format=torch.channels_last)
        if self.conv_dtype is None:
            return self.net(x, scores_only) if scores_only else self.net(x)
        with torch.autocast(self.device.type, dtype=self.conv_dtype):
            pred = self.net(x)
        return {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in pred.items()}

    @torch.no_grad()
    def network_outputs(self, image_chw):
        """The raw network outputs for a CHW [0,1] float input (for tests and profiling)."""
        x = torch.as_tensor(image_chw, dtype=torch.float32, device=self.device)
        img = x.permute(1, 2, 0) * 255            # image.transpose((1, 2, 0)) * 255
        img = resize_f32_cv(img, 640, 480)        # cv2.resize(image, (640, 480))
        return self._forward(fast_base_transform(img)), img

    @torch.no_grad()
    def eval_chw(self, image_chw):
        pred, img = self.network_outputs(image_chw)
        h, w = img.shape[:2]
        mask = person_mask(detect(pred), w, h)
        if mask is None:  # the reference raises here; its caller keeps the pre-zeroed mask (Tracking.cc:305)
            return None
        return mask

    @torch.no_grad()
    def eval_bgr(self, bgr_u8):
        frame = torch.as_tensor(bgr_u8, dtype=torch.uint8, device=self.device)
        return self.eval_chw(cxx_marshalling(frame))

    @staticmethod
    def _person_masks(pred, width, height):
        """(masks uint8 [B, height, width], found bool [B]) of a forward's outputs: the fused library call on the GPU, the torch ops elsewhere"""
        fused = person_masks_fused(pred, width, height)
        if fused is None and "conf" not in pred:
            raise RuntimeError("the pass was run for the fused post-processing (class scores only), which does not apply to its outputs")
        return fused if fused is not None else person_mask_batch(detect_batch(pred), width, height)

    @torch.no_grad()
    def eval_net_input_batch(self, x, chunk=16, height=480, width=640):
        """x: [B, 3, 550, 550] float32 network input already on the engine's device (amos_orb_detect_color_with_mask_pre_batch_device
        or amos_mask_preprocess_batch_device wrote it).  Returns [B, height, width] uint8 masks like eval_bgr_batch."""
        B = x.shape[0]
        out = torch.zeros((B, height, width), dtype=torch.uint8, device=self.device)
        for b0 in range(0, B, chunk):
            masks, _found = self._masks_of(x[b0:b0 + chunk], width, height)
            out[b0:b0 + masks.shape[0]] = masks
        return out

    # ---- one HIP graph for network + detection + mask assembly at a fixed batch: the frame-by-frame caller (Tracking.cc:366 calls
    # evalImage once per frame) is bound by the ~250 launches of a pass, not by their work (3.5 ms per frame eagerly, tools/mask_latency.py)
    @torch.no_grad()
    def capture_graph(self, batch=1):
        """Capture `network input [batch, 3, 550, 550] -> masks [batch, 480, 640], found [batch]` (the static-shape batch path:
        detect_batch + person_mask_batch, every kernel on the capturing stream) into a HIP graph and keep it for eval_net_input_graph.
        MIOpen picks its solvers in the eager warm-up passes before the capture.  Returns self."""
        if self.device.type != "cuda":
            raise RuntimeError("capture_graph needs the GPU")
        self._g_in = torch.zeros((batch, 3, 550, 550), dtype=torch.float32, device=self.device)

        def body():
            return self._masks_of(self._g_in, 640, 480)

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._g_masks, self._g_found = body()
        self._graph, self._g_batch = graph, batch
        return self

    @torch.no_grad()
    def eval_net_input_graph(self, x):
        """x: [batch, 3, 550, 550] float32 on the engine's device, batch as captured.  Replays the graph; returns (masks uint8
        [batch, 480, 640], found bool [batch]) -- the same tensors every call (copy them to keep them)."""
        if getattr(self, "_graph", None) is None or x.shape[0] != self._g_batch:
            raise RuntimeError("eval_net_input_graph: call capture_graph(batch=%d) first" % x.shape[0])
        self._g_in.copy_(x)
        self._graph.replay()
        return self._g_masks, self._g_found

    @torch.no_grad()
    def eval_bgr_graph(self, frames_u8):
        """[batch, H, W, 3] uint8 BGR frames on the engine's device -> (masks, found) through the HIP pre-processing kernels and the
        captured graph (the pre-processing runs eagerly: three kernels on its own stream)."""
        frames = torch.as_tensor(frames_u8, dtype=torch.uint8, device=self.device)
        return self.eval_net_input_graph(self._preprocess_hip(frames))

    @torch.no_grad()
    def frame_session(self, height, width):
        """The per-frame path of `yolact::evalImage` (yolact.cc:203-318 hands over ONE frame per call, Tracking.cc:366) as a FrameSession:
        pinned host buffers the C++ class writes the frame into / reads the mask from, and ONE HIP graph holding the three pre-processing
        kernels, the network, the detection and the mask assembly.  Kept per frame size."""
        key = (int(height), int(width))
        sessions = self.__dict__.setdefault("_sessions", {})
        if key not in sessions:
            sessions[key] = FrameSession(self, *key)
        return sessions[key]

    @torch.no_grad()
    def eval_bgr_batch(self, frames_u8, chunk=16):
        """frames_u8: [B, H, W, 3] uint8 on the engine's device.  Returns [B, H, W] uint8 masks (zeros where
        the network finds nothing, which is what the reference's caller ends up using, Tracking.cc:305).
        The network runs on `chunk` frames at a time; pre- and post-processing are per frame."""
        frames = torch.as_tensor(frames_u8, dtype=torch.uint8, device=self.device)
        B, H, W = frames.shape[:3]
        out = torch.zeros((B, H, W), dtype=torch.uint8, device=self.device)
        for b0 in range(0, B, chunk):
            part = frames[b0:b0 + chunk]
            if self.device.type == "cuda" and self.use_hip_pre:
                x = self._preprocess_hip(part)                                  # [b, 3, 550, 550]
            else:
                chw = cxx_marshalling(part)                                      # [b, 3, 640, 480]
                imgs = resize_f32_cv(chw.permute(0, 2, 3, 1) * 255, 640, 480)    # [b, 480, 640, 3]
                x = fast_base_transform(imgs)
            masks, _found = self._masks_of(x, 640, 480)                         # eval_image resizes to 640 x 480 whatever came in
            out[b0:b0 + part.shape[0]] = masks
        return out


class FrameSession:
    """One frame at a time through a captured HIP graph, host buffers pinned (see MaskEngine.frame_session).

    frame_in  : pinned uint8 [height, width, 3] -- the caller writes the BGR frame here (address: in_ptr)
    mask_out  : pinned uint8 [480, 640]          -- the person mask of the last run() (address: out_ptr)
    run()     : H2D copy, graph replay, D2H copies, one stream synchronisation; returns whether a detection passed the score threshold
                (the reference raises IndexError otherwise and its caller keeps the pre-zeroed mask, Tracking.cc:305)."""

    def __init__(self, engine, height, width):
        if engine.device.type != "cuda":
            raise RuntimeError("FrameSession needs the GPU")
        from .. import MaskPreprocessor
        dev = engine.device
        self.engine, self.height, self.width = engine, height, width
        self.stream = torch.cuda.Stream(device=dev)
        self.frame_in = torch.zeros((height, width, 3), dtype=torch.uint8).pin_memory()
        self.mask_out = torch.zeros((480, 640), dtype=torch.uint8).pin_memory()
        self._found_host = torch.zeros(1, dtype=torch.bool).pin_memory()
        self._d_frame = torch.zeros((1, height, width, 3), dtype=torch.uint8, device=dev)
        self._net_in = torch.zeros((1, 3, 550, 550), dtype=torch.float32, device=dev)
        # the pre-processing kernels issue on THIS session's stream, so the capture below records them as nodes of the graph
        self._pre = MaskPreprocessor(width, height, 1, device=dev.index or 0, stream=self.stream.cuda_stream)

        def body():
            self._pre.run(self._d_frame.data_ptr(), 1, self._net_in.data_ptr())
            return engine._masks_of(self._net_in, 640, 480)

        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(3):  # MIOpen picks its solvers, the allocator warms up
                body()
        self.stream.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=self.stream):
            self._masks, self._found = body()
        self.stream.synchronize()

    @property
    def in_ptr(self):
        return self.frame_in.data_ptr()

    @property
    def out_ptr(self):
        return self.mask_out.data_ptr()

    @torch.no_grad()
    def run(self):
        with torch.cuda.stream(self.stream):
            self._d_frame[0].copy_(self.frame_in, non_blocking=True)
            self._graph.replay()
            self.mask_out.copy_(self._masks[0], non_blocking=True)
            self._found_host.copy_(self._found, non_blocking=True)
        self.stream.synchronize()
        return bool(self._found_host[0])
