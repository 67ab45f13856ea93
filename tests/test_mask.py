"""The mask pass (YOLACT ResNet-50-FPN on PyTorch) against golden tensors captured from the
reference's own Python network code (tools/gen_yolact_golden.py, run once in the build container).
Tolerances: float32 network => rtol 2e-4 on tensors (CPU, same ops as the reference) / 2e-3 (GPU
kernels differ); final person mask IoU >= 1 - 1e-3, the bar BASELINE.json's north_star states."""
import importlib
import os

import numpy as np
import pytest
import torch

import mask_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = {c: np.load(os.path.join(ROOT, "tests", "golden", f"yolact_{c}.npz")) for c in mask_cases.CASES}
CASES = list(mask_cases.CASES)   # six (frame, weight seed) cases: a checkerboard, the reference's two sample inputs, a synthetic scene; seeds 0 / 1 / 3
G = GOLD["seed0"]
SUB = int(G["sub"][0])


@pytest.fixture(scope="module")
def mask(pkg):
    return importlib.import_module("amos_slam_amd.mask")


def _frame(case="seed0"):
    import zlib
    f = mask_cases.frame(case)
    assert zlib.crc32(f.tobytes()) == int(GOLD[case]["frame_crc"][0]), "the case's frame is not the one the fixture was made from"
    return f


def _engine(mask, device, case="seed0"):
    eng = mask.MaskEngine(device=device, seed=mask_cases.weight_seed(case))
    mask_cases.bias_class_head(eng.net, case)   # same bias tweak as the fixture generator
    return eng


def _spy_winograd(gpu_lib, monkeypatch, calls, lo, hi):
    """Records arguments [lo:hi] of every Winograd convolution launch of either family (F(2 x 2): mask_winograd_conv, F(2 x 4): mask_winograd24_conv)."""
    for name in ("mask_winograd_conv", "mask_winograd24_conv", "mask_winograd24_conv_layout"):  # (the layout form: the same argument positions)
        real = getattr(gpu_lib, name)
        monkeypatch.setattr(gpu_lib, name, lambda *a, _real=real: (calls.append(a[lo:hi]), _real(*a))[1])


def _iou(got, want):
    """Intersection over union of two boolean masks; two empty masks agree (the cars-only case)."""
    union = int((got | want).sum())
    return 1.0 if union == 0 else int((got & want).sum()) / union


def _sub(t):
    return t.detach().float().cpu().reshape(-1)[::SUB].numpy()


def _run(mask, device, rtol, atol, fold=False, case="seed0"):
    G = GOLD[case]
    eng = _engine(mask, device, case)
    if fold:
        eng.prepare()
    frame = torch.from_numpy(_frame(case)).to(device)
    chw = mask.cxx_marshalling(frame)
    img = mask.resize_f32_cv(chw.permute(1, 2, 0) * 255, 640, 480)
    batch = mask.fast_base_transform(img)
    np.testing.assert_allclose(_sub(batch), G["batch"], rtol=1e-5, atol=1e-4)
    with torch.no_grad():
        feats = eng.net.backbone(batch)
        pyramid = eng.net.fpn(feats[1:])
        pred = eng.net(batch)
    np.testing.assert_allclose(_sub(feats[1]), G["c3"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(_sub(feats[3]), G["c5"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(_sub(pyramid[0]), G["p3"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(_sub(pyramid[4]), G["p7"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(_sub(pred["proto"][0]), G["proto"], rtol=rtol, atol=atol)
    det = mask.detect(pred)
    assert det is not None and len(det["score"]) == len(G["det_score"])
    np.testing.assert_allclose(det["score"].cpu().numpy(), G["det_score"], rtol=rtol, atol=atol)
    same_order = np.array_equal(det["class"].cpu().numpy(), G["det_class"])
    if rtol <= 2e-4:  # CPU: identical arithmetic => identical ranking
        assert same_order
        np.testing.assert_allclose(det["box"].cpu().numpy(), G["det_box"], rtol=rtol, atol=atol)
        np.testing.assert_allclose(det["mask"].cpu().numpy(), G["det_mask"], rtol=rtol, atol=atol)
    out = mask.postprocess_masks(det, 640, 480)
    assert out is not None
    classes, scores, masks = out
    assert len(scores) == len(G["post_scores"])
    if same_order and rtol <= 2e-4:  # GPU kernels may swap near-tied detections; the mask IoU below is the bar
        area = masks.sum((1, 2)).cpu().numpy()
        assert np.abs(area - G["post_mask_area"]).max() <= max(2.0, 2e-3 * G["post_mask_area"].max())
    person = mask.person_mask(det, 640, 480).cpu().numpy()
    want = np.unpackbits(G["person_mask_bits"])[:480 * 640].reshape(480, 640).astype(bool)
    got = person > 0
    iou = _iou(got, want)
    assert iou >= 1 - 1e-3, iou
    assert set(np.unique(person)) <= set(G["person_mask_values"].tolist()) | {0}
    return iou


def test_state_dict_keys_match_reference(mask):
    """Every key and shape of the reference's Yolact().state_dict() exists here: its .pth loads strictly."""
    net = mask.YolactR50()
    assert sorted(net.state_dict().keys()) == [str(k) for k in G["keys"]]
    assert sum(p.numel() for p in net.parameters()) == 31164943


def test_priors(mask):
    pri = mask.build_priors([(69, 69), (35, 35), (18, 18), (9, 9), (5, 5)])
    assert pri.shape == (19248, 4)
    assert torch.allclose(pri[0], torch.tensor([0.5 / 69, 0.5 / 69, 24 / 550, 24 / 550]))
    assert torch.allclose(pri[1, 2], torch.tensor(24 * (0.5 ** 0.5) / 550)) and pri[1, 2] == pri[1, 3]  # square anchors


def test_u8_resize_matches_oracle(mask, ob):
    """The torch restatement of cv::resize(8-bit) equals the C oracle's, channel by channel."""
    frame = _frame()
    got = mask.resize_u8_cv(torch.from_numpy(frame), 480, 640).numpy()
    for c in range(3):
        assert np.array_equal(got[:, :, c], ob.resize_linear_u8(np.ascontiguousarray(frame[:, :, c]), 480, 640))
    chw = mask.cxx_marshalling(torch.from_numpy(frame))
    assert chw.shape == (3, 640, 480) and chw.dtype == torch.float32
    assert np.array_equal(chw.numpy()[1], (got[:, :, 1].astype(np.float64) / 255.0).astype(np.float32))


def test_fast_nms_tie_and_threshold_rules(mask):
    boxes = torch.tensor([[0.1, 0.1, 0.5, 0.5], [0.12, 0.1, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9], [0.1, 0.1, 0.5, 0.5]])
    coefs = torch.arange(4, dtype=torch.float32).view(4, 1).repeat(1, 32)
    scores = torch.tensor([[0.9, 0.8, 0.7, 0.1], [0.2, 0.3, 0.1, 0.95]])
    b, m, c, s = mask.fast_nms(boxes, coefs, scores)
    # class 0: box 1 suppressed by box 0, box 3 (same as 0) suppressed; class 1: box 3 best, 0 and 1 suppressed by it
    assert c.tolist() == [1, 0, 0, 1] and [round(v, 2) for v in s.tolist()] == [0.95, 0.9, 0.7, 0.1]


@pytest.mark.parametrize("case", CASES)
def test_network_vs_reference_cpu(mask, case):
    assert _run(mask, "cpu", 2e-4, 2e-5, case=case) >= 1 - 1e-3


@pytest.mark.parametrize("case", CASES)
def test_network_with_folded_batch_norms_vs_reference_cpu(mask, case):
    """MaskEngine.prepare(): the inference form bench.py and the C++ class run."""
    assert _run(mask, "cpu", 2e-4, 2e-5, fold=True, case=case) >= 1 - 1e-3


def test_folding_batch_norms_with_nontrivial_statistics(mask):
    """Freshly initialised batch norms are almost the identity; give them statistics a trained checkpoint has and
    compare the folded network with the unfolded one on the same input."""
    eng = _engine(mask, "cpu")
    torch.manual_seed(3)
    n_bn = 0
    for m in eng.net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            n_bn += 1
    assert n_bn == 53
    x = torch.randn(1, 3, 550, 550)
    with torch.no_grad():
        a = eng.net(x)
        eng.prepare()
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in eng.net.modules())
        b = eng.net(x)
        eng.prepare()  # idempotent
        c = eng.net(x)
    for k in ("loc", "conf", "mask", "proto"):
        scale = float(a[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 2e-5 * max(scale, 1e-3), k
        assert torch.equal(b[k], c[k])


def test_batch_path_equals_single(mask):
    eng = _engine(mask, "cpu")
    f0 = _frame()
    f1 = np.ascontiguousarray(f0[:, ::-1])
    batch = eng.eval_bgr_batch(torch.from_numpy(np.stack([f0, f1])), chunk=2)
    assert torch.equal(batch[0], eng.eval_bgr(f0)) and torch.equal(batch[1], eng.eval_bgr(f1))


def test_static_shape_batch_post_equals_reference_order_post(mask):
    """detect_batch / person_mask_batch (static shapes, one launch set per batch) give the masks of the
    per-frame detect / person_mask, which follow the reference's code path operation by operation."""
    eng = _engine(mask, "cpu")
    f0 = _frame()
    frames = torch.from_numpy(np.stack([f0, np.ascontiguousarray(f0[::-1]), np.ascontiguousarray(f0[:, ::-1])]))
    chw = mask.cxx_marshalling(frames)
    imgs = mask.resize_f32_cv(chw.permute(0, 2, 3, 1) * 255, 640, 480)
    with torch.no_grad():
        pred = eng.net(mask.fast_base_transform(imgs))
    got, found = mask.person_mask_batch(mask.detect_batch(pred), 640, 480)
    assert bool(found.all())
    for k in range(3):
        want = mask.person_mask(mask.detect(pred, k), 640, 480)
        g, w_ = got[k].numpy() > 0, want.numpy() > 0
        assert (g & w_).sum() / max((g | w_).sum(), 1) >= 1 - 1e-3
        assert torch.equal(got[k], want)
    # nothing detected: zeros and found == False
    none = mask.MaskEngine(device="cpu", seed=1)
    with torch.no_grad():
        pred0 = none.net(mask.fast_base_transform(imgs[:1]))
    m0, f0_ = mask.person_mask_batch(mask.detect_batch(pred0), 640, 480)
    assert not bool(f0_[0]) and int(m0.sum()) == 0


def test_small_pass_and_layout_rules_on_the_host(mask, pkg, monkeypatch):
    """Host logic of round 5's dispatch helpers (mask/net.py): no side streams, no channel-blocked chain and no one-kernel stem for tensors that are
    not float32 on a GPU; the environment switches; the engine's CPU passes never ask for class scores only."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    x = torch.zeros(1, 3, 550, 550)
    for mode in ("auto", "1", "0"):
        monkeypatch.setenv("AMOS_MASK_BRANCHES", mode)
        assert net_mod.branches_for(x) is None   # a CPU tensor: never
    monkeypatch.delenv("AMOS_MASK_BRANCHES")
    head = net_mod.SharedHead()
    p3 = torch.zeros(8, 256, 69, 69).contiguous(memory_format=torch.channels_last)
    convs = [torch.nn.Conv2d(256, 256, 3, padding=1) for _ in range(2)]
    assert net_mod.blocked_chain_for(p3, convs, convs[:1]) is False
    monkeypatch.setenv("AMOS_MASK_BLOCKED_CHAIN", "0")
    assert net_mod.blocked_chain_for(p3, convs, convs[:1]) is False
    assert head.fused_applies(p3) is False and not net_mod._stem_eligible(torch.nn.Conv2d(3, 64, 7, stride=2, padding=3))
    monkeypatch.setenv("AMOS_MASK_STEM", "library")
    assert not net_mod.stem_kernel_enabled()
    monkeypatch.delenv("AMOS_MASK_STEM")
    assert net_mod.stem_kernel_enabled()
    b = net_mod.Blocked(torch.zeros(2, 4, 5, 6, 8), (2, 32, 5, 6))
    assert b.shape == (2, 32, 5, 6) and b.dtype == torch.float32 and b.device.type == "cpu"
    eng = mask.MaskEngine(device="cpu", seed=1)
    seen = []
    real = eng.net.forward
    monkeypatch.setattr(eng.net, "forward", lambda x, scores_only=False: (seen.append(scores_only), real(x, scores_only))[1])
    monkeypatch.setenv("AMOS_MASK_HEAD_SCORES", "1")
    with torch.no_grad():
        eng._masks_of(torch.zeros(1, 3, 550, 550), 640, 480)
    assert seen == [False]


def test_winograd_rule(mask, pkg, monkeypatch):
    """Which convolutions go to the Winograd kernel (mask/net.py winograd_rule; bench.py counts the executed FLOPs with the same function):
    3 x 3 / stride 1 / pad 1, cin % 16 == 0 from 32 up, cout % 64 == 0, at least 256 work-groups of 64 tiles x 64 channels.  Host logic only."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    rule = lambda cin, cout, b, hw, k=(3, 3), s=(1, 1), p=(1, 1): net_mod.winograd_rule(cin, cout, k, s, p, (1, 1), 1, b, hw, hw)
    monkeypatch.delenv("AMOS_MASK_WINOGRAD", raising=False)
    assert rule(256, 256, 32, 138) and rule(256, 384, 32, 69) and rule(64, 64, 32, 138) and rule(512, 512, 32, 18) and rule(256, 256, 1, 138)
    # one frame per pass: the layers whose stage count x stage latency beats the library's direct kernel + bias pass (32 output channels per
    # work-group where 64-channel groups would be few) ...
    assert rule(256, 256, 1, 69) and rule(256, 384, 1, 69) and rule(128, 128, 1, 69) and rule(64, 64, 1, 138)
    # ... and the ones that stay with the library: 32 or 64 stages of latency for a few GFLOP
    assert not rule(256, 256, 1, 35) and not rule(256, 384, 1, 35) and not rule(512, 512, 1, 18) and not rule(256, 256, 1, 18) and not rule(256, 256, 1, 9)
    assert not rule(256, 256, 32, 138, s=(2, 2)) and not rule(256, 256, 32, 138, k=(1, 1), p=(0, 0)) and not rule(256, 256, 32, 138, p=(0, 0))
    assert not rule(256, 352, 32, 69) and not rule(24, 64, 32, 138) and not rule(16, 64, 32, 138)   # channel counts the kernel does not take
    assert not rule(2048, 64, 64, 138)                  # input of 2 GiB and more: the buffer descriptor's range
    monkeypatch.setenv("AMOS_MASK_WINOGRAD", "2")
    assert rule(256, 256, 1, 35) and not rule(256, 352, 1, 69)
    monkeypatch.setenv("AMOS_MASK_WINOGRAD", "0")
    assert not rule(256, 256, 32, 138)
    assert pkg.mask_winograd_supported(32, 64) and not pkg.mask_winograd_supported(8, 64)
    # the merged prediction-head output layer is padded to a channel count the kernels take
    head = net_mod.SharedHead()
    head.merge_output_layers()
    assert head.merged.out_channels == 384 and float(head.merged.weight[351:].abs().sum()) == 0.0 and float(head.merged.bias[351:].abs().sum()) == 0.0
    assert torch.equal(head.merged.weight[:12], head.bbox_layer.weight) and torch.equal(head.merged.weight[12:255], head.conf_layer.weight)


def test_no_detection_returns_none(mask):
    eng = mask.MaskEngine(device="cpu", seed=1)  # unbiased random weights: softmax ~ 1/81 < 0.05
    assert eng.eval_bgr(_frame()) is None


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_network_vs_reference_gpu(mask, gpu_lib, case):
    assert _run(mask, "cuda:0", 5e-3, 5e-3, case=case) >= 1 - 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_network_with_folded_batch_norms_vs_reference_gpu(mask, gpu_lib, case):
    assert _run(mask, "cuda:0", 5e-3, 5e-3, fold=True, case=case) >= 1 - 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_network_on_the_project_gemm_vs_reference_gpu(mask, gpu_lib, monkeypatch, case):
    """The golden tensors again with EVERY convolution the project's MFMA GEMM can take forced onto it (at one frame the automatic
    rule leaves them to MIOpen): same tolerances, same person-mask IoU bar."""
    monkeypatch.setenv("AMOS_MASK_CONV1X1", "1")
    monkeypatch.setenv("AMOS_MASK_CONV3X3", "2")
    calls = []
    real = gpu_lib.mask_conv

    def counting(*a):
        calls.append(a[9:15])  # cin, cout, kh, kw, stride, pad
        return real(*a)

    monkeypatch.setattr(gpu_lib, "mask_conv", counting)
    assert _run(mask, "cuda", 5e-3, 5e-3, fold=True, case=case) >= 1 - 1e-3
    kinds = {(c[2], c[4]) for c in calls}
    assert len(calls) >= 60 and {(1, 1), (1, 2), (3, 1), (3, 2)} <= kinds, (len(calls), kinds)  # 1x1 and 3x3, strides 1 and 2


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["22", "24"])
@pytest.mark.parametrize("case", CASES)
def test_network_with_winograd_everywhere_vs_reference_gpu(mask, gpu_lib, monkeypatch, case, family):
    """The golden tensors and the person-mask IoU with EVERY stride-1 3 x 3 convolution of the network forced onto the Winograd kernel of
    either family, F(2 x 2) and F(2 x 4) (the rule leaves the small launches of one frame to the other paths): same tolerances, same IoU bar."""
    monkeypatch.setenv("AMOS_MASK_WINOGRAD", "2")
    monkeypatch.setenv("AMOS_MASK_WINOGRAD_F", family)
    calls = []
    _spy_winograd(gpu_lib, monkeypatch, calls, 7, 11)
    assert _run(mask, "cuda", 5e-3, 5e-3, fold=True, case=case) >= 1 - 1e-3
    # one forward: 13 bottleneck conv2 with stride 1, 3 FPN prediction layers, 4 protonet layers, 5 levels x (upfeature + merged head
    # output); _run also calls the backbone and the FPN once more on their own
    assert len(calls) == (13 + 3 + 4 + 10) + 13 + 3, len(calls)
    assert {c[2:] for c in calls} >= {(64, 64), (128, 128), (256, 256), (512, 512), (256, 384)}


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_engine_end_to_end_gpu(mask, gpu_lib, case):
    eng = _engine(mask, "cuda:0", case)
    m = eng.eval_bgr(_frame(case))
    assert m.shape == (480, 640) and m.dtype == torch.uint8
    want = np.unpackbits(GOLD[case]["person_mask_bits"])[:480 * 640].reshape(480, 640).astype(bool)
    got = m.cpu().numpy() > 0
    assert _iou(got, want) >= 1 - 1e-3


@pytest.mark.gpu
def test_hip_preprocessing_matches_the_torch_chain(mask, gpu_lib):
    """8f-4: amos_mask_preprocess_batch_device (three HIP kernels) against the torch restatement of the same chain
    (cxx_marshalling -> * 255 -> resize_f32_cv -> fast_base_transform).  The two OpenCV resizes and the u8 -> float
    step are exact by construction; torch's bilinear kernel may fuse multiply-adds, hence the small tolerance
    (values are normalised pixels in about [-2.2, 2.7])."""
    rng = np.random.default_rng(4)
    frames = torch.from_numpy(rng.integers(0, 256, (5, 480, 640, 3), dtype=np.uint8)).cuda()
    frames[0] = torch.from_numpy(_frame()).cuda()
    eng = _engine(mask, "cuda:0")
    got = eng._preprocess_hip(frames)
    chw = mask.cxx_marshalling(frames)
    imgs = mask.resize_f32_cv(chw.permute(0, 2, 3, 1) * 255, 640, 480)
    want = mask.fast_base_transform(imgs)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (5, 3, 550, 550)
    err = (got - want).abs().max().item()
    assert err < 1e-4, err  # measured 3.5e-5 on random-noise frames (one ulp of a bilinear weight times 255 / 57)
    # end to end: the engine's masks do not depend on which implementation prepared the input
    m_hip = eng.eval_bgr_batch(frames)
    eng.use_hip_pre = False
    m_torch = eng.eval_bgr_batch(frames)
    a, b = m_hip > 0, m_torch > 0
    assert ((a & b).sum().item() / max((a | b).sum().item(), 1)) >= 1 - 1e-3


def _yolact_class_roundtrip(mask, tmp_path, device):
    """The C++ ORB_SLAM2::yolact class (embedded CPython) end to end: ctor(py file, weights, categories),
    evalImage(BGR frame) -> 8-bit mask; weights travel through a .pth exactly like the reference's."""
    import host_binding as hb
    os.environ["AMOS_MASK_DEVICE"] = device
    eng = _engine(mask, "cpu")
    pth = str(tmp_path / "yolact_test_weights.pth")
    torch.save(eng.net.state_dict(), pth)
    py_file = os.path.join(ROOT, "amos-slam_amd", "mask", "yolact_interface.py")
    got = hb.host_yolact_eval(py_file, pth, _frame()) > 0
    want = np.unpackbits(G["person_mask_bits"])[:480 * 640].reshape(480, 640).astype(bool)
    assert (got & want).sum() / max((got | want).sum(), 1) >= 1 - 1e-3
    # nothing detected (unbiased random weights) => evalImage returns false with a description, mask untouched
    torch.manual_seed(5)
    torch.save(mask.YolactR50().state_dict(), pth + ".none")
    with pytest.raises(RuntimeError, match="yolact_eval"):
        hb.host_yolact_eval(py_file, pth + ".none", _frame())
    # constructor failures leave isInitializedResult() false
    with pytest.raises(RuntimeError, match="import py moudle"):
        hb.host_yolact_eval(os.path.join(ROOT, "amos-slam_amd", "mask", "no_such_module.py"), pth, _frame())


def test_cxx_yolact_class_cpu(mask, tmp_path):
    if not os.path.exists(os.path.join(ROOT, "amos-slam_amd", "host", "libamos_host.so")):
        pytest.skip("host library not built")
    _yolact_class_roundtrip(mask, tmp_path, "cpu")


@pytest.mark.gpu
def test_cxx_yolact_class_gpu(mask, gpu_lib, tmp_path):
    _yolact_class_roundtrip(mask, tmp_path, "cuda:0")


@pytest.mark.gpu
def test_fused_conv_epilogue_equals_the_torch_ops(mask, gpu_lib):
    """amos_mask_bias_act_device (bias + residual + ReLU in one in-place pass behind the convolution) against PyTorch's
    separate passes: same summation order, so the same bits -- vector path (channels % 4 == 0), scalar path (243 channels),
    with and without residual / ReLU."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    torch.manual_seed(5)
    for cin, cout, k, stride in ((64, 256, 1, 1), (128, 128, 3, 2), (256, 243, 3, 1), (3, 64, 7, 2)):
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=True).cuda().to(memory_format=torch.channels_last)
        x = torch.randn(3, cin, 37, 41, device="cuda").contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            plain = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding)
            res = torch.randn_like(plain)
            for relu in (True, False):
                for r in (None, res):
                    want = plain + conv.bias.view(1, -1, 1, 1)
                    if r is not None:
                        want = want + r
                    if relu:
                        want = torch.relu(want)
                    got = plain.clone(memory_format=torch.channels_last)  # the epilogue alone, on the same convolution output
                    assert got.is_contiguous(memory_format=torch.channels_last)
                    gpu_lib.mask_bias_act(torch.cuda.current_stream().cuda_stream, got.data_ptr(), conv.bias.data_ptr(),
                                          r.data_ptr() if r is not None else None, got.numel(), cout, relu)
                    torch.cuda.synchronize()
                    assert torch.equal(got, want), (cin, cout, k, relu, r is not None)
                    fused = net_mod.conv_bias_act(conv, x, relu, residual=r)  # with its own convolution call (MIOpen may pick another solver)
                    assert fused.shape == want.shape and torch.allclose(fused, want, rtol=1e-4, atol=1e-4)
    # the engine end to end is covered by test_network_with_folded_batch_norms_vs_reference_gpu (golden tensors, IoU)


@pytest.mark.gpu
def test_mfma_conv1x1_against_float64(mask, gpu_lib, monkeypatch):
    """amos_mask_conv1x1_device (fp32 MFMA GEMM, bias + residual + ReLU in the epilogue) against a float64 convolution: both
    work-group shapes (output channels % 128 == 0 and 64 / 192), strides 1 and 2, a row count that is not a multiple of the tile,
    with and without bias / residual / ReLU.  Tolerance: float32 rounding of a K-term sum (1e-5 relative to the sum of |terms|),
    and no worse than the library convolution's own error.  Then the same through conv_bias_act's dispatch, both forced sides."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    F = torch.nn.functional
    torch.manual_seed(7)
    cl = torch.channels_last
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_lib.mask_conv1x1_supported(64, 256, 1) and gpu_lib.mask_conv1x1_supported(2048, 64, 2)
    assert not gpu_lib.mask_conv1x1_supported(3, 64, 1) and not gpu_lib.mask_conv1x1_supported(256, 32, 1) and not gpu_lib.mask_conv1x1_supported(80, 64, 1)
    for b, cin, cout, stride, h, w in ((3, 64, 256, 1, 37, 41), (2, 256, 64, 1, 23, 19), (2, 512, 1024, 2, 35, 33), (1, 2048, 512, 1, 18, 18),
                                       (2, 128, 192, 1, 9, 7), (1, 32, 128, 3, 10, 11)):
        x = torch.randn(b, cin, h, w, device="cuda").contiguous(memory_format=cl)
        wgt = (torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5)
        bias = torch.randn(cout, device="cuda")
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        res = torch.randn(b, cout, oh, ow, device="cuda").contiguous(memory_format=cl)
        exact = F.conv2d(x.double(), wgt.double(), None, stride)
        bound = 1e-5 * F.conv2d(x.double().abs(), wgt.double().abs(), None, stride) + 1e-6
        for wfmt in (torch.contiguous_format, cl):
            wt = wgt.contiguous(memory_format=wfmt)
            for use_bias, use_res, relu in ((True, True, True), (True, False, True), (False, True, False), (False, False, False), (True, False, False)):
                # both work-group shapes: the automatic rule gives these small launches 128 x 64 tiles, the batched path's 128 x 128
                # kernels (k_conv_gemm<2, 2, 2, *>) run when forced (cout % 128 == 0 only; elsewhere mode 0 changes nothing)
                for tile_mode in (0, 1):
                    gpu_lib.mask_conv_tile_mode(tile_mode)
                    assert ("<2, 2, 2, false>" in gpu_lib.mask_conv_kernel_name(b, h, w, cin, cout, 1, 1, stride, 0)) == (tile_mode == 0 and cout % 128 == 0)
                    y = torch.full((b, cout, oh, ow), float("nan"), device="cuda").contiguous(memory_format=cl)
                    gpu_lib.mask_conv1x1(st, x.data_ptr(), wt.data_ptr(), bias.data_ptr() if use_bias else None, res.data_ptr() if use_res else None,
                                         y.data_ptr(), b, h, w, cin, cout, stride, relu)
                    torch.cuda.synchronize()
                    gpu_lib.mask_conv_tile_mode(-1)
                    want = exact + (bias.double().view(1, -1, 1, 1) if use_bias else 0) + (res.double() if use_res else 0)
                    if relu:
                        want = want.relu()
                    err = (y.double() - want).abs()
                    assert torch.isfinite(y).all() and bool((err <= bound).all()), (cin, cout, stride, use_bias, use_res, relu, tile_mode, err.max().item())
        conv = torch.nn.Conv2d(cin, cout, 1, stride=stride, bias=True).cuda().to(memory_format=cl)
        with torch.no_grad():
            conv.weight.copy_(wgt)
            conv.bias.copy_(bias)
            outs = {}
            for side in ("0", "1"):
                monkeypatch.setenv("AMOS_MASK_CONV1X1", side)
                outs[side] = net_mod.conv_bias_act(conv, x, True, residual=res)
            want = (exact + bias.double().view(1, -1, 1, 1) + res.double()).relu()
            for side, y in outs.items():
                assert y.is_contiguous(memory_format=cl) and bool(((y.double() - want).abs() <= bound).all()), (side, cin, cout)
    # the general entry point: k x k taps as an implicit GEMM (weight [cout][kh][kw][cin] = a channels-last Conv2d weight)
    assert gpu_lib.mask_conv_supported(256, 384, 3, 3, 1, 1) and not gpu_lib.mask_conv_supported(256, 243, 3, 3, 1, 1) and not gpu_lib.mask_conv_supported(64, 64, 3, 3, 1, 3)
    for b, cin, cout, k, stride, pad, h, w in ((2, 64, 64, 3, 1, 1, 21, 17), (1, 128, 128, 3, 2, 1, 30, 31), (2, 256, 384, 3, 1, 1, 9, 9), (1, 32, 64, 3, 1, 0, 12, 7),
                                               (1, 64, 128, 5, 2, 2, 13, 16), (2, 32, 64, 1, 2, 0, 7, 7)):
        x = torch.randn(b, cin, h, w, device="cuda").contiguous(memory_format=cl)
        wgt = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).contiguous(memory_format=cl)
        bias = torch.randn(cout, device="cuda")
        exact = F.conv2d(x.double(), wgt.double(), bias.double(), stride, pad)
        bound = 1e-5 * F.conv2d(x.double().abs(), wgt.double().abs(), None, stride, pad) + 1e-6
        res = torch.randn(exact.shape, device="cuda").contiguous(memory_format=cl)
        for use_res, relu in ((False, True), (True, False)):
            for tile_mode in (0, 1):  # 128 x 128 (k_conv_gemm<2, 2, 2, true>, the batched path's kernel) and 128 x 64 work-groups
                gpu_lib.mask_conv_tile_mode(tile_mode)
                assert ("<2, 2, 2, true>" in gpu_lib.mask_conv_kernel_name(b, h, w, cin, cout, k, k, stride, pad)) == (tile_mode == 0 and cout % 128 == 0 and (k > 1 or pad > 0))
                y = torch.full(exact.shape, float("nan"), device="cuda").contiguous(memory_format=cl)
                gpu_lib.mask_conv(st, x.data_ptr(), wgt.data_ptr(), bias.data_ptr(), res.data_ptr() if use_res else None, y.data_ptr(), b, h, w, cin, cout, k, k,
                                  stride, pad, relu)
                torch.cuda.synchronize()
                gpu_lib.mask_conv_tile_mode(-1)
                want = exact + (res.double() if use_res else 0)
                if relu:
                    want = want.relu()
                err = (y.double() - want).abs()
                assert torch.isfinite(y).all() and bool((err <= bound).all()), (cin, cout, k, stride, pad, use_res, relu, tile_mode, err.max().item())
    # one launch large enough for the automatic rule to pick the 128 x 128 kernel by itself (>= 1 024 wide work-groups)
    x = torch.randn(4, 64, 138, 138, device="cuda").contiguous(memory_format=cl)
    wgt = (torch.randn(256, 64, 3, 3, device="cuda") / 24.0).contiguous(memory_format=cl)
    assert "<2, 2, 2, true>" in gpu_lib.mask_conv_kernel_name(4, 138, 138, 64, 256, 3, 3, 1, 1)
    y = torch.full((4, 256, 138, 138), float("nan"), device="cuda").contiguous(memory_format=cl)
    gpu_lib.mask_conv(st, x.data_ptr(), wgt.data_ptr(), None, None, y.data_ptr(), 4, 138, 138, 64, 256, 3, 3, 1, 1, False)
    exact = F.conv2d(x.double(), wgt.double(), None, 1, 1)
    bound = 1e-5 * F.conv2d(x.double().abs(), wgt.double().abs(), None, 1, 1) + 1e-6
    assert bool(((y.double() - exact).abs() <= bound).all())
    monkeypatch.delenv("AMOS_MASK_CONV1X1")
    # the rule of the automatic choice: large launches; one-frame launches of at least ~100 work-groups with <= 512 input channels
    big, small = torch.empty(32, 64, 138, 138, device="meta"), torch.empty(1, 64, 138, 138, device="meta")
    c = torch.nn.Conv2d(64, 256, 1)
    assert net_mod._gemm_conv(c, big) and net_mod._gemm_conv(c, small)                                            # 149 x 4 groups of 2 k-stages
    assert net_mod._gemm_conv(torch.nn.Conv2d(256, 1024, 1), torch.empty(1, 256, 35, 35, device="meta"))          # 10 x 16
    assert not net_mod._gemm_conv(torch.nn.Conv2d(1024, 256, 1), torch.empty(1, 1024, 35, 35, device="meta"))    # 10 x 4 groups of 32 k-stages: the library's
    assert not net_mod._gemm_conv(torch.nn.Conv2d(512, 2048, 1), torch.empty(1, 512, 18, 18, device="meta"))     # 3 x 32
    assert net_mod._gemm_conv(torch.nn.Conv2d(1024, 256, 1), torch.empty(32, 1024, 35, 35, device="meta"))       # 614 work-groups
    assert net_mod._gemm_conv(torch.nn.Conv2d(2048, 512, 1), torch.empty(32, 2048, 18, 18, device="meta"))       # 324 wide = 648 narrow
    assert not net_mod._gemm_conv(torch.nn.Conv2d(2048, 256, 1), torch.empty(32, 2048, 18, 18, device="meta"))   # 324 narrow
    c3 = torch.nn.Conv2d(64, 64, 3, padding=1).to(memory_format=cl)
    assert net_mod._gemm_conv(c3, big) and net_mod._gemm_conv(c3, small)   # one frame: 149 groups of 18 k-stages beat the library kernel + the bias pass
    assert not net_mod._gemm_conv(torch.nn.Conv2d(128, 128, 3, padding=1).to(memory_format=cl), torch.empty(1, 128, 69, 69, device="meta"))   # 38 x 2
    assert not net_mod._gemm_conv(torch.nn.Conv2d(256, 243, 3, padding=1).to(memory_format=cl), torch.empty(32, 256, 69, 69, device="meta"))  # 243 channels
    s2 = torch.nn.Conv2d(256, 256, 3, padding=1, stride=2).to(memory_format=cl)
    assert net_mod._gemm_conv(s2, torch.empty(32, 256, 69, 69, device="meta"))          # strided: 307 x 4 work-groups of 128 x 64
    assert not net_mod._gemm_conv(torch.nn.Conv2d(512, 512, 3, padding=1, stride=2).to(memory_format=cl), torch.empty(32, 512, 35, 35, device="meta"))  # 81 x 8
    monkeypatch.setenv("AMOS_MASK_CONV3X3", "0")
    assert not net_mod._gemm_conv(c3, big)
    monkeypatch.delenv("AMOS_MASK_CONV3X3")
    with pytest.raises(RuntimeError):
        gpu_lib.mask_conv1x1(st, 0, 0, None, None, 0, 1, 8, 8, 64, 64, 1, True)


@pytest.mark.gpu
def test_split_k_gemm_small_launches(mask, gpu_lib):
    """amos_mask_conv_ws_device: the k loop of a small launch cut into splits, the last work-group to arrive at a tile adding the splits in
    split order and running the fused epilogue (one launch; the splits of a tile meet in one XCD's L2, which every group verifies from its
    XCC_ID).  One-frame shapes of the network from 3 to 152 tiles, 1 x 1 / 3 x 3 / strided, with and without residual: against a float64
    convolution to the direct kernel's bound, twenty launches bit-identical (the sum does not depend on which group arrives last), the
    tile counters left zero, and the same library call without a workspace (no split) within float32 rounding."""
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(31)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    n_split = 0
    for ci, co, k, s, H, res in ((2048, 512, 1, 1, 18, 0), (1024, 256, 1, 1, 35, 1), (512, 2048, 1, 1, 18, 1), (256, 256, 3, 1, 35, 0), (256, 256, 3, 2, 18, 0),
                                 (256, 384, 3, 1, 18, 0), (512, 128, 1, 1, 69, 0), (256, 256, 3, 1, 69, 1), (1024, 2048, 1, 2, 35, 0), (64, 64, 1, 1, 138, 0)):
        pad = k // 2
        x = torch.randn(1, ci, H, H, device="cuda").contiguous(memory_format=cl)
        w = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).contiguous(memory_format=cl)
        b = torch.randn(co, device="cuda")
        Ho = (H + 2 * pad - k) // s + 1
        r = torch.randn(1, co, Ho, Ho, device="cuda").contiguous(memory_format=cl) if res else None
        need = gpu_lib.mask_conv_workspace_bytes(1, H, H, ci, co, k, k, s, pad)
        assert need <= ws.numel()
        n_split += need > 0
        outs = []
        for _ in range(20):
            y = torch.full((1, co, Ho, Ho), float("nan"), device="cuda").contiguous(memory_format=cl)
            gpu_lib.mask_conv_ws(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), 1, H, H, ci, co, k, k, s, pad, True,
                                 ws.data_ptr(), ws.numel())
            outs.append(y)
        torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), (ci, co, k, s, H)
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "tile counters must be left zero"
        want = F.conv2d(x.double(), w.double(), b.double(), s, pad) + (r.double() if res else 0)
        bound = 1e-5 * (F.conv2d(x.double().abs(), w.double().abs(), None, s, pad) + 1)
        assert bool(((outs[0].double() - want.relu()).abs() <= bound).all()), (ci, co, k, s, H)
        plain = torch.empty_like(outs[0])
        gpu_lib.mask_conv(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, plain.data_ptr(), 1, H, H, ci, co, k, k, s, pad, True)
        torch.cuda.synchronize()
        assert float((plain - outs[0]).abs().max()) <= 1e-4 * max(float(plain.abs().max()), 1.0)
        if need == 0:
            assert torch.equal(plain, outs[0])   # the plan for this shape is an ordinary launch
    assert n_split >= 8
    with pytest.raises(gpu_lib.AmosError):   # a workspace too small for the plan is refused, not overrun
        gpu_lib.mask_conv_ws(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 1, 18, 18, 2048, 512, 1, 1, 1, 0, True, ws.data_ptr(), 20000)


@pytest.mark.gpu
def test_strided_3x3_layer_on_the_automatic_tile_rule_at_32_frames(mask, gpu_lib):
    """The strided-layer dispatch (net.py _gemm_conv: groups recomputed at 128 x 64 when stride >= 2) at the launch the bench issues: the
    256 -> 256 stride-2 3 x 3 layer, 69 -> 35 at 32 frames (307 x 4 narrow work-groups), through conv_bias_act under the AUTOMATIC
    tile mode, against a float64 convolution with the bound of the other kernels (float32 rounding of a 2 304-term sum)."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(21)
    conv = torch.nn.Conv2d(256, 256, 3, stride=2, padding=1).cuda().to(memory_format=cl)
    x = torch.randn(32, 256, 69, 69, device="cuda").contiguous(memory_format=cl)
    assert net_mod._gemm_conv(conv, x)
    name = gpu_lib.mask_conv_kernel_name(32, 69, 69, 256, 256, 3, 3, 2, 1)
    assert "k_conv_gemm" in name and "<2, 2, 2, true>" not in name, name   # the rule's 128 x 64 work-groups, not the 128 x 128 ones
    calls = []
    real = gpu_lib.mask_conv
    try:
        gpu_lib.mask_conv = lambda *a: (calls.append(a[6:15]), real(*a))[1]
        with torch.no_grad():
            y = net_mod.conv_bias_act(conv, x, True)
    finally:
        gpu_lib.mask_conv = real
    torch.cuda.synchronize()
    assert calls == [(32, 69, 69, 256, 256, 3, 3, 2, 1)] and y.shape == (32, 256, 35, 35) and y.is_contiguous(memory_format=cl)
    with torch.no_grad():
        for lo in range(0, 32, 8):  # float64 reference in slices (memory)
            xs = x[lo:lo + 8].double()
            want = F.conv2d(xs, conv.weight.double(), conv.bias.double(), 2, 1).relu()
            bound = 1e-5 * F.conv2d(xs.abs(), conv.weight.double().abs(), None, 2, 1) + 1e-6
            assert bool(((y[lo:lo + 8].double() - want).abs() <= bound).all()), lo


@pytest.mark.gpu
def test_first_forward_on_a_second_stream_after_prepare(mask, gpu_lib):
    """The Winograd-transformed weights are shared by every stream: MaskEngine.prepare() makes all of them (and waits), so a FIRST
    forward issued on a non-default stream -- and two lanes' first forwards issued back to back on two streams, as bench.py's lanes
    do -- read finished weights.  The results must equal the default-stream forward of a second engine with the same seed."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    eng = _engine(mask, "cuda:0", "seed0").prepare()
    eligible = [m for m in list(eng.net.modules()) + [eng.net.prediction_layers[0].merged] if isinstance(m, torch.nn.Conv2d)
                and m.kernel_size == (3, 3) and m.stride == (1, 1) and gpu_lib.mask_winograd_supported(m.in_channels, m.out_channels)]
    assert len(eligible) >= 20 and all(getattr(m, "_amos_winograd", None) is not None for m in eligible)
    made = []
    real = (gpu_lib.mask_winograd_weights, gpu_lib.mask_winograd24_weights)
    frames = torch.from_numpy(np.stack([mask_cases.frame(c) for c in ("seed0", "ref122_w0", "tum_w0", "blobs7_w1")] * 8)).cuda()
    x = eng._preprocess_hip(frames)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        gpu_lib.mask_winograd_weights = lambda *a: (made.append(a), real[0](*a))[1]
        gpu_lib.mask_winograd24_weights = lambda *a: (made.append(a), real[1](*a))[1]
        with torch.no_grad():
            with torch.cuda.stream(s1):
                p1 = eng._forward(x)
            with torch.cuda.stream(s2):
                p2 = eng._forward(x)
    finally:
        gpu_lib.mask_winograd_weights, gpu_lib.mask_winograd24_weights = real
    torch.cuda.synchronize()
    assert made == [], "a forward after prepare() must not create shared weight tensors"
    ref = _engine(mask, "cuda:0", "seed0").prepare()
    with torch.no_grad():
        want = ref._forward(x)
    torch.cuda.synchronize()
    for k in ("loc", "conf", "mask", "proto"):   # (the library's convolutions are not bit-reproducible from call to call: float32 rounding)
        for got in (p1[k], p2[k]):
            assert bool(torch.isfinite(got).all()) and float((got - want[k]).abs().max()) <= 2e-4 * max(float(want[k].abs().max()), 1.0), k
    # a layer transformed lazily (no prepare) on a side stream is published only after its stream has drained
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda().to(memory_format=torch.channels_last)
    with torch.cuda.stream(s1):
        u = net_mod._winograd_weight(conv)
    assert s1.query() and bool(torch.isfinite(u).all())


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["22", "24"])
def test_winograd_conv3x3_against_float64(mask, gpu_lib, family):
    """amos_mask_winograd_conv_device (Winograd F(2 x 2, 3 x 3)) and amos_mask_winograd24_conv_device (F(2 x 4, 3 x 3): F(2, 3) vertically,
    F(4, 3) horizontally) on the fp32 MFMA units -- input transform, 16 / 24 GEMMs, output transform, bias + residual + ReLU in one kernel --
    against a float64 convolution, to the bound the direct kernels are held to (1e-5 of the sum of |terms|): odd and even image sizes
    (half-empty last tiles, widths of every residue modulo 4), images smaller than a tile block, tile blocks spanning frames, channel
    counts from four stages (32) to many, tile runs of many one-tile segments (40 frames of 2 x 4, 70 of 2 x 2), launches of
    600 - 1 200 work-groups, every epilogue combination; and the transformed weight against G g G^T in float64."""
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_lib.mask_winograd_supported(32, 64) and gpu_lib.mask_winograd_supported(256, 384)
    assert not gpu_lib.mask_winograd_supported(16, 64) and not gpu_lib.mask_winograd_supported(40, 64) and not gpu_lib.mask_winograd_supported(64, 32)
    assert not gpu_lib.mask_winograd_supported(64, 352)
    G2 = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64, device="cuda")
    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                      dtype=torch.float64, device="cuda")
    Gc, npos, make_weights, run_conv = ((G2, 16, gpu_lib.mask_winograd_weights, gpu_lib.mask_winograd_conv) if family == "22" else
                                        (G4, 24, gpu_lib.mask_winograd24_weights, gpu_lib.mask_winograd24_conv))
    worst = 0.0
    # F(2 x 4): both launch forms (one work-group per id / the persistent walk), forced in turn; every output of the second form must equal the first's
    forms = (0, 1) if family == "24" else (None,)   # (the 32-channel-per-group form is held to the same bound and the same bits further down)
    for b, cin, cout, h, w in ((1, 32, 64, 6, 6), (2, 64, 64, 21, 17), (1, 32, 128, 9, 9), (3, 256, 64, 5, 5), (3, 48, 192, 12, 7), (2, 32, 64, 1, 1),
                               (1, 64, 128, 2, 37), (5, 80, 64, 7, 3), (40, 32, 64, 2, 4), (70, 32, 64, 2, 2), (3, 32, 64, 1, 1), (2, 128, 256, 35, 35), (1, 256, 384, 69, 69),
                               (2, 32, 64, 3, 130), (1, 32, 64, 9, 127), (33, 32, 64, 4, 1),
                               (8, 64, 64, 138, 138), (4, 32, 128, 138, 138)):   # the last two: several groups of work-groups per XCD, the last one partly empty
        x = torch.randn(b, cin, h, w, device="cuda").contiguous(memory_format=cl)
        wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
        bias = torch.randn(cout, device="cuda")
        res = torch.randn(b, cout, h, w, device="cuda").contiguous(memory_format=cl)
        u = torch.full((npos * cin * cout,), float("nan"), device="cuda")
        make_weights(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
        # the transformed weight in MFMA fragment order: [cout tile][stage][position][32-channel block][k half][32 channels][4]
        U = G2 @ wgt.double() @ Gc.T                                                                     # [cout][cin][4][4 or 6]
        img = u.view(cout // 64, cin // 8, npos, 2, 2, 32, 4)
        want_img = U.reshape(cout // 64, 2, 32, cin // 8, 2, 4, npos).permute(0, 3, 6, 1, 4, 2, 5)
        assert bool(((img.double() - want_img).abs() <= 6e-8 * want_img.abs() + 1e-30).all()), (cin, cout)   # one rounding of a double sum
        exact = F.conv2d(x.double(), wgt.double(), None, 1, 1)
        bound = 1e-5 * F.conv2d(x.double().abs(), wgt.double().abs(), None, 1, 1) + 1e-6
        for use_bias, use_res, relu in ((True, True, True), (True, False, True), (False, True, False), (False, False, False), (True, False, False)):
            first = None
            for form in forms:
                if form is not None:
                    gpu_lib.mask_winograd24_persistent_mode(form)
                y = torch.full((b, cout, h, w), float("nan"), device="cuda").contiguous(memory_format=cl)
                try:
                    run_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr() if use_bias else None, res.data_ptr() if use_res else None, y.data_ptr(),
                             b, h, w, cin, cout, relu)
                    torch.cuda.synchronize()
                finally:
                    if form is not None:
                        gpu_lib.mask_winograd24_persistent_mode(-1)
                want = exact + (bias.double().view(1, -1, 1, 1) if use_bias else 0) + (res.double() if use_res else 0)
                if relu:
                    want = want.relu()
                err = (y.double() - want).abs()
                assert torch.isfinite(y).all() and bool((err <= bound).all()), (family, form, b, cin, cout, h, w, use_bias, use_res, relu, err.max().item())
                worst = max(worst, float((err / bound).max()))
                if first is None:
                    first = y
                else:
                    assert torch.equal(y, first), ("launch forms differ", b, cin, cout, h, w, use_bias, use_res, relu)
    print("winograd family", family, "worst error / bound", worst)
    assert worst < (0.2 if family == "22" else 0.6)   # F(2 x 2): measured 0.02 (the transforms cost less rounding than the 9-tap sums save); F(2 x 4): a few times that
    if family == "24":
        with pytest.raises(RuntimeError):
            gpu_lib.mask_winograd24_conv(st, 0, 0, None, None, 0, 1, 8, 8, 64, 64, True)
        return
    with pytest.raises(RuntimeError):
        gpu_lib.mask_winograd_conv(st, 0, 0, None, None, 0, 1, 8, 8, 64, 64, True)
    # through the network's dispatch: the rule, both forced sides, the cached transformed weight following an in-place weight update
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    conv = torch.nn.Conv2d(64, 128, 3, padding=1).cuda().to(memory_format=cl)
    big, small = torch.empty(32, 64, 69, 69, device="meta"), torch.empty(1, 64, 35, 35, device="meta")
    assert net_mod._winograd_conv(conv, big) and not net_mod._winograd_conv(conv, small)
    assert not net_mod._winograd_conv(torch.nn.Conv2d(64, 128, 3, padding=1, stride=2), big) and not net_mod._winograd_conv(torch.nn.Conv2d(64, 100, 3, padding=1), big)


@pytest.mark.gpu
def test_winograd24_32_channels_per_work_group(mask, gpu_lib):
    """The F(2 x 4) kernel's small-launch form (32 output channels per work-group: twice the groups, channel tile nt = block nt & 1 of U's
    64-channel slice nt >> 1) forced on and off: the same bits as the 64-channel form on one-frame and multi-frame shapes, every epilogue
    combination, cout = 64 (one slice) to 384 (an odd number of 64-slices x 2); and the automatic choice picks it for one-frame launches."""
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(23)
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_lib.mask_winograd24_narrow_mode() == -1
    for b, cin, cout, h, w in ((1, 256, 256, 69, 69), (1, 64, 64, 138, 138), (1, 256, 384, 69, 69), (3, 128, 128, 33, 35), (2, 32, 64, 9, 7), (1, 48, 192, 5, 5)):
        x = torch.randn(b, cin, h, w, device="cuda").contiguous(memory_format=cl)
        wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
        bias = torch.randn(cout, device="cuda")
        res = torch.randn(b, cout, h, w, device="cuda").contiguous(memory_format=cl)
        u = torch.empty(24 * cin * cout, device="cuda")
        gpu_lib.mask_winograd24_weights(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
        exact = F.conv2d(x.double(), wgt.double(), None, 1, 1)
        bound = 1e-5 * F.conv2d(x.double().abs(), wgt.double().abs(), None, 1, 1) + 1e-6
        for use_bias, use_res, relu in ((True, True, True), (False, False, False), (True, False, True)):
            ys = {}
            for mode in (0, 1, -1):
                gpu_lib.mask_winograd24_narrow_mode(mode)
                try:
                    y = torch.full((b, cout, h, w), float("nan"), device="cuda").contiguous(memory_format=cl)
                    gpu_lib.mask_winograd24_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr() if use_bias else None, res.data_ptr() if use_res else None,
                                                 y.data_ptr(), b, h, w, cin, cout, relu)
                    torch.cuda.synchronize()
                finally:
                    gpu_lib.mask_winograd24_narrow_mode(-1)
                ys[mode] = y
            want = exact + (bias.double().view(1, -1, 1, 1) if use_bias else 0) + (res.double() if use_res else 0)
            if relu:
                want = want.relu()
            assert bool(((ys[1].double() - want).abs() <= bound).all()), (b, cin, cout, h, w)
            assert torch.equal(ys[0], ys[1]) and torch.equal(ys[-1], ys[0]), (b, cin, cout, h, w, use_bias, use_res, relu)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["22", "24"])
def test_winograd_conv3x3_at_the_headline_launch_size(mask, gpu_lib, family):
    """The largest launch of the headline run as it is made there: proto_net's 256 -> 256 channels at 138 x 138 with the frames per
    forward of bench.CONFIGS["c3"] (64: a 1.25 GB input, past 1 GiB of the 2 GiB a launch's buffer descriptor can address, several groups
    of work-groups per XCD).  Frames 0, 1, the middle pair and the last one against a float64 convolution to the direct kernels' bound;
    EVERY frame bit for bit against the same kernel launched on 8 frames at a time (a frame's arithmetic does not depend on the launch
    it rides in), so an index that wraps or a group decoded wrongly anywhere in the large launch shows."""
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    b, cin, cout, h, w = _bench_frames_per_forward(), 256, 256, 138, 138
    assert b * h * w * cin * 4 > 2 ** 30 and b * h * w * cin * 4 < 2 ** 31 - 4096
    npos, make_weights, run_conv = ((16, gpu_lib.mask_winograd_weights, gpu_lib.mask_winograd_conv) if family == "22" else
                                    (24, gpu_lib.mask_winograd24_weights, gpu_lib.mask_winograd24_conv))
    x = torch.randn(b, cin, h, w, device="cuda").contiguous(memory_format=cl)
    wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
    bias = torch.randn(cout, device="cuda")
    u = torch.empty(npos * cin * cout, device="cuda")
    make_weights(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
    y = torch.full((b, cout, h, w), float("nan"), device="cuda").contiguous(memory_format=cl)
    run_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), None, y.data_ptr(), b, h, w, cin, cout, True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    for f in sorted({0, 1, b // 2 - 1, b // 2, b - 1}):
        xf = x[f:f + 1].double()
        want = (F.conv2d(xf, wgt.double(), None, 1, 1) + bias.double().view(1, -1, 1, 1)).relu()
        bound = 1e-5 * F.conv2d(xf.abs(), wgt.double().abs(), None, 1, 1) + 1e-6
        err = (y[f:f + 1].double() - want).abs()
        assert bool((err <= bound).all()), (family, f, float((err / bound).max()))
    if family == "24":   # the other launch form on the whole launch: the same bits
        assert gpu_lib.mask_winograd24_persistent_mode(1) == -1   # (automatic until here: one work-group per id; now the persistent walk)
        try:
            y0 = torch.full((b, cout, h, w), float("nan"), device="cuda").contiguous(memory_format=cl)
            run_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), None, y0.data_ptr(), b, h, w, cin, cout, True)
            torch.cuda.synchronize()
        finally:
            gpu_lib.mask_winograd24_persistent_mode(-1)
        assert torch.equal(y0, y)
        del y0
    part = torch.empty((8, cout, h, w), device="cuda").contiguous(memory_format=cl)
    for f0 in range(0, b, 8):
        part.fill_(float("nan"))
        xs = x[f0:f0 + 8]
        assert xs.is_contiguous(memory_format=cl)
        run_conv(st, xs.data_ptr(), u.data_ptr(), bias.data_ptr(), None, part.data_ptr(), 8, h, w, cin, cout, True)
        torch.cuda.synchronize()
        assert torch.equal(part, y[f0:f0 + 8]), (family, f0)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["22", "24"])
def test_winograd_dispatch_and_weight_cache(mask, gpu_lib, monkeypatch, family):
    monkeypatch.setenv("AMOS_MASK_WINOGRAD_F", family)
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    F = torch.nn.functional
    cl = torch.channels_last
    torch.manual_seed(12)
    conv = torch.nn.Conv2d(64, 128, 3, padding=1).cuda().to(memory_format=cl)
    x = torch.randn(2, 64, 23, 31, device="cuda").contiguous(memory_format=cl)
    res = torch.randn(2, 128, 23, 31, device="cuda").contiguous(memory_format=cl)
    calls = []
    _spy_winograd(gpu_lib, monkeypatch, calls, 7, 11)
    with torch.no_grad():
        for round_ in range(2):
            want = (F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), 1, 1) + res.double()).relu()
            bound = 1e-5 * F.conv2d(x.double().abs(), conv.weight.double().abs(), None, 1, 1) + 1e-6
            outs = {}
            for side in ("0", "2"):
                monkeypatch.setenv("AMOS_MASK_WINOGRAD", side)
                outs[side] = net_mod.conv_bias_act(conv, x, True, residual=res)
            assert len(calls) == 2 * round_ + 1   # only the forced side called the kernel (conv_raw below adds one more per round)
            for side, y in outs.items():
                assert y.is_contiguous(memory_format=cl) and bool(((y.double() - want).abs() <= bound).all()), side
            raw = net_mod.conv_raw(conv, x)     # the convolution alone (what the fused prediction head uses)
            assert bool(((raw.double() - F.conv2d(x.double(), conv.weight.double(), None, 1, 1)).abs() <= bound).all())
            conv.weight.mul_(0.5)               # an in-place update must refresh the cached transformed weight
    assert len(calls) == 4


@pytest.mark.gpu
def test_hip_bilinear_nhwc_equals_torch_interpolate(mask, gpu_lib):
    """amos_mask_bilinear_nhwc_device against F.interpolate(mode="bilinear", align_corners=False) on channels-last tensors: the FPN's
    size-given form (35 -> 69, 18 -> 35) and the prototype network's scale_factor = 2 (69 -> 138).  Same source indices and
    weights; PyTorch's kernel contracts multiply-adds, hence a few ulp."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    torch.manual_seed(6)
    for (n, c, h, w), kw in (((3, 256, 35, 35), dict(size=(69, 69))), ((2, 256, 18, 18), dict(size=(35, 35))), ((2, 256, 69, 69), dict(scale_factor=2)),
                             ((1, 8, 5, 7), dict(size=(11, 13)))):
        x = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
        got = net_mod.bilinear(x, **kw)
        want = torch.nn.functional.interpolate(x, mode="bilinear", align_corners=False, **kw)
        torch.cuda.synchronize()
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max())), (n, c, h, w, kw)
    # an exact x 2 enlargement takes the 2 x 2-outputs-per-thread form: the same bits as the one-output form, odd and tiny sources, with and
    # without the ReLU, a NaN in the source
    for (n, c, h, w) in ((2, 256, 69, 69), (1, 8, 5, 7), (3, 4, 1, 1), (1, 12, 2, 9), (2, 64, 35, 34)):
        x = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
        if h > 2:
            x[0, 1, 2, 3] = float("nan")
        for relu in (False, True):
            assert gpu_lib.mask_bilinear_x2_mode(1) in (0, 1)
            a = net_mod.bilinear(x, scale_factor=2, relu=relu)
            gpu_lib.mask_bilinear_x2_mode(0)
            b = net_mod.bilinear(x, scale_factor=2, relu=relu)
            gpu_lib.mask_bilinear_x2_mode(1)
            torch.cuda.synchronize()
            assert a.shape == (n, c, 2 * h, 2 * w) and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), (n, c, h, w, relu)


@pytest.mark.gpu
def test_nms_suppression_kernel_equals_the_torch_ops(mask, gpu_lib):
    """amos_mask_nms_column_max_device against `jaccard(boxes, boxes).triu_(diagonal=1).max(dim=1)` as PyTorch computes it on the same
    GPU: random boxes, heavily overlapping boxes (many IoUs around the 0.5 threshold), duplicates (IoU exactly 1), k = 1 and k = 200.
    Bit-identical; then detect_batch end to end against the reference-order detect() (test_static_shape_batch_post... covers the CPU)."""
    det = importlib.import_module("amos_slam_amd.mask.detect")
    torch.manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    for lists, k, spread in ((7, 200, 1.0), (160, 200, 0.05), (3, 1, 1.0), (5, 37, 0.2), (2, 256, 0.1)):
        c = torch.rand(lists, k, 2, device="cuda") * spread + 0.3
        wh = torch.rand(lists, k, 2, device="cuda") * 0.2 + 0.05
        boxes = torch.cat((c - wh / 2, c + wh / 2), -1)
        if k > 4:
            boxes[:, 3] = boxes[:, 1]  # a duplicate: IoU 1 with a higher-scored box
        want = det._pairwise_iou(boxes).triu_(diagonal=1).max(dim=1)[0]
        got = torch.full((lists, k), float("nan"), device="cuda")
        gpu_lib.mask_nms_column_max(st, boxes.contiguous().data_ptr(), got.data_ptr(), lists, k)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (lists, k, (got - want).abs().max().item())
        assert got[:, 0].abs().max().item() == 0.0
        via = det._suppression_term(boxes.view(1, lists, k, 4))
        assert torch.equal(via.view(lists, k), want)
    # a degenerate pair (two zero-area boxes at the same place): 0 / 0 = NaN wins the maximum, as in torch.max
    boxes = torch.tensor([[[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5], [0.1, 0.1, 0.2, 0.2]]], device="cuda")
    want = det._pairwise_iou(boxes).triu_(diagonal=1).max(dim=1)[0]
    got = torch.zeros((1, 3), device="cuda")
    gpu_lib.mask_nms_column_max(st, boxes.data_ptr(), got.data_ptr(), 1, 3)
    torch.cuda.synchronize()
    assert torch.isnan(want[0, 1]) and torch.isnan(got[0, 1]) and got[0, 0] == 0 and got[0, 2] == want[0, 2]
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.mask_nms_column_max(st, boxes.data_ptr(), got.data_ptr(), 1, 257)


@pytest.mark.gpu
def test_bilinear_with_relu_equals_relu_of_bilinear(mask, gpu_lib):
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    torch.manual_seed(12)
    x = torch.randn(2, 256, 69, 69, device="cuda").contiguous(memory_format=torch.channels_last)
    plain = net_mod.bilinear(x, scale_factor=2)
    fused = net_mod.bilinear(x, scale_factor=2, relu=True)
    assert fused.is_contiguous(memory_format=torch.channels_last) and torch.equal(fused, torch.relu(plain)) and bool((fused == 0).any())
    cpu = net_mod.bilinear(x.cpu(), scale_factor=2, relu=True)
    assert torch.allclose(cpu, fused.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_class_scores_kernel_equals_the_torch_ops(mask, gpu_lib):
    """amos_mask_class_scores_device against detect_batch's torch form (transpose, drop background, max over classes > 0.05, where):
    exact, incl. a ragged last tile of priors and scores exactly at the threshold."""
    torch.manual_seed(13)
    st = torch.cuda.current_stream().cuda_stream
    for B, P, C1 in ((3, 19248, 81), (2, 100, 81), (1, 65, 5)):
        conf = torch.softmax(torch.randn(B, P, C1, device="cuda") * 3, -1)
        conf[0, 3, 1:] = 0.05  # best class exactly at the threshold: not kept
        conf[0, 4, 2] = 0.9
        cls = conf.transpose(1, 2)[:, 1:, :]
        keep = cls.max(dim=1, keepdim=True)[0] > 0.05
        want = torch.where(keep, cls, torch.full_like(cls, -1.0)).contiguous()
        got = torch.full((B, C1 - 1, P), float("nan"), device="cuda")
        gpu_lib.mask_class_scores(st, conf.data_ptr(), got.data_ptr(), B, P, C1, 0.05)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and bool((got[0, :, 3] == -1).all()) and got[0, 1, 4] == conf[0, 4, 2]


@pytest.mark.gpu
def test_person_mask_kernel_equals_the_torch_ops(mask, gpu_lib):
    """amos_mask_person_mask_device against F.interpolate(...) > 0.5, & flags, sum, (x 255).byte(): smooth random masks (values cross 0.5
    along curves, as sigmoid masks do), 0 .. 15 flagged detections, more than one overlapping detection (255 * 2 wraps to 254).
    PyTorch's upsample kernel may contract its multiply-adds differently: a pixel whose interpolated value is within an ulp of 0.5
    may flip, hence the bar of 1e-6 of the pixels (none observed)."""
    F = torch.nn.functional
    torch.manual_seed(14)
    st = torch.cuda.current_stream().cuda_stream
    # the frame's own size (source windows staged in LDS), a size that is not a multiple of the 64 x 4 tile, an output SMALLER than the masks
    # and more detections than the staging area holds (both: per-thread gathers)
    for n, h, w in ((15, 480, 640), (15, 333, 517), (15, 100, 120), (17, 480, 640)):
        B, ph, pw = 4, 138, 138
        base = F.interpolate(torch.randn(B, n, 9, 9, device="cuda"), (ph, pw), mode="bicubic", align_corners=False)
        masks = torch.sigmoid(base * 4).contiguous()
        masks[:, :, :20] = 0  # cropped region
        flags = (torch.rand(B, n, device="cuda") < 0.4)
        flags[0] = False
        flags[1] = True
        want_b = F.interpolate(masks, (h, w), mode="bilinear", align_corners=False) > 0.5
        total = (want_b & flags[..., None, None]).sum(dim=1)
        want = ((total.to(torch.int64) * 255) & 0xFF).to(torch.uint8)
        got = torch.full((B, h, w), 7, dtype=torch.uint8, device="cuda")
        gpu_lib.mask_person_mask(st, masks.data_ptr(), flags.to(torch.uint8).contiguous().data_ptr(), got.data_ptr(), B, n, ph, pw, h, w)
        torch.cuda.synchronize()
        differing = int((got != want).sum())
        assert differing <= max(1e-6 * got.numel(), 1 if h < ph else 0), (n, h, w, differing)
        assert int(got[0].max()) == 0 and int((got[1] == ((255 * 2) & 0xFF)).sum()) > 0, (n, h, w)


@pytest.mark.gpu
def test_bias_relu_maxpool_kernel_equals_the_torch_ops(mask, gpu_lib):
    """amos_mask_bias_relu_maxpool_device against F.max_pool2d(F.relu(x + bias), 3, 2, 1) on channels-last tensors: odd and even sizes
    (the stem's 275 x 275 and others), negative maxima (ReLU clamps after the max), a NaN in one window.  Bit-identical."""
    F = torch.nn.functional
    torch.manual_seed(15)
    st = torch.cuda.current_stream().cuda_stream
    cl = torch.channels_last
    for n, c, h, w in ((2, 64, 275, 275), (1, 8, 6, 9), (3, 4, 1, 1), (1, 64, 14, 15)):
        x = (torch.randn(n, c, h, w, device="cuda") - 0.5).contiguous(memory_format=cl)
        bias = torch.randn(c, device="cuda")
        if h > 4:
            x[0, 1, 3, 4] = float("nan")
        want = F.max_pool2d(F.relu(x + bias.view(1, -1, 1, 1)), 3, stride=2, padding=1)
        got = torch.full(want.shape, 7.0, device="cuda").contiguous(memory_format=cl)
        gpu_lib.mask_bias_relu_maxpool(st, x.data_ptr(), bias.data_ptr(), got.data_ptr(), n, h, w, c)
        torch.cuda.synchronize()
        assert got.shape == want.shape and torch.equal(torch.isnan(got), torch.isnan(want)), (n, c, h, w)
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)), (n, c, h, w, (torch.nan_to_num(got) - torch.nan_to_num(want)).abs().max().item())


@pytest.mark.gpu
def test_stem_kernel_against_float64(mask, gpu_lib):
    """amos_mask_stem_device (conv 7 x 7 / 2, 3 -> 64, + bias + ReLU + max-pool 3 x 3 / 2 in one kernel) against the same chain in float64:
    the network's 550 x 550, sizes that leave partial 6 x 23 blocks in both directions, sizes smaller than one block and than the window,
    planar and channels-last inputs and weights (the entry point takes strides).  Error bound: 4e-7 of the sum of |terms| of the window
    the maximum came from (float32 summation of 147 products; the library's convolution is inside the same bound).  Then non-finite input:
    a NaN must reach exactly the pooled pixels whose windows hold it (the kernel's MFMA steps read one float past every window row; the
    library's convolution does NOT pass this: it spreads the NaN one window further)."""
    F = torch.nn.functional
    torch.manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    cl = torch.channels_last
    n_pack = gpu_lib.mask_stem_weight_floats()
    assert n_pack == 2 * 77 * 64
    for case, (b, h, w) in enumerate(((2, 550, 550), (1, 97, 131), (3, 64, 48), (1, 7, 9), (2, 1, 1), (1, 3, 200), (1, 189, 5), (1, 551, 93))):
        x = torch.randn(b, 3, h, w, device="cuda") * 2.0 + 0.3
        wt = torch.randn(64, 3, 7, 7, device="cuda") / 12.0
        bias = torch.randn(64, device="cuda") * 0.5
        if case % 2:
            x = x.contiguous(memory_format=cl)
            wt = wt.contiguous(memory_format=cl)
        packed = torch.empty(n_pack, device="cuda")
        gpu_lib.mask_stem_weights(st, wt.data_ptr(), wt.stride(), packed.data_ptr())
        ch, cw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        ph, pw = (ch - 1) // 2 + 1, (cw - 1) // 2 + 1
        got = torch.full((b, 64, ph, pw), 7.0, device="cuda").contiguous(memory_format=cl)
        gpu_lib.mask_stem(st, x.data_ptr(), x.stride(), packed.data_ptr(), bias.data_ptr(), got.data_ptr(), b, h, w)
        torch.cuda.synchronize()
        conv = F.conv2d(x.double(), wt.double(), bias.double(), 2, 3)
        want = F.max_pool2d(F.relu(conv), 3, stride=2, padding=1)
        assert want.shape == got.shape, (b, h, w)
        terms = F.max_pool2d(F.conv2d(x.double().abs(), wt.double().abs(), bias.double().abs(), 2, 3), 3, stride=2, padding=1)  # >= the winning window's
        err = float(((got.double() - want).abs() / (terms + 1.0)).max())
        assert err < 4e-7, (b, h, w, err)
        # and against the float32 chain the kernel replaces (library convolution + the project's bias / ReLU / max-pool pass): same bound
        lib = F.max_pool2d(F.relu(F.conv2d(x, wt, bias, 2, 3)), 3, stride=2, padding=1)
        assert float(((lib.double() - want).abs() / (terms + 1.0)).max()) < 4e-7
    # a NaN in the input: the same pooled pixels as in torch's chain, everything else unchanged
    x = torch.randn(1, 3, 60, 120, device="cuda")
    wt = torch.randn(64, 3, 7, 7, device="cuda") / 12.0
    bias = torch.randn(64, device="cuda")
    gpu_lib.mask_stem_weights(st, wt.data_ptr(), wt.stride(), packed.data_ptr())
    clean = torch.empty((1, 64, 15, 30), device="cuda").contiguous(memory_format=cl)
    gpu_lib.mask_stem(st, x.data_ptr(), x.stride(), packed.data_ptr(), bias.data_ptr(), clean.data_ptr(), 1, 60, 120)
    for (c, yy, xx) in ((0, 20, 33), (2, 0, 0), (1, 59, 119), (0, 31, 92)):
        xn = x.clone()
        xn[0, c, yy, xx] = float("nan")
        got = torch.empty_like(clean)
        gpu_lib.mask_stem(st, xn.data_ptr(), xn.stride(), packed.data_ptr(), bias.data_ptr(), got.data_ptr(), 1, 60, 120)
        torch.cuda.synchronize()
        # (the float64 chain on the CPU: the GPU library's own float32 convolution spreads the NaN to a neighbouring column / row of windows --
        # its padded GEMM multiplies it by zero weights -- 768 instead of 576 pooled values for the first case, measured)
        want_nan = torch.isnan(F.max_pool2d(F.relu(F.conv2d(xn.cpu().double(), wt.cpu().double(), bias.cpu().double(), 2, 3)), 3, stride=2, padding=1)).cuda()
        cys, cxs = [cy for cy in range(30) if 2 * cy - 3 <= yy <= 2 * cy + 3], [cx for cx in range(60) if 2 * cx - 3 <= xx <= 2 * cx + 3]
        n_rows = len({py for py in range(15) for cy in cys if 2 * py - 1 <= cy <= 2 * py + 1})
        n_cols = len({px for px in range(30) for cx in cxs if 2 * px - 1 <= cx <= 2 * px + 1})
        assert int(want_nan.sum()) == 64 * n_rows * n_cols and torch.equal(torch.isnan(got), want_nan), (c, yy, xx)
        assert torch.equal(got[~want_nan], clean[~want_nan])
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.mask_stem(st, x.data_ptr(), x.stride(), packed.data_ptr(), bias.data_ptr(), clean.data_ptr(), 0, 60, 120)


@pytest.mark.gpu
def test_trunk_with_the_stem_kernel_equals_the_library_stem(mask, gpu_lib, monkeypatch):
    """ResNet50Trunk.forward with the project's stem kernel (the default) against AMOS_MASK_STEM=library (library convolution + the bias /
    ReLU / max-pool kernel) on the same folded weights: the stem output within float32 rounding of a 147-term sum, for a planar input (what the
    pre-processing writes) and a channels-last one; the packed weight follows an in-place weight change; no packed weight is made inside a
    graph capture."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    torch.manual_seed(22)
    cl = torch.channels_last
    trunk = net_mod.ResNet50Trunk().cuda().eval()
    with torch.no_grad():
        for m in trunk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
    trunk.fold_batch_norms()
    trunk.to(memory_format=cl)
    x = torch.randn(2, 3, 550, 550, device="cuda")
    stem_out = {}

    def hook(_, args):
        stem_out["x"] = args[0]

    h = trunk.layers[0].register_forward_pre_hook(hook)
    with torch.no_grad():
        for mode in ("own", "library"):
            monkeypatch.setenv("AMOS_MASK_STEM", mode)
            for inp in (x, x.contiguous(memory_format=cl)):
                trunk(inp)
                stem_out[(mode, inp.is_contiguous())] = stem_out.pop("x")
        ref = stem_out[("library", False)]  # (False: the channels-last input, which is not contiguous in torch's default sense)
        assert ref.shape == (2, 64, 138, 138) and torch.allclose(ref, stem_out[("library", True)], rtol=1e-5, atol=2e-5)
        for key in (("own", True), ("own", False)):
            assert stem_out[key].is_contiguous(memory_format=cl)
            assert torch.allclose(stem_out[key], ref, rtol=1e-5, atol=2e-5), float((stem_out[key] - ref).abs().max())
        assert torch.equal(stem_out[("own", True)], stem_out[("own", False)])  # the same sums whatever the input layout
        # an in-place weight change is followed
        monkeypatch.setenv("AMOS_MASK_STEM", "own")
        trunk.conv1.weight.mul_(0.5)
        trunk(x)
        half = stem_out.pop("x")
        monkeypatch.setenv("AMOS_MASK_STEM", "library")
        trunk(x)
        assert torch.allclose(half, stem_out.pop("x"), rtol=1e-5, atol=2e-5)
        # a packed weight is never made inside a capture
        monkeypatch.setenv("AMOS_MASK_STEM", "own")
        trunk.conv1.weight.mul_(2.0)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with pytest.raises(RuntimeError, match="capture"):
            with torch.cuda.graph(g, stream=s):
                trunk(x)
    h.remove()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_fused_head_outputs_equal_the_torch_ops(mask, gpu_lib, monkeypatch):
    """The prediction head's fused output path (amos_mask_head_outputs_device: bias + reshape + concatenation + softmax / tanh in one
    kernel per level) against the torch form on the same weights and pyramid: box regressions bit for bit, class scores to float32
    rounding of the softmax sum, coefficients within 2 ulp -- of the same convolution output.  The two forms do not run the same convolution
    kernels: the fused form convolves with the merged 384-channel layer (Winograd F(2x4) at the 69 x 69 and 35 x 35 levels of this
    two-frame batch since the round-5 small-launch rule), the torch form with the three separate layers (12 / 243 / 96 channels: library
    kernels), so the comparison carries the Winograd kernel's float32 bound (5e-7 of the sum of |terms|; 3.2e-6 measured on these layers
    against float64) on top; the kernel alone is checked bit for bit below."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    torch.manual_seed(16)
    cl = torch.channels_last
    head = net_mod.SharedHead().cuda().eval().to(memory_format=cl)
    head.merge_output_layers()
    head.merged.to(memory_format=cl)
    sizes = ((69, 69), (35, 35), (18, 18), (9, 9), (5, 5))
    pyramid = [torch.randn(2, 256, h, w, device="cuda").contiguous(memory_format=cl) for h, w in sizes]
    P = sum(h * w * 3 for h, w in sizes)
    with torch.no_grad():
        loc, conf, coef = head.fused_outputs(pyramid, P)
        monkeypatch.setenv("AMOS_MASK_FUSED_HEAD", "0")
        assert head.fused_outputs(pyramid, P) is None
        locs, confs, coefs = zip(*(head(p) for p in pyramid))
        want_loc, want_conf, want_coef = torch.cat(locs, 1), torch.softmax(torch.cat(confs, 1), -1), torch.cat(coefs, 1)
    assert loc.shape == want_loc.shape == (2, P, 4) and conf.shape == (2, P, 81) and coef.shape == (2, P, 32)
    assert torch.allclose(loc, want_loc, rtol=1e-5, atol=1e-5) and torch.allclose(coef, want_coef, rtol=1e-5, atol=1e-5)
    assert torch.allclose(conf, want_conf, rtol=5e-5, atol=1e-8) and torch.allclose(conf.sum(-1), torch.ones_like(conf[..., 0]), atol=1e-5)
    # the kernel alone on one raw tensor: exact for box / coefficient channels
    raw = torch.randn(2, 352, 7, 5, device="cuda").contiguous(memory_format=cl)
    bias = torch.randn(352, device="cuda")
    l2, c2, m2 = torch.zeros(2, 200, 4, device="cuda"), torch.zeros(2, 200, 81, device="cuda"), torch.zeros(2, 200, 32, device="cuda")
    gpu_lib.mask_head_outputs(torch.cuda.current_stream().cuda_stream, raw.data_ptr(), bias.data_ptr(), l2.data_ptr(), c2.data_ptr(), m2.data_ptr(), 2, 35, 352, 3,
                              81, 32, 200, 50)
    torch.cuda.synchronize()
    y = (raw + bias.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert torch.equal(l2[:, 50:155], y[..., :12].reshape(2, -1, 4)) and torch.equal(m2[:, 50:155], torch.tanh(y[..., 255:351].reshape(2, -1, 32)))
    assert torch.allclose(c2[:, 50:155], torch.softmax(y[..., 12:255].reshape(2, -1, 81), -1), rtol=1e-6, atol=1e-9)
    assert float(l2[:, :50].abs().max()) == 0 and float(c2[:, 155:].abs().max()) == 0  # nothing outside the level's priors
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.mask_head_outputs(0, raw.data_ptr(), bias.data_ptr(), l2.data_ptr(), c2.data_ptr(), m2.data_ptr(), 2, 35, 352, 3, 81, 32, 200, 150)


@pytest.mark.gpu
def test_head_kernel_writes_the_class_scores_of_detect(mask, gpu_lib, monkeypatch):
    """amos_mask_head_outputs_scores_device: the prediction head's output kernel also writes Detect's class scores [B][80][P] (background dropped,
    -1 under the confidence threshold) -- bit for bit what amos_mask_class_scores_device makes of the softmax tensor the same kernel writes; with
    the softmax output left out the other outputs do not change.  Then the engine: a pass run for the detector with scores only
    (AMOS_MASK_HEAD_SCORES=1) gives the masks of the pass through the softmax tensor (the default), eager at 1 and 5 frames."""
    torch.manual_seed(23)
    st = torch.cuda.current_stream().cuda_stream
    cl = torch.channels_last
    for b, h, w, P, off in ((2, 7, 5, 200, 50), (1, 69, 69, 19248, 0), (3, 9, 9, 243, 0)):
        raw = (torch.randn(b, 384, h, w, device="cuda") * 3).contiguous(memory_format=cl)
        raw[:, :, 0, :] *= 0.01   # flat class distributions in the first row of cells: nothing above the threshold there
        bias = torch.randn(384, device="cuda") * 0.01
        cells = h * w
        outs = {}
        for want_conf, want_scores in ((True, True), (False, True), (True, False)):
            loc, conf, coef = torch.zeros(b, P, 4, device="cuda"), torch.zeros(b, P, 81, device="cuda"), torch.zeros(b, P, 32, device="cuda")
            cls = torch.full((b, 80, P), 7.0, device="cuda")
            gpu_lib.mask_head_outputs_scores(st, raw.data_ptr(), bias.data_ptr(), loc.data_ptr(), conf.data_ptr() if want_conf else None, coef.data_ptr(),
                                             cls.data_ptr() if want_scores else None, 0.05, b, cells, 384, 3, 81, 32, P, off)
            torch.cuda.synchronize()
            outs[(want_conf, want_scores)] = (loc, conf, coef, cls)
        loc, conf, coef, cls = outs[(True, True)]
        plain = outs[(True, False)]
        assert torch.equal(loc, plain[0]) and torch.equal(conf, plain[1]) and torch.equal(coef, plain[2])  # the plain form's outputs
        only = outs[(False, True)]
        assert torch.equal(only[0], loc) and torch.equal(only[2], coef) and torch.equal(only[3], cls) and float(only[1].abs().max()) == 0
        n = cells * 3
        want = torch.full((b, 80, n), float("nan"), device="cuda")
        gpu_lib.mask_class_scores(st, conf[:, off:off + n].contiguous().data_ptr(), want.data_ptr(), b, n, 81, 0.05)
        torch.cuda.synchronize()
        assert torch.equal(cls[:, :, off:off + n], want) and bool((want == -1).any()) and bool((want > 0.05).any())
        if off:
            assert bool((cls[:, :, :off] == 7.0).all()) and bool((cls[:, :, off + n:] == 7.0).all())  # nothing outside the level's priors
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.mask_head_outputs_scores(st, raw.data_ptr(), bias.data_ptr(), loc.data_ptr(), None, coef.data_ptr(), None, 0.05, b, cells, 384, 3, 81, 32, P, off)
    eng = _engine(mask, "cuda:0", "seed0").prepare()
    frames = torch.from_numpy(np.stack([mask_cases.frame(c) for c in ("seed0", "ref122_w0", "tum_w0", "seed0", "ref122_w0")])).cuda()
    frames[3:] = frames[3:].flip(2)
    seen = []
    real = eng.net.forward
    monkeypatch.setattr(eng.net, "forward", lambda x, scores_only=False: (seen.append(scores_only), real(x, scores_only))[1])
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AMOS_MASK_HEAD_SCORES", mode)
        del seen[:]
        got[mode] = (eng.eval_bgr_batch(frames[:1], chunk=1).clone(), eng.eval_bgr_batch(frames, chunk=5).clone())
        assert seen == [mode == "1"] * 2
    assert torch.equal(got["1"][0], got["0"][0]) and torch.equal(got["1"][1], got["0"][1]) and int((got["1"][1] > 0).sum()) > 0
    with torch.no_grad():
        pred = eng._forward(eng._preprocess_hip(frames[:2]), True)
    assert "conf" not in pred and pred["cls"].shape == (2, 80, 19248)


@pytest.mark.gpu
def test_topk_rows_kernel_against_torch_topk(mask, gpu_lib):
    """amos_mask_topk_rows_device against torch.topk(sorted=True): the values bit for bit; the indices point at those values, are unique
    per row, and equal torch's wherever the row's top values are distinct.  Rows of distinct random values, rows that are mostly one
    value (-1 for priors under the threshold) with a handful / exactly k / more than k real scores, ties across the k-th place (lowest
    index first), negative values, k = 1, k = n, n not a multiple of 256."""
    torch.manual_seed(17)
    st = torch.cuda.current_stream().cuda_stream

    def run(x, k):
        rows, n = x.shape
        v = torch.full((rows, k), float("nan"), device="cuda")
        i = torch.full((rows, k), -1, dtype=torch.int64, device="cuda")
        gpu_lib.mask_topk_rows(st, x.data_ptr(), v.data_ptr(), i.data_ptr(), rows, n, k)
        torch.cuda.synchronize()
        return v, i

    # (launches of at most 256 rows of at most 32 768 values take the LDS-resident kernel, the others the five-scan kernel: both here)
    for rows, n, k in ((64, 19248, 200), (5, 1000, 200), (3, 257, 256), (4, 50, 1), (2, 37, 37), (3, 40000, 200), (2, 32768, 256), (2, 32769, 100), (3, 19250, 200), (300, 19248, 200)):
        x = torch.randn(rows, n, device="cuda")
        v, i = run(x, k)
        wv, wi = x.topk(k, dim=1)
        assert torch.equal(v, wv) and torch.equal(i, wi), (rows, n, k)
    # the detector's rows: -1 everywhere except a few real scores
    for real in (0, 7, 200, 1500):
        x = torch.full((6, 19248), -1.0, device="cuda")
        if real:
            pos = torch.stack([torch.randperm(19248, device="cuda")[:real] for _ in range(6)])
            x.scatter_(1, pos, torch.rand(6, real, device="cuda") * 0.9 + 0.05)
        v, i = run(x, 200)
        wv, _ = x.topk(200, dim=1)
        assert torch.equal(v, wv), real
        assert torch.equal(torch.gather(x, 1, i), v) and all(len(set(r.tolist())) == 200 for r in i.cpu()), real
        if real < 200:  # the padding: the lowest indices among the -1 entries
            tail = i[0, real:].cpu().tolist()
            free = [j for j in range(19248) if x[0, j].item() == -1.0][:200 - real]
            assert tail == free
    # ties across the k-th place: lowest index first
    x = torch.zeros(1, 600, device="cuda")
    x[0, 100:110] = 2.0
    x[0, 300:400] = 1.0
    v, i = run(x, 60)
    assert v[0].tolist() == [2.0] * 10 + [1.0] * 50 and i[0].tolist() == list(range(100, 110)) + list(range(300, 350))
    with pytest.raises(gpu_lib.AmosError):
        run(x, 257)

    # amos_mask_topk_rows_sparse_device (one scan for rows that are mostly `fill`): ALWAYS the generic kernel's result, values and indices --
    # rows that fit its assumption (0 / 7 / 199 / 200 / 1 024 live scores), rows that do not (1 025, 1 500 and 9 000 live scores: the list
    # overflows; values BELOW the fill value with fewer than k live ones; no fill value at all), ties among the live scores, short rows
    def run_sparse(x, k, fill):
        rows, n = x.shape
        v = torch.full((rows, k), float("nan"), device="cuda")
        i = torch.full((rows, k), -1, dtype=torch.int64, device="cuda")
        gpu_lib.mask_topk_rows_sparse(st, x.data_ptr(), v.data_ptr(), i.data_ptr(), rows, n, k, fill)
        torch.cuda.synchronize()
        return v, i

    for real in (0, 7, 199, 200, 1024, 1025, 1500, 9000):
        x = torch.full((5, 19248), -1.0, device="cuda")
        if real:
            pos = torch.stack([torch.randperm(19248, device="cuda")[:real] for _ in range(5)])
            x.scatter_(1, pos, torch.rand(5, real, device="cuda") * 0.9 + 0.05)
            x[1, pos[1, :min(real, 40)]] = 0.5   # ties among the live scores
        if real == 7:
            x[2, 5000:5100] = -3.0   # values below the fill value while the list is short of k: the generic path
        for k in (200, 1, 256):
            gv, gi = run(x, k)
            sv, si = run_sparse(x, k, -1.0)
            assert torch.equal(sv, gv) and torch.equal(si, gi), (real, k)
    for n in (40000, 32768):   # the same on both sides of the LDS kernel's row limit
        x = torch.full((3, n), -1.0, device="cuda")
        pos = torch.stack([torch.randperm(n, device="cuda")[:150] for _ in range(3)])
        x.scatter_(1, pos, torch.rand(3, 150, device="cuda") * 0.9 + 0.05)
        x[2, 100:300] = -2.0
        gv, gi = run(x, 200)
        sv, si = run_sparse(x, 200, -1.0)
        wv, _ = x.topk(200, dim=1)
        assert torch.equal(sv, gv) and torch.equal(si, gi) and torch.equal(gv, wv), n
        assert gi[0, 150:].tolist() == [j for j in range(n) if x[0, j].item() == -1.0][:50]
    x = torch.randn(4, 3001, device="cuda")   # no fill value anywhere, a row length that is not a multiple of 4
    for fill in (-1.0, 10.0, -10.0):
        gv, gi = run(x, 100)
        sv, si = run_sparse(x, 100, fill)
        assert torch.equal(sv, gv) and torch.equal(si, gi), fill


@pytest.mark.gpu
def test_fused_post_processing_equals_the_torch_ops_path(mask, gpu_lib, monkeypatch):
    """amos_mask_person_masks_device (Detect + postprocess + prep_display in seven launches) against the torch-op chain it replaces
    (detect_batch + person_mask_batch, themselves held to the reference-order path by test_static_shape_batch_post_equals_reference_order_post)
    on real network outputs -- the golden frames of weight set seed0, plain and mirrored -- and on edited outputs: nothing above the score
    threshold (found False, zero mask), a class list with equal scores, more displayed candidates than 15.  Same `found`, same detections,
    masks within a handful of boundary pixels (the 32-term prototype sums are rounded in another order)."""
    post = importlib.import_module("amos_slam_amd.mask.post")
    det_mod = importlib.import_module("amos_slam_amd.mask.detect")
    eng = _engine(mask, "cuda:0", "seed0").prepare()
    frames = np.stack([mask_cases.frame(c) for c in ("seed0", "ref122_w0", "tum_w0")] * 2)
    frames[3:] = frames[3:, :, ::-1]
    x = eng._preprocess_hip(torch.from_numpy(np.ascontiguousarray(frames)).cuda())
    with torch.no_grad():
        pred = eng._forward(x)

    def both(p):
        fused = post.person_masks_fused(p, 640, 480)
        assert fused is not None
        want = post.person_mask_batch(det_mod.detect_batch(p), 640, 480)
        torch.cuda.synchronize()
        return fused, want

    (m, f), (mw, fw) = both(pred)
    assert m.shape == mw.shape == (6, 480, 640) and m.dtype == torch.uint8 and torch.equal(f, fw) and bool(f.all())
    assert int((mw > 0).sum()) > 6 * 5000
    for k in range(6):
        assert int((m[k] != mw[k]).sum()) <= 40 and _iou((m[k] > 0).cpu().numpy(), (mw[k] > 0).cpu().numpy()) >= 1 - 1e-3, k
    # a batch whose class scores are all below the threshold
    quiet = dict(pred)
    quiet["conf"] = torch.zeros_like(pred["conf"])
    quiet["conf"][..., 0] = 1.0
    (m, f), (mw, fw) = both(quiet)
    assert not bool(f.any()) and not bool(fw.any()) and int(m.sum()) == 0 and int(mw.sum()) == 0
    # equal scores inside a class list and 40 strong person candidates on a grid (more than the 15 displayed): same selection both ways
    tied = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in pred.items()}
    tied["conf"].zero_()
    tied["conf"][..., 0] = 1.0
    pick = torch.arange(40, device="cuda") * 431 + 7
    tied["conf"][:, pick, 1] = 0.9
    tied["conf"][:, pick, 0] = 0.1
    tied["conf"][:, pick[:20], 1] = torch.linspace(0.99, 0.91, 20, device="cuda")   # twenty distinct leaders (the displayed 15 among them), then 20 tied at 0.9
    (m, f), (mw, fw) = both(tied)                                                    # (torch.topk leaves the order of EQUAL scores open: the ties sit behind the displayed ones)
    assert bool(f.all()) and torch.equal(f, fw)
    for k in range(6):
        assert _iou((m[k] > 0).cpu().numpy(), (mw[k] > 0).cpu().numpy()) >= 1 - 1e-3, k


@pytest.mark.gpu
def test_mask_pass_as_one_hip_graph_equals_the_eager_pass(mask, gpu_lib):
    """MaskEngine.capture_graph / eval_bgr_graph: network + detection + mask assembly of a fixed batch replayed as one HIP graph give the
    masks of the eager pass on different frames (two replays), and refuse another batch size."""
    torch.manual_seed(18)
    eng = mask.MaskEngine(device="cuda:0", seed=3).prepare()
    rng = np.random.default_rng(5)
    eng.capture_graph(batch=2)
    for _ in range(2):
        frames = torch.as_tensor(rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8), device="cuda:0")
        want = eng.eval_bgr_batch(frames, chunk=2)
        masks, found = eng.eval_bgr_graph(frames)
        torch.cuda.synchronize()
        assert masks.shape == (2, 480, 640) and masks.dtype == torch.uint8 and found.shape == (2,)
        assert torch.equal(torch.where(found[:, None, None], masks, torch.zeros_like(masks)), want)
    with pytest.raises(RuntimeError):
        eng.eval_bgr_graph(frames[:1])


@pytest.mark.gpu
def test_channel_blocked_chain_gives_the_channels_last_pass(mask, gpu_lib, monkeypatch):
    """Level 0 of the pyramid channel-blocked ([b][c / 8][h][w][8]) through the prototype network and its prediction head (passes of at least 8
    frames where every layer on the way is an F(2 x 4) launch: net.blocked_chain_for) against AMOS_MASK_BLOCKED_CHAIN=0: the same kernels' other
    layout, the same arithmetic -- the same bits in every network output.  Seven layout launches per pass (pyramid output, three + one
    prototype layers, the head's two), none at 4 frames (the small-launch form keeps channels-last)."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    eng = _engine(mask, "cuda:0", "seed0").prepare()
    frames = torch.from_numpy(np.stack([mask_cases.frame(c) for c in ("seed0", "ref122_w0", "tum_w0", "blobs7_w1")] * 2)).cuda()
    frames[4:] = frames[4:].flip(2)
    calls = []
    real = gpu_lib.mask_winograd24_conv_layout
    monkeypatch.setattr(gpu_lib, "mask_winograd24_conv_layout", lambda *a: (calls.append((a[7], a[8], a[12], a[13])), real(*a))[1])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AMOS_MASK_BLOCKED_CHAIN", mode)
        del calls[:]
        with torch.no_grad():
            pred = eng._forward(eng._preprocess_hip(frames))
        torch.cuda.synchronize()
        out[mode] = {k: pred[k].clone() for k in ("loc", "conf", "mask", "proto")}
        if mode == "1":
            assert sorted(calls) == sorted([(69, 69, False, True), (69, 69, True, True), (69, 69, True, True), (69, 69, True, True), (138, 138, True, False),
                                            (69, 69, True, True), (69, 69, True, False)]), calls
        else:
            assert not calls
    # (the passes share no bits by themselves: at 8 frames the lateral layers are library split-k kernels that sum with atomics)
    for k in out["1"]:
        assert torch.allclose(out["1"][k], out["0"][k], rtol=1e-4, atol=1e-5), k
    # the chain's layers on ONE fixed input, both layouts: bit for bit
    net = eng.net
    pn, head = net.proto_net, net.prediction_layers[0]
    torch.manual_seed(24)
    m0 = torch.randn(8, 256, 69, 69, device="cuda").contiguous(memory_format=torch.channels_last)
    res = {}
    with torch.no_grad():
        for blk in (False, True):
            p3 = net_mod.conv_bias_act(net.fpn.pred_layers[2], m0, True, out_blocked=blk)
            assert isinstance(p3, net_mod.Blocked) == blk and tuple(p3.shape) == (8, 256, 69, 69)
            p = p3
            for j in (0, 2, 4):
                p = net_mod.conv_bias_act(pn[j], p, True, out_blocked=blk)
            p = net_mod.bilinear(p, scale_factor=2, relu=True)
            assert isinstance(p, net_mod.Blocked) == blk and tuple(p.shape) == (8, 256, 138, 138)
            p = net_mod.conv_bias_act(pn[8], p, True)
            u = net_mod.conv_bias_act(head.upfeature[0], p3, True, out_blocked=blk)
            raw = net_mod.conv_raw(head.merged, u)
            res[blk] = (p, raw)
    torch.cuda.synchronize()
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert res[True][0].is_contiguous(memory_format=torch.channels_last) and res[True][1].shape == (8, 384, 69, 69)
    monkeypatch.setenv("AMOS_MASK_BLOCKED_CHAIN", "1")
    del calls[:]
    with torch.no_grad():
        eng._forward(eng._preprocess_hip(frames[:4]))
    assert not calls


@pytest.mark.gpu
def test_side_stream_branches_give_the_one_stream_pass(mask, gpu_lib, monkeypatch):
    """A small pass with the pyramid's side levels and the prediction head on side streams (net.Branches; the default up to 16 frames per
    pass) against the one-stream pass (AMOS_MASK_BRANCHES=0): the same kernels on the same operands, so the same network outputs (to float32
    rounding: the library's split-k kernels sum with atomics) and the same person masks -- eager, with one and two frames, and inside the
    one-frame HIP graph of MaskEngine.frame_session, replayed on several frames.  A pass above the frame limit takes no side stream."""
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    eng = _engine(mask, "cuda:0", "seed0")
    eng.prepare()
    frames = torch.from_numpy(np.stack([mask_cases.frame(c) for c in ("seed0", "ref122_w0", "tum_w0")])).cuda()
    made = []
    real = net_mod.Branches
    monkeypatch.setattr(net_mod, "Branches", lambda *a, **k: (made.append(1), real(*a, **k))[1])
    for n in (1, 2):
        out = {}
        for mode in ("0", "auto"):
            monkeypatch.setenv("AMOS_MASK_BRANCHES", mode)
            del made[:]
            x = eng._preprocess_hip(frames[:n])
            with torch.no_grad():
                pred = eng._forward(x)
            torch.cuda.synchronize()
            assert len(made) == (0 if mode == "0" else 1)
            out[mode] = ({k: pred[k].clone() for k in ("loc", "conf", "mask", "proto")}, eng.eval_net_input_batch(x, chunk=n).clone())
        for k in ("loc", "conf", "mask", "proto"):
            assert out["0"][0][k].shape == out["auto"][0][k].shape
            assert torch.allclose(out["0"][0][k], out["auto"][0][k], rtol=1e-4, atol=1e-5), (n, k, float((out["0"][0][k] - out["auto"][0][k]).abs().max()))
        assert torch.equal(out["0"][1], out["auto"][1]), n
        assert int((out["auto"][1] > 0).sum()) > 0
    # above the limit: one stream
    monkeypatch.setenv("AMOS_MASK_BRANCHES", "auto")
    monkeypatch.setenv("AMOS_MASK_BRANCH_MAX_BATCH", "2")
    del made[:]
    with torch.no_grad():
        eng._forward(eng._preprocess_hip(frames))
    assert not made
    monkeypatch.delenv("AMOS_MASK_BRANCH_MAX_BATCH")
    # the one-frame session: a graph with parallel branches, replayed on three frames, against the eager one-stream masks
    monkeypatch.setenv("AMOS_MASK_BRANCHES", "0")
    want = eng.eval_bgr_batch(frames, chunk=1).cpu().numpy()
    monkeypatch.setenv("AMOS_MASK_BRANCHES", "auto")
    del made[:]
    s = eng.frame_session(480, 640)
    assert made, "the session's capture took no side stream"
    for rep in range(2):
        for k in range(3):
            s.frame_in.numpy()[...] = frames[k].cpu().numpy()
            assert s.run()
            assert np.array_equal(s.mask_out.numpy(), want[k]), (rep, k)


def _bench_frames_per_forward():
    """Frames per network forward of the headline run: bench.py CONFIGS["c3"] (default_batch frames per step over default_streams lanes)."""
    keep = os.environ.get("GPU_MAX_HW_QUEUES")   # bench.py sets its own default on import: not this process's business
    import bench
    if keep is None:
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
    c3 = bench.CONFIGS["c3"]
    return c3["default_batch"] // c3["default_streams"]


# weight sets of the golden cases: the cases of one set share the weight seed AND the class-head bias tweak, so one engine serves them
WEIGHT_SETS = {"seed0": ("seed0", "ref122_w0", "tum_w0"), "blobs7_w1": ("blobs7_w1", "tum_w1"), "ref122_w3": ("ref122_w3",)}


def _bench_sized_batch(fpf, cases):
    """fpf frames: the golden cases of one weight set first, in order and unmirrored, then the four frame sources repeated; the second
    half mirrored (different inputs), EXCEPT the last len(cases) slots, which hold the golden frames again (the launch's last frames)."""
    names = list(cases) + [("seed0", "ref122_w0", "tum_w0", "blobs7_w1")[k % 4] for k in range(fpf - len(cases))]
    frames = np.stack([mask_cases.frame(c) for c in names])
    frames[fpf // 2:] = frames[fpf // 2:, :, ::-1]
    tail = list(range(fpf - len(cases), fpf))
    for slot, c in zip(tail, cases):
        frames[slot] = mask_cases.frame(c)
    return torch.from_numpy(np.ascontiguousarray(frames)).cuda(), tail


@pytest.mark.gpu
@pytest.mark.parametrize("fpf", sorted({32, _bench_frames_per_forward()}))
def test_bench_sized_pass_winograd_against_direct_kernels(mask, gpu_lib, monkeypatch, fpf):
    """The mask pass at the bench's launch sizes -- 32 frames per forward and the headline's own count, read from bench.CONFIGS["c3"]
    (64: a 1.25 GB Winograd input, the 18 x 18 layers past the 256-work-group rule, every index of the resize / top-k / mask kernels at
    twice the extent) -- against the same pass with the Winograd kernel switched off: the same detections and person masks, network
    outputs within float32 rounding of each other; and the golden cases riding in the first and the last slots of the launch give the
    reference's person masks (same weights => the masks of batch 1)."""
    cases = WEIGHT_SETS["seed0"]
    frames, tail = _bench_sized_batch(fpf, cases)
    eng = _engine(mask, "cuda:0", "seed0")
    eng.prepare()
    calls = []
    _spy_winograd(gpu_lib, monkeypatch, calls, 6, 11)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AMOS_MASK_WINOGRAD", mode)
        x = eng._preprocess_hip(frames)
        with torch.no_grad():
            pred = eng._forward(x)
        out[mode] = ({k: pred[k] for k in ("loc", "conf", "mask", "proto")}, eng.eval_net_input_batch(x, chunk=fpf))
        del pred, x
        torch.cuda.synchronize()
    # per forward: 13 bottleneck conv2, the FPN prediction layers and the head's two convolutions at 69 x 69 and 35 x 35, 4 protonet layers,
    # and the 18 x 18 layers (round 5: launches of fewer than 256 work-groups go by the rule's stage-latency estimate).  Two forwards in this
    # mode (_forward and eval_net_input_batch)
    base = 2 * (13 + 2 + 4 + 4)
    assert all(c[0] == fpf for c in calls), sorted({c[0] for c in calls})
    assert len(calls) > base and len(calls) % 2 == 0, len(calls)
    # (the 18 x 18 layers: 164 work-groups at 32 frames, 324 at 64 -- below / above one per CU; both sides of the rule's small-launch clause)
    assert any(c[1] == 18 and c[2] == 18 for c in calls), "the 18 x 18 layers are expected on the Winograd kernel at this launch size"
    for k in ("loc", "conf", "mask", "proto"):
        a, b = out["1"][0][k], out["0"][0][k]
        assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1.0), k
    m1, m0 = out["1"][1] > 0, out["0"][1] > 0
    assert int(m0.sum()) > fpf * 5000
    for f in range(fpf):
        assert _iou(m1[f].cpu().numpy(), m0[f].cpu().numpy()) >= 1 - 1e-3, f
    for slots in (range(len(cases)), tail):
        for slot, c in zip(slots, cases):
            want = np.unpackbits(GOLD[c]["person_mask_bits"])[:480 * 640].reshape(480, 640).astype(bool)
            assert _iou(m1[slot].cpu().numpy(), want) >= 1 - 1e-3, (slot, c)


@pytest.mark.gpu
@pytest.mark.parametrize("weights", ["blobs7_w1", "ref122_w3"])
def test_bench_sized_pass_gives_the_golden_masks_of_the_other_weight_sets(mask, gpu_lib, weights):
    """The other two weight sets of the golden cases (seeds 1 and 3; the latter is the cars-only case: an EMPTY person mask) at the
    headline's frames per forward, through the batch path bench.py runs (eval_net_input_batch on the HIP pre-processing): the
    reference's person masks in the first and the last slots of the launch."""
    fpf = _bench_frames_per_forward()
    cases = WEIGHT_SETS[weights]
    frames, tail = _bench_sized_batch(fpf, cases)
    eng = _engine(mask, "cuda:0", weights).prepare()
    masks = eng.eval_net_input_batch(eng._preprocess_hip(frames), chunk=fpf) > 0
    torch.cuda.synchronize()
    assert masks.shape == (fpf, 480, 640)
    for slots in (range(len(cases)), tail):
        for slot, c in zip(slots, cases):
            want = np.unpackbits(GOLD[c]["person_mask_bits"])[:480 * 640].reshape(480, 640).astype(bool)
            assert _iou(masks[slot].cpu().numpy(), want) >= 1 - 1e-3, (slot, c)
