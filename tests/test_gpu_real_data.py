"""GPU parity on the reference's OWN data files (tests/golden/ref_data/: the two RGB frames of
src/python/input/ -- 1341846313.553992.png is named like a TUM fr3 frame -- and the two person masks of
src/python/output/mask/, copied as data) and the BASELINE configs[2] chain end to end.

This does not pin the OpenCV-derived stages (nothing in this image can: parity unpinned): it
checks the HIP path against the oracle on real texture and real mask shapes instead of synthetic rectangles."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "ref_data")
FRAMES = ("1341846313.553992", "122_rgb")


def _load(name):
    from PIL import Image
    rgb = np.array(Image.open(os.path.join(DATA, name + ".png")).convert("RGB"))
    mask = np.array(Image.open(os.path.join(DATA, name + "_person_mask.png")).convert("L"))
    assert rgb.shape == (480, 640, 3) and mask.shape == (480, 640)
    return np.ascontiguousarray(rgb), np.ascontiguousarray(mask)


def _same(a, b, what):
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert a.tobytes() == b.tobytes(), f"{what} differ"


@pytest.mark.parametrize("name", FRAMES)
def test_real_frame_every_stage(gpu_lib, ob, name):
    """Tracking::GrabImageRGBD's cvtColor(RGB2GRAY) + the 4-argument operator() on a real frame, stage by stage."""
    rgb, _ = _load(name)
    gray = ob.color_to_gray(rgb, rgb_order=True)
    ext, orc = gpu_lib.OrbExtractor(), ob.Oracle()
    ext.detect(gray)
    orc.detect(gray)
    for l in range(8):
        _same(ext.level_image(l, padded=True), orc.level_image(l, padded=True), f"{name} padded level {l}")
        _same(ext.level_candidates(l), orc.level_candidates(l), f"{name} FAST candidates level {l}")
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"{name} keypoints level {l}")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    for l in range(8):
        _same(ext.blurred_image(l), orc.blurred_image(l), f"{name} blurred level {l}")
    _same(kg, ko, f"{name} keypoints")
    _same(dg, do, f"{name} descriptors")
    assert len(kg) > 900  # real indoor texture fills the 1000-feature budget


def test_real_frames_colour_batch_and_match(gpu_lib, ob):
    """Both real frames as one resident RGB batch (gray conversion fused into the import) and the N x N best-2 match
    between them, against the oracle."""
    import torch
    rgbs = np.stack([_load(n)[0] for n in FRAMES])
    d = torch.from_numpy(rgbs).cuda()
    ext = gpu_lib.OrbExtractor(max_batch=2)
    ext.extract_batch_device_color(d.data_ptr(), 480 * 640 * 3, 640 * 3, 640, 480, 2, channels=3, rgb_order=True)
    ext.sync()
    descs = []
    for f in range(2):
        ko, do = ob.Oracle().extract(ob.color_to_gray(rgbs[f], rgb_order=True))
        kg, dg = ext.batch_fetch(f)
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")
        descs.append(dg)
    m = gpu_lib.OrbMatcher()
    for a, b in ((0, 1), (1, 0), (0, 0)):
        _same(m.bruteforce_best2(descs[a], descs[b]), ob.bruteforce_best2(descs[a], descs[b]), f"match {a} vs {b}")
    self_match = m.bruteforce_best2(descs[0], descs[0])
    assert (self_match["best_dist"] == 0).all()


@pytest.mark.parametrize("name", FRAMES)
def test_real_person_mask_gate(gpu_lib, ob, name):
    """ORBextractor::MovingKeyPoints with the reference's own person mask of that frame: closing, removed list,
    kept lists, descriptors."""
    rgb, mask = _load(name)
    assert set(np.unique(mask).tolist()) == {0, 255} and 0.02 < (mask > 0).mean() < 0.6
    gray = ob.color_to_gray(rgb, rgb_order=True)
    ext, orc = gpu_lib.OrbExtractor(), ob.Oracle()
    ext.detect(gray)
    orc.detect(gray)
    rg, ro = ext.gate(mask), orc.gate(mask)
    _same(ext.closed_mask(), orc.closed_mask(), "closed mask")
    _same(rg, ro, "removed keypoints")
    for l in range(8):
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"kept keypoints level {l}")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    _same(kg, ko, "keypoints after the gate")
    _same(dg, do, "descriptors after the gate")
    assert len(rg) > 20, "the person carries keypoints in both frames"
    closed = ext.closed_mask()
    xs, ys = kg["x"].astype(int).clip(0, 639), kg["y"].astype(int).clip(0, 479)  # (int)search_coord, ORBextractor.cc:1721-1730
    assert (closed[ys, xs] != 0).mean() < 0.01  # next to nothing on the (closed) person survives


def _biased_engine(mask_mod, device):
    import torch
    eng = mask_mod.MaskEngine(device=device, seed=0)
    with torch.no_grad():  # random weights with a class head biased towards "person": detections to post-process
        head = eng.net.prediction_layers[0].conf_layer.bias
        b = head.detach().cpu().view(3, 81).clone()
        b[:, 1] += 5.0
        b[1, 3] += 5.5
        head.copy_(b.view(-1).to(head.device))
    return eng.prepare()


@pytest.mark.parametrize("fused_import", [True, False])
def test_configs2_chain_end_to_end(gpu_lib, ob, synth, pkg, fused_import):
    """BASELINE configs[2] exactly as bench.py issues it: the colour frames read once into the gray pyramid and the mask
    network's input (fused_import; or the separate gray detect + three-kernel mask pre-processing), the rest of detect on
    the lane's stream, the mask network on the SAME stream through torch, gate with the network's device-resident masks,
    describe, N x N best-2 match of frame k against frame k-1 -- compared frame by frame with the oracle's detect ->
    gate(the same masks, copied to the host) -> describe and orc bruteforce_best2.  Covers the torch-stream <->
    handle-stream hand-off and the device mask pointer."""
    import torch
    mask_mod = importlib.import_module("amos_slam_amd.mask")
    n = 6
    grays = np.concatenate([np.stack([ob.color_to_gray(_load(nm)[0], rgb_order=True) for nm in FRAMES]), synth.frames(5, 0, n - 2)])
    d_frames = torch.from_numpy(grays).cuda()
    bgr = d_frames.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
    bgr[0] = torch.from_numpy(np.ascontiguousarray(_load(FRAMES[0])[0][:, :, ::-1])).cuda()  # real colour for the real frames
    bgr[1] = torch.from_numpy(np.ascontiguousarray(_load(FRAMES[1])[0][:, :, ::-1])).cuda()
    eng = _biased_engine(mask_mod, "cuda:0")
    ext = pkg.OrbExtractor(max_batch=n)
    matcher = pkg.OrbMatcher(stream=ext.stream)
    _, d_desc, d_counts, cap = ext.batch_results_device()
    lane = torch.cuda.ExternalStream(ext.stream, device=0)
    pairs_q = torch.arange(n, dtype=torch.int32, device="cuda")
    pairs_t = (pairs_q - 1) % n
    d_match = torch.full((n, cap, 4), 1 << 30, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    pre = pkg.MaskPreprocessor(640, 480, n, stream=ext.stream)
    net_in = torch.empty((n, 3, 550, 550), dtype=torch.float32, device="cuda")
    for _ in range(2):  # twice: the second pass overwrites a used handle, as every bench step after the first does
        if fused_import:
            ext.detect_color_with_mask_pre_batch_device(pre, bgr.data_ptr(), 480 * 640 * 3, 640 * 3, 640, 480, n, net_in.data_ptr())
        else:
            ext.detect_batch_device(d_frames.data_ptr(), 480 * 640, 640, 640, 480, n)
        with torch.cuda.stream(lane):
            masks = eng.eval_net_input_batch(net_in, chunk=4) if fused_import else eng.eval_bgr_batch(bgr, chunk=4)
            ext.gate_batch_device(masks.data_ptr(), 480 * 640, 640)
            ext.describe_batch_device()
        matcher.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pairs_q.data_ptr(), pairs_t.data_ptr(), n, cap, 256, d_match.data_ptr())
    ext.sync()
    torch.cuda.synchronize()
    masks_h = masks.cpu().numpy()
    assert masks_h.shape == (n, 480, 640) and masks_h.dtype == np.uint8
    assert sum(int(m.any()) for m in masks_h) >= n - 1, "the biased head finds 'persons' in (nearly) every frame"
    got_match = d_match.cpu().numpy()
    descs, removed_any = [], 0
    for f in range(n):
        orc = ob.Oracle()
        orc.detect(grays[f])
        n_before = sum(len(orc.level_keypoints(l)) for l in range(8))
        orc.gate(masks_h[f])
        ko, do = orc.describe()
        removed_any += int(len(ko) < n_before)
        kg, dg = ext.batch_fetch(f)
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")
        descs.append(do)
    assert removed_any >= 2, "the gate removed keypoints in several frames"
    for f in range(n):
        want = ob.bruteforce_best2(descs[f], descs[(f - 1) % n])
        got = got_match[f, :len(descs[f])].view(pkg.BEST2_DTYPE).reshape(-1)
        _same(got, want, f"match of frame {f} against frame {(f - 1) % n}")


@pytest.mark.parametrize("w,h", [(640, 480), (413, 307)])
def test_fused_colour_import_feeds_pyramid_and_mask_network(gpu_lib, ob, w, h):
    """amos_orb_detect_color_with_mask_pre_batch_device reads the colour frames once; the padded gray level 0
    and everything detect derives from it equal the oracle's cvtColor + detect, and the mask network's input tensor equals
    amos_mask_preprocess_batch_device's bit for bit."""
    import torch
    rng = np.random.default_rng(w)
    n = 3
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if (w, h) == (640, 480):
        frames[0] = _load(FRAMES[0])[0][:, :, ::-1]  # a real frame, BGR
    else:  # smooth structure so that FAST finds corners at the small size
        yy, xx = np.mgrid[0:h, 0:w]
        frames[0] = ((90 * ((xx // 16 + yy // 12) % 2))[..., None] + rng.integers(0, 40, (h, w, 3))).astype(np.uint8)
    d = torch.from_numpy(frames).cuda()
    nl = 8 if w == 640 else 5
    ext = gpu_lib.OrbExtractor(n_levels=nl, max_width=w, max_height=h, max_batch=n)
    pre = gpu_lib.MaskPreprocessor(w, h, n, stream=ext.stream)
    x = torch.full((n, 3, 550, 550), -7.0, dtype=torch.float32, device="cuda")
    want_x = torch.empty_like(x)
    torch.cuda.synchronize()
    ext.detect_color_with_mask_pre_batch_device(pre, d.data_ptr(), h * w * 3, w * 3, w, h, n, x.data_ptr(), channels=3, rgb_order=False)
    ext.describe_batch_device()
    ext.sync()
    pre.run(d.data_ptr(), n, want_x.data_ptr())  # the three-kernel chain on the same stream
    ext.sync()
    torch.cuda.synchronize()
    assert torch.equal(x, want_x), float((x - want_x).abs().max())
    for f in range(n):
        gray = ob.color_to_gray(frames[f], rgb_order=False)
        orc = ob.Oracle(n_levels=nl)
        ko, do = orc.extract(gray)
        kg, dg = ext.batch_fetch(f)
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")
    orc = ob.Oracle(n_levels=nl)
    orc.detect(ob.color_to_gray(frames[0], rgb_order=False))
    ext.shape = (h, w)
    _same(ext.level_image(0, padded=True, frame=0), orc.level_image(0, padded=True), "padded level 0 (reflect-101 border written by the tiles)")
    assert len(ext.batch_fetch(0)[0]) > 50
