"""The golden cases of the mask network: (input frame, weight seed, class-bias tweak) per fixture.

Shared by tools/gen_yolact_golden.py (which runs the REFERENCE's Python network on each case in the build
container and writes tests/golden/yolact_<case>.npz) and by tests/test_mask.py (which rebuilds the same
frame and weights and holds this project's network to the recorded tensors).  Frames are BGR, as the
reference's C++ hands them to `yolact::evalImage` (cv::imread order, rgbd_tum.cc:96).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DATA = os.path.join(ROOT, "tests", "golden", "ref_data")

# name -> (frame source, weight seed, [(anchor or None for all, class index incl. background, bias increment)])
CASES = {
    "seed0": ("checker:42", 0, [(None, 1, 5.0), (1, 3, 5.5)]),            # the round-1 fixture (person everywhere, car on anchor 1)
    "ref122_w0": ("png:122_rgb.png", 0, [(None, 1, 5.0), (1, 3, 5.5)]),    # the reference's own sample input src/python/input/122_rgb.png
    "tum_w0": ("png:1341846313.553992.png", 0, [(None, 1, 5.0), (1, 3, 5.5)]),   # its TUM-named sample input
    # second and third weight seeds; person on anchors 0 and 2, car on anchor 1 (which class wins the top 100 depends on the seed)
    "blobs7_w1": ("synth:7:3", 1, [(0, 1, 5.0), (2, 1, 5.0), (1, 3, 5.0)]),   # synthetic scene: persons
    "tum_w1": ("png:1341846313.553992.png", 1, [(0, 1, 5.0), (2, 1, 5.0), (1, 3, 5.0)]),   # persons
    "ref122_w3": ("png:122_rgb.png", 3, [(0, 1, 5.0), (2, 1, 5.0), (1, 3, 5.0)]),   # cars only: the empty person mask
}


def frame(case):
    """The case's 480 x 640 x 3 uint8 BGR frame."""
    src = CASES[case][0]
    kind, _, arg = src.partition(":")
    if kind == "checker":
        rng = np.random.default_rng(int(arg))
        yy, xx = np.mgrid[0:480, 0:640]
        return (rng.integers(0, 60, (480, 640, 3)) + 90 * ((xx // 80 + yy // 60) % 2)[..., None] + np.array([10, 40, 70])).astype(np.uint8)
    if kind == "png":
        from PIL import Image
        rgb = np.asarray(Image.open(os.path.join(REF_DATA, arg)).convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])
    if kind == "synth":
        import importlib
        synth = importlib.import_module("amos_slam_amd.synth")
        s, k = (int(v) for v in arg.split(":"))
        g = synth.frame(s, k).astype(np.int16)
        bgr = np.stack([g - 12, g, g + 9 + (np.arange(640) // 64)[None, :]], axis=-1)
        return np.clip(bgr, 0, 255).astype(np.uint8)
    raise KeyError(src)


def bias_class_head(net, case):
    """Random weights never reach the 0.05 class threshold: raise the bias of a few classes of the first prediction
    head (3 anchors x 81 classes), the same way in the generator and in the tests."""
    import torch
    head = net.prediction_layers[0].conf_layer.bias
    with torch.no_grad():
        b = head.detach().cpu().view(3, 81).clone()
        for anchor, cls, inc in CASES[case][2]:
            if anchor is None:
                b[:, cls] += inc
            else:
                b[anchor, cls] += inc
        head.copy_(b.view(-1).to(head.device))


def weight_seed(case):
    return CASES[case][1]
