"""CPU: the oracle on the reference's own data files (tests/golden/ref_data/, copied from src/python/input and
src/python/output/mask).  The files carry no ORB outputs, so this pins nothing about OpenCV; it checks that the fixtures
decode to what the GPU tests assume and records the oracle's behaviour on real texture."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "ref_data")


@pytest.mark.parametrize("name,sha_frame,sha_mask", [
    ("1341846313.553992", None, None), ("122_rgb", None, None)])
def test_reference_frames_and_person_masks(ob, name, sha_frame, sha_mask):
    from PIL import Image
    rgb = np.ascontiguousarray(np.array(Image.open(os.path.join(DATA, name + ".png")).convert("RGB")))
    mask = np.ascontiguousarray(np.array(Image.open(os.path.join(DATA, name + "_person_mask.png")).convert("L")))
    assert rgb.shape == (480, 640, 3) and mask.shape == (480, 640)
    assert set(np.unique(mask).tolist()) == {0, 255}
    gray = ob.color_to_gray(rgb, rgb_order=True)
    want = ((rgb[..., 0].astype(np.int64) * 9798 + rgb[..., 1].astype(np.int64) * 19235 + rgb[..., 2].astype(np.int64) * 3735 + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(gray, want)
    orc = ob.Oracle()
    orc.detect(gray)
    per_level = [len(orc.level_keypoints(l)) for l in range(8)]
    assert sum(per_level) > 900 and all(n > 0 for n in per_level)
    removed = orc.gate(mask)
    kps, desc = orc.describe()
    assert len(removed) > 20 and len(kps) + len(removed) == sum(per_level)
    closed = orc.closed_mask()
    assert (closed >= mask).all()  # closing is extensive
    xs, ys = kps["x"].astype(int).clip(0, 639), kps["y"].astype(int).clip(0, 479)  # (int)search_coord, ORBextractor.cc:1721-1730
    assert (closed[ys, xs] != 0).mean() < 0.01  # (the gate tests pt * scale before the final rescale rounds differently: a few border cases)
    assert desc.shape == (len(kps), 32) and 100 < np.unpackbits(desc, axis=1).sum(1).mean() < 156
