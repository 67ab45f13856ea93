"""TUM RGB-D sequences as front-end input (BASELINE.json north_star: "timed ... on TUM RGB-D fr3/walking_xyz").

The reference's driver (Examples/RGB-D/rgbd_tum.cc:60-70, 96-97, 182-210) takes a sequence directory and an
associations file with one line per frame, `timestamp rgb/<file>.png timestamp depth/<file>.png`, and decodes both
images with cv::imread(IMREAD_UNCHANGED): an 8-bit 3-channel BGR frame and a 16-bit depth map.  This module reads
the same two things with PIL (the PNG decode is lossless, so the bytes equal OpenCV's; the channel order is
turned to BGR here) for bench.py (`--tum-root` / AMOS_TUM_ROOT) and the tests.  No sequence ships with this
repository or exists in the build container: whoever has the data points the bench at it.

    root/
      rgbd_dataset_freiburg3_walking_xyz/
        rgb/*.png  depth/*.png  [associations.txt | associate.txt]

`associations` may also be given explicitly (the reference keeps its own under Examples/RGB-D/associations/).
"""
import os

import numpy as np

# BASELINE.json configs[0..2] name these three sequences; short names as the reference's association files use them
SEQUENCES = {
    "fr1_xyz": "rgbd_dataset_freiburg1_xyz",
    "fr3_walking_xyz": "rgbd_dataset_freiburg3_walking_xyz",
    "fr3_walking_halfsphere": "rgbd_dataset_freiburg3_walking_halfsphere",
}
_ASSOC_NAMES = ("associations.txt", "associate.txt", "association.txt")


def load_associations(path):
    """[(timestamp, rgb relative path, depth relative path)] in file order -- LoadImages, rgbd_tum.cc:182-210:
    whitespace-separated `t rgb t depth`, empty lines skipped, the frame's timestamp is the FIRST number."""
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if len(parts) < 4:
                raise ValueError(f"{path}: expected `timestamp rgb timestamp depth`, got {line!r}")
            rows.append((float(parts[0]), parts[1], parts[3]))
    return rows


def sequence_dir(root, sequence):
    """`sequence` may be a short name (fr3_walking_xyz), a directory name under `root`, or a path."""
    for cand in (sequence, os.path.join(root, SEQUENCES.get(sequence, sequence)), os.path.join(root, sequence)):
        if os.path.isdir(cand):
            return cand
    raise FileNotFoundError(f"TUM sequence {sequence!r} not found under {root!r}")


def find_associations(seq_dir, sequence=None, explicit=None):
    if explicit:
        return explicit
    for name in _ASSOC_NAMES + ((sequence + ".txt",) if sequence else ()):
        p = os.path.join(seq_dir, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no associations file in {seq_dir} (looked for {', '.join(_ASSOC_NAMES)}); pass one explicitly")


def read_bgr(path):
    """cv::imread(path, IMREAD_UNCHANGED) of an 8-bit colour PNG: uint8 [H, W, 3] in B, G, R order."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def read_depth(path):
    """cv::imread(path, IMREAD_UNCHANGED) of a 16-bit depth PNG: uint16 [H, W] (TUM: 5000 units per metre)."""
    from PIL import Image
    with Image.open(path) as im:
        d = np.asarray(im)
    if d.dtype != np.uint16:
        d = d.astype(np.uint16)
    return np.ascontiguousarray(d)


CAMERA_RGB = 1  # `Camera.RGB: 1` in every RGB-D settings file of the reference (Examples/RGB-D/TUM1.yaml:29, TUM2.yaml:29, TUM3.yaml:28)


def bgr_to_gray(bgr, rgb_flag=CAMERA_RGB):
    """The gray frame Tracking::GrabImageRGBD hands the extractor (Tracking.cc:308-314) for a cv::imread buffer (B, G, R in memory), in
    OpenCV's 8-bit fixed point (c0 * w0 + c1 * 19235 + c2 * w2 + 16384) >> 15.  `rgb_flag` is the settings file's Camera.RGB (mbRGB):
    the TUM yamls set it to 1, so the reference applies CV_RGB2GRAY to the B, G, R buffer -- the R weight 9798 lands on the BLUE channel
    and the B weight 3735 on the RED one.  That quirk is part of what the reference's extractor sees on TUM, so it is the default here;
    rgb_flag=0 is the plain CV_BGR2GRAY.  Data preparation for the gray-input leg of the bench (the colour-input leg hands the buffer
    to the library with rgb_order=rgb_flag, which does this on the GPU)."""
    c0 = bgr[..., 0].astype(np.uint32)
    c1 = bgr[..., 1].astype(np.uint32)
    c2 = bgr[..., 2].astype(np.uint32)
    w0, w2 = (9798, 3735) if rgb_flag else (3735, 9798)
    return ((c0 * w0 + c1 * 19235 + c2 * w2 + 16384) >> 15).astype(np.uint8)


def load_sequence(root, sequence, n_frames, associations=None, with_depth=False, start=0):
    """The first `n_frames` frames (from `start`) of a sequence: dict(name, timestamps [n] float64, bgr [n, H, W, 3] uint8,
    depth [n, H, W] uint16 or None, frames_in_sequence).  A sequence shorter than n_frames wraps around (the bench wants
    full batches; `wrapped` says so)."""
    seq = sequence_dir(root, sequence)
    rows = load_associations(find_associations(seq, sequence, associations))
    if not rows:
        raise ValueError(f"{seq}: empty associations file")
    idx = [(start + k) % len(rows) for k in range(n_frames)]
    cache = {}
    bgr, depth, ts = [], [], []
    for i in idx:
        if i not in cache:
            t, rgb_rel, d_rel = rows[i]
            cache[i] = (t, read_bgr(os.path.join(seq, rgb_rel)), read_depth(os.path.join(seq, d_rel)) if with_depth else None)
        t, b, d = cache[i]
        ts.append(t)
        bgr.append(b)
        depth.append(d)
    shapes = {b.shape for b in bgr}
    if len(shapes) != 1:
        raise ValueError(f"{seq}: frames of different sizes {shapes}")
    return {"name": os.path.basename(os.path.normpath(seq)), "timestamps": np.asarray(ts, np.float64), "bgr": np.stack(bgr),
            "depth": np.stack(depth) if with_depth else None, "frames_in_sequence": len(rows), "wrapped": n_frames > len(rows)}
