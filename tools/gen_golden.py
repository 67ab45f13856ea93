#!/usr/bin/env python3
"""Writes tests/golden/*.npz: oracle outputs on seeded synthetic frames.

The reference C++ cannot be built or run here (needs OpenCV 4.5.1 etc.), and it ships no golden
vectors for this path, so these fixtures are produced by the CPU oracle and pin it (and, through the
GPU parity tests, the HIP path) against accidental change.  They are data: inputs are regenerated
from the seed, only their sha256 is stored."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import oracle_binding as ob  # noqa: E402

entry.load_package()
import importlib  # noqa: E402
synth = importlib.import_module("amos_slam_amd.synth")

CASES = {"c1_640x480_s0k0": (640, 480, 1000, 8, 0, 0), "c1_640x480_s3k17": (640, 480, 1000, 8, 3, 17),
         "small_320x240_s1k2": (320, 240, 500, 4, 1, 2)}
for name, (w, h, nf, nl, stream, k) in CASES.items():
    img = synth.frame(stream, k, h, w)
    orc = ob.Oracle(n_features=nf, n_levels=nl)
    kps, desc = orc.extract(img)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), meta=np.array([w, h, nf, nl, stream, k]),
                        image_sha256=np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8), keypoints=kps,
                        descriptors=desc, candidates_per_level=np.array([len(orc.level_candidates(l)) for l in range(nl)]))
    print(name, len(kps))
