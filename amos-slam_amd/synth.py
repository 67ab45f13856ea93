"""Seeded synthetic RGB-D-like gray streams.

No TUM sequence exists where this runs, so benchmark and parity inputs are synthetic: stream `s`
is a fixed scene (three octaves of box-filtered uniform noise plus filled rectangles and discs,
which gives FAST-rich corners) and frame `k` is the window of that scene translated by (2k, k)
pixels plus +-2 grey levels of per-frame noise seeded by 1000*s + k.  Pure numpy; deterministic.
"""
from functools import lru_cache

import numpy as np

_MARGIN_X, _MARGIN_Y = 512, 256


@lru_cache(maxsize=8)
def _scene(stream: int, height: int, width: int) -> np.ndarray:
    rng = np.random.default_rng(stream)
    H, W = height + _MARGIN_Y, width + _MARGIN_X
    canvas = np.zeros((H, W), np.float32)
    for box, weight in ((4, 30.0), (16, 50.0), (64, 70.0)):
        coarse = rng.random((H // box + 2, W // box + 2)).astype(np.float32)
        up = np.kron(coarse, np.ones((box, box), np.float32))[:H, :W]
        canvas += weight * up
    n_shapes = int(200 * (H * W) / (480 * 640))
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_shapes):
        val = float(rng.integers(0, 256))
        cx, cy = int(rng.integers(0, W)), int(rng.integers(0, H))
        if rng.random() < 0.5:
            hw, hh = int(rng.integers(4, 40)), int(rng.integers(4, 40))
            canvas[max(cy - hh, 0):cy + hh, max(cx - hw, 0):cx + hw] = val
        else:
            r = int(rng.integers(4, 30))
            y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, H), max(cx - r, 0), min(cx + r + 1, W)
            sub = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
            canvas[y0:y1, x0:x1][sub] = val
    return np.clip(canvas, 0, 255).astype(np.uint8)


def frame(stream: int, k: int, height: int = 480, width: int = 640) -> np.ndarray:
    """Frame `k` of stream `stream` as a C-contiguous uint8 (height, width) array."""
    scene = _scene(stream, height, width)
    ox, oy = (2 * k) % _MARGIN_X, k % _MARGIN_Y
    win = scene[oy:oy + height, ox:ox + width].astype(np.int16)
    rng = np.random.default_rng(1000 * stream + k)
    win = win + rng.integers(-2, 3, size=win.shape, dtype=np.int16)
    return np.ascontiguousarray(np.clip(win, 0, 255).astype(np.uint8))


def frames(stream: int, k0: int, count: int, height: int = 480, width: int = 640) -> np.ndarray:
    """`count` consecutive frames of one stream, shape (count, height, width), uint8."""
    return np.stack([frame(stream, k0 + i, height, width) for i in range(count)])


def person_mask(stream: int, k: int, height: int = 480, width: int = 640) -> np.ndarray:
    """A synthetic 'person' mask (values 0/255): two or three blobs with holes, moving with k."""
    rng = np.random.default_rng(7000 + stream)
    mask = np.zeros((height, width), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(int(rng.integers(2, 4))):
        cx = int(rng.integers(width // 6, 5 * width // 6)) + 3 * k
        cy = int(rng.integers(height // 4, 3 * height // 4))
        rx, ry = int(rng.integers(30, 90)), int(rng.integers(60, 160))
        mask[((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0] = 255
        hx, hy = cx + int(rng.integers(-rx // 2, rx // 2)), cy + int(rng.integers(-ry // 2, ry // 2))
        mask[(xx - hx) ** 2 + (yy - hy) ** 2 <= int(rng.integers(3, 12)) ** 2] = 0
    return mask
