"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Integer / byte / index work => every comparison is bit-exact (tobytes equality)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(640, 480, 1000, 8), (320, 240, 500, 4), (752, 480, 1200, 8), (413, 307, 300, 5)]


def _pair(gpu_lib, ob, w, h, nf, nl, **kw):
    ext = gpu_lib.OrbExtractor(n_features=nf, n_levels=nl, max_width=w, max_height=h, **kw)
    orc = ob.Oracle(n_features=nf, n_levels=nl)
    return ext, orc


def _same(a, b, what):
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.tobytes() != b.tobytes():
        if a.dtype.names:
            bad = [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
            idx = np.nonzero(a[bad[0]] != b[bad[0]])[0][:5]
            raise AssertionError(f"{what}: fields {bad} differ, first at {idx}: {a[idx]} vs {b[idx]}")
        idx = np.argwhere(a != b)[:5]
        raise AssertionError(f"{what}: {np.count_nonzero(a != b)} elements differ, first at {idx.tolist()}")


def test_tables(gpu_lib, ob):
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    tg, to = ext.tables(), orc.tables()
    for k in to:
        _same(tg[k], to[k], k)
    assert list(tg["features_per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(tg["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    lw, lh = ext.level_sizes(640, 480)
    assert list(lw) == [640, 533, 444, 370, 309, 257, 214, 179]
    assert list(lh) == [480, 400, 333, 278, 231, 193, 161, 134]


@pytest.mark.parametrize("config", ["c2", "c3", "c5"])
def test_every_single_gpu_baseline_config(gpu_lib, ob, synth, config):
    """Collected FIRST on purpose (tests/conftest.py orders this file first): two consecutive frames of every single-GPU
    BASELINE.json configuration at full geometry -- configs[1] (c2: 640x480 L8 N1000, mask off), configs[2] (c3: the same
    geometry through detect -> gate with a person mask -> describe; the mask NETWORK has its own parity tests in test_mask.py)
    and configs[4] (c5: 1920x1080 L12 N4000) -- extract + N x N best-2 match of frame 1 against frame 0, bit-exact vs the oracle,
    so that no later failure can leave a configuration unexercised."""
    w, h, nf, nl = (1920, 1080, 4000, 12) if config == "c5" else (640, 480, 1000, 8)
    ext, orc = _pair(gpu_lib, ob, w, h, nf, nl)
    got_d, want_d = [], []
    for k in range(2):
        img = synth.frame(31, k, h, w)
        if config == "c3":
            mask = synth.person_mask(31, k)
            ext.detect(img)
            orc.detect(img)
            _same(ext.gate(mask), orc.gate(mask), f"{config} frame {k} removed keypoints")
            kg, dg = ext.describe()
            ko, do = orc.describe()
        else:
            kg, dg = ext.extract(img)
            ko, do = orc.extract(img)
        _same(kg, ko, f"{config} frame {k} keypoints")
        _same(dg, do, f"{config} frame {k} descriptors")
        assert len(kg) > nf * 0.5
        got_d.append(dg)
        want_d.append(do)
    m = gpu_lib.OrbMatcher()
    _same(m.bruteforce_best2(got_d[1], got_d[0]), ob.bruteforce_best2(want_d[1], want_d[0]), f"{config} match frame 1 vs 0")


@pytest.mark.parametrize("w,h,nf,nl", SIZES)
def test_stages_bit_exact(gpu_lib, ob, synth, w, h, nf, nl):
    img = synth.frame(11, 3, h, w)
    ext, orc = _pair(gpu_lib, ob, w, h, nf, nl)
    ext.detect(img)
    orc.detect(img)
    for l in range(nl):
        _same(ext.level_image(l, padded=True), orc.level_image(l, padded=True), f"padded level {l}")
    for l in range(nl):
        _same(ext.level_candidates(l), orc.level_candidates(l), f"FAST candidates level {l}")
    for l in range(nl):
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"keypoints level {l}")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    for l in range(nl):
        _same(ext.blurred_image(l), orc.blurred_image(l), f"blurred level {l}")
    _same(kg, ko, "final keypoints")
    _same(dg, do, "descriptors")
    assert len(kg) >= nf * 0.9


@pytest.mark.parametrize("sf,nl", [(1.5, 5), (2.0, 3), (1.1, 6)])
def test_other_scale_factors(gpu_lib, ob, synth, sf, nl):
    """scaleFactor 2.0 takes the four-loads-per-row pyramid kernel (the 8-byte tap window needs scaleFactor < 2)."""
    img = synth.frame(17, 1)
    ext = gpu_lib.OrbExtractor(n_features=800, scale_factor=sf, n_levels=nl)
    orc = ob.Oracle(n_features=800, scale_factor=sf, n_levels=nl)
    kg, dg = ext.extract(img)
    ko, do = orc.extract(img)
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")
    for l in range(nl):
        _same(ext.level_image(l, padded=True), orc.level_image(l, padded=True), f"level {l} incl. border")


def test_random_geometries(gpu_lib, ob):
    """Fuzz over frame sizes, level counts, scale factors, feature counts and thresholds: every combination
    exercises different edge handling (partial 16-byte import pieces, partial 8-pixel FAST groups, level widths that
    are not multiples of 4, cells clipped at the right / bottom border, quotas larger than the candidate count)."""
    rng = np.random.default_rng(2024)
    done = 0
    for trial in range(60):
        w, h = int(rng.integers(130, 820)), int(rng.integers(110, 620))
        nl = int(rng.integers(1, 9))
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.33, 1.5])))
        nf = int(rng.integers(50, 2500))
        ini, mn = int(rng.integers(8, 40)), int(rng.integers(2, 8))
        kind = trial % 3
        if kind == 0:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:  # smooth blobs + sharp rectangles
            yy, xx = np.mgrid[0:h, 0:w]
            img = (127 + 90 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
            for _ in range(40):
                x0, y0 = int(rng.integers(0, w - 20)), int(rng.integers(0, h - 20))
                img[y0:y0 + int(rng.integers(4, 40)), x0:x0 + int(rng.integers(4, 40))] = int(rng.integers(0, 256))
        else:            # mostly flat with a few dots (min threshold path)
            img = np.full((h, w), 120, np.uint8)
            for _ in range(300):
                img[int(rng.integers(0, h)), int(rng.integers(0, w))] = int(rng.integers(0, 256))
        try:
            # every other handle is allocated for a larger frame than it is given
            mw, mh = (w, h) if trial % 2 else (w + int(rng.integers(0, 200)), h + int(rng.integers(0, 150)))
            ext = gpu_lib.OrbExtractor(n_features=nf, scale_factor=sf, n_levels=nl, ini_th=ini, min_th=mn, max_width=mw, max_height=mh)
            if trial % 2 == 0:
                ext.level_sizes(w, h)
                ext.detect(img)   # geometry of the smaller frame
        except gpu_lib.AmosError:
            with pytest.raises(RuntimeError):   # the oracle rejects the same geometry (a level without a FAST cell)
                ob.Oracle(n_features=nf, scale_factor=sf, n_levels=nl, ini_th=ini, min_th=mn).detect(img)
            continue
        orc = ob.Oracle(n_features=nf, scale_factor=sf, n_levels=nl, ini_th=ini, min_th=mn)
        kg, dg = ext.extract(img)
        ko, do = orc.extract(img, cap=max(4 * nf + 64, 4096))
        _same(kg, ko, f"trial {trial} ({w}x{h} L{nl} sf{sf} N{nf} th{ini}/{mn}) keypoints")
        _same(dg, do, f"trial {trial} descriptors")
        done += 1
    assert done >= 40


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_extract_many_frames(gpu_lib, ob, synth, seed):
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    img = synth.frame(seed, 7 * seed)
    kg, dg = ext.extract(img)
    ko, do = orc.extract(img)
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")


def test_low_texture_uses_min_threshold(gpu_lib, ob):
    """Cells with no corner at iniThFAST fall back to minThFAST (ORBextractor.cc:1126-1139)."""
    rng = np.random.default_rng(5)
    img = np.full((480, 640), 120, np.uint8)
    img[::16, :] = 128  # faint grid: corners only at the low threshold
    img[:, ::16] = 128
    img[100:200, 100:300] = rng.integers(0, 255, (100, 200), dtype=np.uint8)  # one textured patch
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    kg, dg = ext.extract(img)
    ko, do = orc.extract(img)
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")
    assert len(kg) > 0


@pytest.mark.parametrize("kind", ["noise", "checker", "salt"])
def test_dense_corner_images(gpu_lib, ob, kind):
    """Adversarial corner densities: white noise (most pixels pass the FAST pre-test, the per-cell
    candidate list overflows), a 2-px checkerboard and isolated salt pixels (maximal NMS survivors)."""
    rng = np.random.default_rng(12)
    if kind == "noise":
        img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    elif kind == "checker":
        yy, xx = np.mgrid[0:480, 0:640]
        img = (((yy // 2 + xx // 2) & 1) * 200 + 20).astype(np.uint8)
    else:
        img = np.full((480, 640), 30, np.uint8)
        img[::2, ::2] = rng.integers(100, 255, (240, 320), dtype=np.uint8)
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    ext.detect(img)
    orc.detect(img)
    for l in range(8):
        _same(ext.level_candidates(l), orc.level_candidates(l), f"FAST candidates level {l}")
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"keypoints level {l}")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")


def test_constant_image_gives_nothing(gpu_lib, ob):
    """No corner anywhere => zero keypoints, descriptors released (ORBextractor.cc:1590)."""
    img = np.full((480, 640), 77, np.uint8)
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    kg, dg = ext.extract(img)
    ko, do = orc.extract(img)
    assert len(kg) == 0 and len(ko) == 0 and dg.shape == (0, 32)


def test_bright_square_corners(gpu_lib, ob):
    """Hand-checkable micro case: one bright square on black.  At level 0 the corner pixel and its
    diagonal inner neighbour both score 199, so the strict 3x3 non-max suppression removes both and
    the level is empty; the resampled levels keep exactly the four corners."""
    img = np.zeros((480, 640), np.uint8)
    img[200:260, 300:380] = 200
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    ext.detect(img)
    orc.detect(img)
    for l in range(8):
        _same(ext.level_candidates(l), orc.level_candidates(l), f"candidates level {l}")
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"keypoints level {l}")
    assert len(ext.level_keypoints(0)) == 0
    assert len(ext.level_keypoints(1)) == 4


def test_frame_too_small_is_rejected(gpu_lib):
    """A level without a FAST cell divides by zero in the reference (ORBextractor.cc:1083-1086);
    the C ABI reports AMOS_ERR_INVALID instead."""
    with pytest.raises(gpu_lib.AmosError):
        gpu_lib.OrbExtractor(n_features=500, n_levels=8, max_width=160, max_height=120)


def test_gate_mask_only(gpu_lib, ob, synth):
    img = synth.frame(2, 5)
    mask = synth.person_mask(2, 5)
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    ext.detect(img)
    orc.detect(img)
    rg = ext.gate(mask)
    ro = orc.gate(mask)
    _same(ext.closed_mask(), orc.closed_mask(), "closed mask")
    _same(rg, ro, "removed keypoints")
    for l in range(8):
        _same(ext.level_keypoints(l), orc.level_keypoints(l), f"kept keypoints level {l}")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    _same(kg, ko, "keypoints after gate")
    _same(dg, do, "descriptors after gate")
    assert 0 < len(rg) < 1000


def test_gate_with_labels(gpu_lib, ob, synth):
    img = synth.frame(4, 1)
    mask = synth.person_mask(4, 1)
    rng = np.random.default_rng(9)
    labels = np.kron(rng.integers(1, 16, (480 // 32, 640 // 32)), np.ones((32, 32))).astype(np.float64)
    center_ids = rng.permutation(15).astype(np.int32)
    rm = np.zeros(15, np.int32)
    rm[[2, 7, 11]] = 1
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    ext.detect(img)
    orc.detect(img)
    rg = ext.gate(mask, labels, center_ids, rm)
    ro = orc.gate(mask, labels, center_ids, rm)
    _same(rg, ro, "removed keypoints")
    kg, dg = ext.describe()
    ko, do = orc.describe()
    _same(kg, ko, "keypoints after gate")
    _same(dg, do, "descriptors after gate")


def test_gate_zero_mask_equals_extract(gpu_lib, synth):
    """a7 -> a8 (all-zero mask) -> a9 == a11."""
    img = synth.frame(6, 2)
    ext = gpu_lib.OrbExtractor()
    k1, d1 = ext.extract(img)
    ext.detect(img)
    removed = ext.gate(np.zeros((480, 640), np.uint8))
    k2, d2 = ext.describe()
    assert len(removed) == 0
    _same(k1, k2, "keypoints")
    _same(d1, d2, "descriptors")


def test_set_level_keypoints_roundtrip(gpu_lib, ob, synth):
    """The caller hands the per-level vectors back (Frame.cc:491-496): drop every other keypoint
    on the host and describe."""
    img = synth.frame(8, 0)
    ext, orc = _pair(gpu_lib, ob, 640, 480, 1000, 8)
    ext.detect(img)
    orc.detect(img)
    for l in range(8):
        k = ext.level_keypoints(l)[::2]
        ext.set_level_keypoints(l, k)
        orc.set_level_keypoints(l, k)
    kg, dg = ext.describe()
    ko, do = orc.describe()
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")


def test_full_benchmark_size_properties(gpu_lib, ob, synth):
    """BASELINE configs[1] at the bench's launch size (128 frames of 640x480 per lane), where the oracle would take
    minutes: size-independent properties on every frame + the oracle on a sample.
      determinism (two passes give identical bytes), keypoints ordered by level and inside their level's FAST region,
      per-level counts within the quad-tree's bound, sizes = int(31 * scale[level]), a frame matched against itself
      returns the identity with distance 0 (and second best > 0 unless the descriptor is duplicated),
      matching k against k-1 equals matching through the single-pair API."""
    import torch
    n = 128
    frames = synth.frames(2, 0, n)
    ext = gpu_lib.OrbExtractor(max_batch=n)
    mt = gpu_lib.OrbMatcher(stream=ext.stream)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    d_kps, d_desc, d_counts, cap = ext.batch_results_device()
    pairs_q = torch.arange(n, dtype=torch.int32, device="cuda")
    pairs_prev = (pairs_q - 1) % n
    d_self = torch.zeros((n, cap, 4), dtype=torch.int32, device="cuda")
    d_prev = torch.zeros((n, cap, 4), dtype=torch.int32, device="cuda")
    digests = []
    for _ in range(2):
        ext.extract_batch_device(d.data_ptr(), 480 * 640, 640, 640, 480, n)
        mt.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pairs_q.data_ptr(), pairs_q.data_ptr(), n, cap, 256, d_self.data_ptr())
        mt.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pairs_q.data_ptr(), pairs_prev.data_ptr(), n, cap, 256, d_prev.data_ptr())
        ext.sync()
        torch.cuda.synchronize()
        res = [ext.batch_fetch(f) for f in range(n)]
        digests.append((b"".join(k.tobytes() + dd.tobytes() for k, dd in res), d_prev.cpu().numpy().tobytes()))
    assert digests[0] == digests[1], "two passes over the same batch differ"
    tab = ext.tables()
    lw, lh = ext.level_sizes(640, 480)
    self_m, prev_m = d_self.cpu().numpy(), d_prev.cpu().numpy()
    for f, (k, dd) in enumerate(res):
        m = len(k)
        assert 900 <= m <= cap
        assert np.all(np.diff(k["octave"]) >= 0), "levels out of order"
        counts = np.bincount(k["octave"], minlength=8)
        assert np.all(counts <= tab["features_per_level"] + 3 * 4), counts       # each of the last divisions adds <= 3 nodes
        sc = tab["scale"][k["octave"]]
        assert np.array_equal(k["size"], np.floor(np.float32(31) * sc).astype(np.float32))   # int scaledPatchSize, ORBextractor.cc:1180
        xl, yl = k["x"] / np.where(k["octave"] == 0, 1, sc), k["y"] / np.where(k["octave"] == 0, 1, sc)
        assert np.all(xl >= 16 - 1e-3) and np.all(yl >= 16 - 1e-3)
        assert np.all(xl <= lw[k["octave"]] - 16 + 1e-3) and np.all(yl <= lh[k["octave"]] - 16 + 1e-3)
        sm = self_m[f, :m]
        assert np.array_equal(sm[:, 1], np.zeros(m, np.int32)), "self distance"
        same = sm[:, 0] == np.arange(m)
        # best index differs from i only where an EARLIER keypoint has the same descriptor (first wins)
        for i in np.nonzero(~same)[0]:
            assert sm[i, 0] < i and np.array_equal(dd[sm[i, 0]], dd[i])
    for f in (0, 77, 127):                                                         # oracle on a sample
        ko, do = ob.Oracle().extract(frames[f])
        _same(res[f][0], ko, f"frame {f} keypoints")
        _same(res[f][1], do, f"frame {f} descriptors")
        want = ob.bruteforce_best2(res[f][1], res[(f - 1) % n][1])
        got = prev_m[f, :len(ko)]
        for c, name in enumerate(("best_idx", "best_dist", "second_idx", "second_dist")):
            assert np.array_equal(got[:, c], want[name]), (f, name)


def test_batch_equals_single(gpu_lib, ob, synth):
    """The batched device-resident path gives, per frame, what the single-frame path gives."""
    import torch
    n = 6
    frames = synth.frames(1, 0, n)
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ext.extract_batch_device(d.data_ptr(), 480 * 640, 640, 640, 480, n)
    ext.sync()
    orc = ob.Oracle()
    for f in range(n):
        kg, dg = ext.batch_fetch(f)
        ko, do = orc.extract(frames[f])
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")


def test_batch_with_unaligned_strides(gpu_lib, ob, synth):
    """Frames inside a larger device buffer with odd row / frame strides and an odd base offset
    (the level-0 import then takes its byte path)."""
    import torch
    n, w, h = 3, 322, 241
    frames = synth.frames(7, 0, n, h, w)
    row_stride, frame_stride = w + 5, (w + 5) * h + 3
    buf = torch.zeros(7 + n * frame_stride + 64, dtype=torch.uint8)
    for f in range(n):
        for y in range(h):
            o = 7 + f * frame_stride + y * row_stride
            buf[o:o + w] = torch.from_numpy(frames[f, y])
    d = buf.cuda()
    torch.cuda.synchronize()
    ext = gpu_lib.OrbExtractor(n_features=500, n_levels=4, max_width=w, max_height=h, max_batch=n)
    ext.extract_batch_device(d.data_ptr() + 7, frame_stride, row_stride, w, h, n)
    ext.sync()
    for f in range(n):
        kg, dg = ext.batch_fetch(f)
        ko, do = ob.Oracle(n_features=500, n_levels=4).extract(frames[f])
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")


def test_random_batches_masks_and_colour(gpu_lib, ob):
    """Fuzz of the device-resident batch entry points: random sizes, batch fill, strides and base offsets, gray or
    colour input, optionally the staged detect -> gate (random masks) -> describe flow."""
    import torch
    rng = np.random.default_rng(77)
    for trial in range(14):
        w = int(rng.integers(240, 700))
        h = int(rng.integers(160, min(500, w)))          # landscape: the reference needs round(width / height) >= 1 quad-tree roots
        nmax = int(rng.integers(1, 5))
        n = int(rng.integers(1, nmax + 1))
        nf, nl = int(rng.integers(200, 1500)), int(rng.integers(2, 6))
        ch = int(rng.choice([1, 1, 3, 4]))
        rgb = bool(rng.integers(0, 2))
        pad, off = int(rng.integers(0, 9)) * ch, int(rng.integers(0, 16))
        row_stride = w * ch + pad
        frame_stride = row_stride * h + int(rng.integers(0, 64))
        host = rng.integers(0, 256, off + n * frame_stride + 64, dtype=np.uint8)
        # blocky content gives corners at several levels
        for f in range(n):
            img = np.kron(rng.integers(0, 256, ((h + 15) // 16, (w + 15) // 16, ch), dtype=np.uint8), np.ones((16, 16, 1), np.uint8))[:h, :w]
            img = np.clip(img.astype(np.int16) + rng.integers(-12, 12, img.shape), 0, 255).astype(np.uint8)
            view = host[off + f * frame_stride: off + f * frame_stride + row_stride * h].reshape(h, row_stride)
            view[:, :w * ch] = img.reshape(h, w * ch)
        d = torch.from_numpy(host).cuda()
        torch.cuda.synchronize()
        ext = gpu_lib.OrbExtractor(n_features=nf, n_levels=nl, max_width=w + int(rng.integers(0, 50)), max_height=h + int(rng.integers(0, 20)),
                                   max_batch=nmax)
        frames = []
        for f in range(n):
            v = host[off + f * frame_stride: off + f * frame_stride + row_stride * h].reshape(h, row_stride)[:, :w * ch].reshape(h, w, ch)
            frames.append(v[..., 0].copy() if ch == 1 else ob.color_to_gray(v.copy(), rgb))
        staged = ch == 1 and trial % 2 == 0
        if ch > 1:
            ext.extract_batch_device_color(d.data_ptr() + off, frame_stride, row_stride, w, h, n, ch, rgb)
        elif staged:
            masks = (rng.random((n, h, w)) < 0.002).astype(np.uint8) * 255
            d_masks = torch.from_numpy(masks).cuda()
            torch.cuda.synchronize()
            ext.detect_batch_device(d.data_ptr() + off, frame_stride, row_stride, w, h, n)
            ext.gate_batch_device(d_masks.data_ptr(), h * w, w)
            ext.describe_batch_device()
        else:
            ext.extract_batch_device(d.data_ptr() + off, frame_stride, row_stride, w, h, n)
        ext.sync()
        torch.cuda.synchronize()
        for f in range(n):
            orc = ob.Oracle(n_features=nf, n_levels=nl)
            if staged:
                orc.detect(frames[f])
                orc.gate(masks[f])
                ko, do = orc.describe()
            else:
                ko, do = orc.extract(frames[f])
            kg, dg = ext.batch_fetch(f)
            what = f"trial {trial} frame {f} ({w}x{h}x{ch} n={n}/{nmax} strides {row_stride},{frame_stride}+{off} staged={staged})"
            _same(kg, ko, what + " keypoints")
            _same(dg, do, what + " descriptors")


def test_staged_batch_with_masks(gpu_lib, ob, synth):
    """Full front-end, device resident: detect -> gate (closed mask) -> describe for a batch,
    frame by frame equal to the oracle's a7 -> a8 -> a9."""
    import torch
    n = 5
    frames = synth.frames(3, 4, n)
    masks = np.stack([synth.person_mask(3, 4 + k) for k in range(n)])
    masks[2] = 0  # one frame without dynamic objects
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d_frames, d_masks = torch.from_numpy(frames).cuda(), torch.from_numpy(masks).cuda()
    torch.cuda.synchronize()
    ext.detect_batch_device(d_frames.data_ptr(), 480 * 640, 640, 640, 480, n)
    ext.gate_batch_device(d_masks.data_ptr(), 480 * 640, 640)
    ext.describe_batch_device()
    ext.sync()
    for f in range(n):
        orc = ob.Oracle()
        orc.detect(frames[f])
        orc.gate(masks[f])
        ko, do = orc.describe()
        kg, dg = ext.batch_fetch(f)
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")


@pytest.mark.parametrize("channels,rgb", [(3, False), (3, True), (4, False)])
def test_color_import_and_rgbd_glue(gpu_lib, ob, synth, channels, rgb):
    """8f rows: cvtColor fused into the level-0 import, then ComputeStereoFromRGBD + grid cells on the device."""
    import torch
    n = 3
    rng = np.random.default_rng(20 + channels + int(rgb))
    gray = synth.frames(9, 0, n)
    color = np.clip(gray[..., None].astype(np.int16) + rng.integers(-40, 40, (n, 480, 640, channels)), 0, 255).astype(np.uint8)
    raw = rng.integers(0, 30000, (n, 480, 640)).astype(np.uint16)
    raw[rng.random(raw.shape) < 0.25] = 0
    factor = float(np.float32(1.0) / np.float32(5000.0))
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d_color, d_raw = torch.from_numpy(color).cuda(), torch.from_numpy(raw.view(np.int16)).cuda()
    torch.cuda.synchronize()
    ext.extract_batch_device_color(d_color.data_ptr(), 480 * 640 * channels, 640 * channels, 640, 480, n, channels, rgb)
    _, _, _, cap = ext.batch_results_device()
    d_ur = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    d_dep = torch.zeros_like(d_ur)
    d_cell = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    bounds = (0.0, 640.0, 0.0, 480.0)
    ext.rgbd_glue_batch_device(d_raw.data_ptr(), True, factor, 480 * 640 * 2, 640 * 2, 40.0, bounds, d_ur.data_ptr(), d_dep.data_ptr(),
                               d_cell.data_ptr())
    ext.sync()
    torch.cuda.synchronize()
    for f in range(n):
        g = ob.color_to_gray(color[f], rgb)
        ko, do = ob.Oracle().extract(g)
        kg, dg = ext.batch_fetch(f)
        _same(kg, ko, f"frame {f} keypoints")
        _same(dg, do, f"frame {f} descriptors")
        depth = (raw[f].astype(np.float32) * np.float32(factor)).astype(np.float32)
        ur, dep, cell = ob.rgbd_glue(ko, depth, 40.0, bounds)
        m = len(ko)
        _same(d_ur[f, :m].cpu().numpy(), ur, "mvuRight")
        _same(d_dep[f, :m].cpu().numpy(), dep, "mvDepth")
        _same(d_cell[f, :m].cpu().numpy(), cell, "grid cell")
        assert (ur > 0).sum() > m // 2


def test_undistort_bounds_and_glue_with_distortion(gpu_lib, ob, synth):
    """8f-1: UndistortKeyPoints + ComputeImageBounds + ComputeStereoFromRGBD / grid cells with a distorted camera (TUM1)."""
    import torch
    n = 2
    fx, fy, cx, cy = 517.306408, 516.469215, 318.643040, 255.313989
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    rng = np.random.default_rng(31)
    frames = synth.frames(11, 0, n)
    depth = (0.5 + 3.0 * rng.random((n, 480, 640))).astype(np.float32)
    depth[rng.random(depth.shape) < 0.2] = 0
    bounds = gpu_lib.image_bounds(640, 480, fx, fy, cx, cy, dist)
    assert bounds == ob.image_bounds(640, 480, fx, fy, cx, cy, dist)
    assert gpu_lib.image_bounds(640, 480, fx, fy, cx, cy, dist[:0]) == (0.0, 640.0, 0.0, 480.0)
    ext = gpu_lib.OrbExtractor(max_batch=n)
    d_frames, d_depth = torch.from_numpy(frames).cuda(), torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    ext.extract_batch_device(d_frames.data_ptr(), 480 * 640, 640, 640, 480, n)
    _, _, _, cap = ext.batch_results_device()
    d_un = torch.zeros((n, cap, 7), dtype=torch.float32, device="cuda")
    d_ur = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    d_dep = torch.zeros_like(d_ur)
    d_cell = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ext.undistort_batch_device(fx, fy, cx, cy, dist, d_un.data_ptr())
    ext.rgbd_glue_batch_device(d_depth.data_ptr(), False, 1.0, 480 * 640 * 4, 640 * 4, 40.0, bounds, d_ur.data_ptr(), d_dep.data_ptr(),
                               d_cell.data_ptr(), d_kps_un=d_un.data_ptr())
    ext.sync()
    torch.cuda.synchronize()
    un = d_un.cpu().numpy().view(np.uint8).reshape(n, cap, 28)
    for f in range(n):
        kps, _ = ext.batch_fetch(f)
        m = len(kps)
        want = kps.copy()
        xy = ob.undistort_points(np.stack([kps["x"], kps["y"]], 1), fx, fy, cx, cy, dist)
        want["x"], want["y"] = xy[:, 0], xy[:, 1]
        got = np.frombuffer(un[f, :m].tobytes(), ob.KP_DTYPE)
        _same(got, want, "mvKeysUn")
        assert np.abs(want["x"] - kps["x"]).max() > 0.5
        ur, dep, cell = ob.rgbd_glue(kps, depth[f], 40.0, bounds, kps_un=want)
        _same(d_ur[f, :m].cpu().numpy(), ur, "mvuRight")
        _same(d_dep[f, :m].cpu().numpy(), dep, "mvDepth")
        _same(d_cell[f, :m].cpu().numpy(), cell, "grid cell")
    # k1 == 0: mvKeysUn = mvKeys
    ext.undistort_batch_device(fx, fy, cx, cy, np.zeros(5, np.float32), d_un.data_ptr())
    ext.sync()
    torch.cuda.synchronize()
    kps, _ = ext.batch_fetch(0)
    _same(np.frombuffer(d_un.cpu().numpy().view(np.uint8).reshape(n, cap, 28)[0, :len(kps)].tobytes(), ob.KP_DTYPE), kps, "identity")


def test_hd_config(gpu_lib, ob, synth):
    """BASELINE.json configs[4] geometry: 1920x1080, 4000 features, 12 levels."""
    img = synth.frame(21, 0, 1080, 1920)
    ext, orc = _pair(gpu_lib, ob, 1920, 1080, 4000, 12)
    kg, dg = ext.extract(img)
    ko, do = orc.extract(img)
    _same(kg, ko, "keypoints")
    _same(dg, do, "descriptors")


# ------------------------------------------------------------------------------- matcher

def _descs(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def test_hamming_known_answers(gpu_lib):
    m = gpu_lib.OrbMatcher()
    zero, one = np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8)
    assert m.distances(zero, one)[0, 0] == 256
    assert m.distances(zero, zero)[0, 0] == 0
    flips = np.zeros((256, 32), np.uint8)
    for b in range(256):
        flips[b, b // 8] = 1 << (b % 8)
    assert (m.distances(zero, flips) == 1).all()
    rng = np.random.default_rng(0)
    a, b = _descs(rng, 37), _descs(rng, 53)
    want = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(m.distances(a, b), want)


def test_dense_distances_vs_oracle(gpu_lib, ob):
    rng = np.random.default_rng(1)
    q, t = _descs(rng, 1003), _descs(rng, 997)
    m = gpu_lib.OrbMatcher()
    _same(m.distances(q, t), ob.distances(q, t), "dense distances")


def _random_lists(rng, nq, nt, maxlen):
    lens = rng.integers(0, maxlen, nq)
    off = np.zeros(nq + 1, np.int32)
    off[1:] = np.cumsum(lens)
    idx = rng.integers(0, nt, off[-1]).astype(np.int32)
    return off, idx


def test_list_primitives_vs_oracle(gpu_lib, ob):
    rng = np.random.default_rng(2)
    q, t = _descs(rng, 700), _descs(rng, 900)
    t[100:140] = t[100]  # exact duplicates: ties must resolve to the first candidate
    off, idx = _random_lists(rng, 700, 900, 150)
    m = gpu_lib.OrbMatcher()
    _same(m.list_distances(q, t, off, idx), ob.list_distances(q, t, off, idx), "list distances")
    for init in (256, 2 ** 31 - 1, 100, 120):
        _same(m.list_best2(q, t, off, idx, init), ob.list_best2(q, t, off, idx, init), f"list best2 init={init}")


BF_KERNELS = ["auto", "popcount", "mfma"]  # amos_match_set_bruteforce_kernel: identical results whichever runs


@pytest.mark.parametrize("kernel", BF_KERNELS)
def test_bruteforce_vs_oracle(gpu_lib, ob, kernel):
    rng = np.random.default_rng(3)
    q, t = _descs(rng, 1001), _descs(rng, 999)
    t[500:520] = q[7]  # many equal best distances
    q[9] = q[7]
    t[3] = 0           # extreme popcounts: all-zero / all-one descriptors on both sides
    t[4] = 255
    q[11] = 0
    q[12] = 255
    m = gpu_lib.OrbMatcher()
    m.set_bruteforce_kernel(kernel)
    for init in (256, 2 ** 31 - 1, 110):
        _same(m.bruteforce_best2(q, t, init), ob.bruteforce_best2(q, t, init), f"bf best2 init={init}")
    # empty / ragged
    assert len(m.bruteforce_best2(q[:0], t)) == 0
    r = m.bruteforce_best2(q[:3], t[:0])
    assert (r["best_idx"] == -1).all() and (r["best_dist"] == 256).all()
    _same(m.bruteforce_best2(q[:1], t[:1]), ob.bruteforce_best2(q[:1], t[:1]), "1x1")


@pytest.mark.parametrize("kernel", BF_KERNELS)
def test_matcher_size_fuzz(gpu_lib, ob, kernel):
    """Brute-force and list reductions over awkward set sizes (wave / tile boundaries, 1, 2, 3 descriptors, train sets
    shorter than the four per-wave quarters or one 32-row MFMA tile) and gates; descriptors drawn from a small pool so
    that ties are common."""
    rng = np.random.default_rng(99)
    pool = _descs(rng, 37)
    m = gpu_lib.OrbMatcher()
    m.set_bruteforce_kernel(kernel)
    for nq, nt in ((1, 1), (1, 2), (2, 3), (3, 1), (5, 7), (63, 65), (64, 64), (65, 63), (127, 4), (4, 129), (257, 255), (1, 1500), (1300, 2), (513, 1025),
                   (128, 32), (129, 33), (31, 31), (96, 64), (97, 127), (1000, 1000)):
        q = pool[rng.integers(0, len(pool), nq)].copy()
        t = pool[rng.integers(0, len(pool), nt)].copy()
        flip = rng.random((nt, 32)) < 0.05                      # perturb some train descriptors
        t ^= (flip * rng.integers(0, 256, (nt, 32))).astype(np.uint8)
        for init in (256, 40, 1, 2 ** 31 - 1):
            _same(m.bruteforce_best2(q, t, init), ob.bruteforce_best2(q, t, init), f"bf {nq}x{nt} init={init}")
        _same(m.distances(q, t), ob.distances(q, t), f"dense {nq}x{nt}")
        off, idx = _random_lists(rng, nq, nt, min(nt, 40))
        _same(m.list_distances(q, t, off, idx), ob.list_distances(q, t, off, idx), f"list dist {nq}x{nt}")
        for init in (256, 30):
            _same(m.list_best2(q, t, off, idx, init), ob.list_best2(q, t, off, idx, init), f"list best2 {nq}x{nt} init={init}")


@pytest.mark.parametrize("kernel", ["popcount", "mfma"])
def test_bruteforce_batch_device_kernels(gpu_lib, ob, kernel):
    """amos_match_bruteforce_best2_batch_device with ragged per-frame counts, both kernels, against the oracle."""
    import torch
    rng = np.random.default_rng(17)
    cap, counts = 1100, [1000, 1, 0, 333, 1100, 97, 128, 31]
    pool = _descs(rng, 200)
    desc = np.zeros((len(counts), cap, 32), np.uint8)
    for f, n in enumerate(counts):
        desc[f, :n] = pool[rng.integers(0, len(pool), n)]
        flip = rng.random((n, 32)) < 0.1
        desc[f, :n] ^= (flip * rng.integers(0, 256, (n, 32))).astype(np.uint8)
    desc[7, 31:] = 0xAB  # garbage beyond the count must not be matched
    pq = np.array([0, 1, 2, 3, 4, 5, 6, 7, 0, 4, 3], np.int32)
    pt = np.array([4, 0, 0, 2, 0, 6, 5, 7, 0, 4, 1], np.int32)
    d_desc, d_counts = torch.from_numpy(desc).cuda(), torch.tensor(counts, dtype=torch.int32).cuda()
    d_pq, d_pt = torch.from_numpy(pq).cuda(), torch.from_numpy(pt).cuda()
    m = gpu_lib.OrbMatcher()
    m.set_bruteforce_kernel(kernel)
    for init in (256, 60, 2 ** 31 - 1):
        d_out = torch.full((len(pq), cap, 4), -7, dtype=torch.int32).cuda()
        torch.cuda.synchronize()
        m.bruteforce_best2_batch_device(d_desc.data_ptr(), cap * 32, d_counts.data_ptr(), d_pq.data_ptr(), d_pt.data_ptr(), len(pq), cap, init, d_out.data_ptr())
        m.sync()
        got = d_out.cpu().numpy()
        for p in range(len(pq)):
            nq, nt = counts[pq[p]], counts[pt[p]]
            want = ob.bruteforce_best2(desc[pq[p], :nq], desc[pt[p], :nt], init)
            _same(got[p, :nq].view(gpu_lib.BEST2_DTYPE).reshape(-1), want, f"{kernel} pair {p} ({nq} x {nt}) init={init}")
            assert (got[p, nq:] == -7).all(), "rows beyond the query count are not written"


def test_match_consecutive_frames(gpu_lib, ob, synth):
    """End to end on real descriptors: frame k against frame k-1 (SURVEY 8d match workload)."""
    ext = gpu_lib.OrbExtractor()
    k0, d0 = ext.extract(synth.frame(5, 10))
    k1, d1 = ext.extract(synth.frame(5, 11))
    m = gpu_lib.OrbMatcher()
    got = m.bruteforce_best2(d1, d0)
    _same(got, ob.bruteforce_best2(d1, d0), "consecutive-frame matches")
    good = (got["best_dist"] <= 50) & (got["best_dist"] < 0.6 * got["second_dist"])
    assert good.sum() > 100  # the scene moved by (2, 1) px: most corners re-match


def _lab_scene(rng, h, w):
    base = np.kron(rng.integers(0, 256, ((h + 23) // 24, (w + 23) // 24, 3)), np.ones((24, 24, 1)))[:h, :w]
    lab = np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    depth = (5000 * (1.5 + 0.5 * np.sin(np.arange(w) / 40.0))[None, :] * np.ones((h, 1))).astype(np.uint16)
    depth[rng.random((h, w)) < 0.1] = 0
    return lab, depth


@pytest.mark.parametrize("w,h,length", [(640, 480, 5), (322, 241, 5), (203, 157, 4), (64, 48, 7)])
def test_slic_vs_oracle(gpu_lib, ob, w, h, length):
    """8f-2 (SLIC half): cluster::SLIC from the Lab image on -- label map (float64) and centres, bit for bit."""
    rng = np.random.default_rng(w + h)
    lab, depth = _lab_scene(rng, h, w)
    sl = gpu_lib.Slic(max_width=w, max_height=h)
    for iters in (0, 2, 5):
        lg, cg = sl.run(lab, depth, length, 10, iters)
        lo, co = ob.slic(lab, depth, length, 10, iters)
        _same(lg, lo, f"label map after {iters} iterations")
        _same(cg, co, f"centres after {iters} iterations")
    assert gpu_lib.Slic.center_count(w, h, length)[0] == len(co)
    # flat image: every distance ties except for the spatial term; uniform depth
    flat = np.full((h, w, 3), 128, np.uint8)
    lg, cg = sl.run(flat, np.full((h, w), 4000, np.uint16), length, 10, 5)
    lo, co = ob.slic(flat, np.full((h, w), 4000, np.uint16), length, 10, 5)
    _same(lg, lo, "flat label map")
    _same(cg, co, "flat centres")


def test_slic_batch_device(gpu_lib, ob):
    import torch
    n, w, h = 3, 320, 240
    rng = np.random.default_rng(5)
    scenes = [_lab_scene(rng, h, w) for _ in range(n)]
    d_lab = torch.from_numpy(np.stack([s[0] for s in scenes])).cuda()
    d_depth = torch.from_numpy(np.stack([s[1] for s in scenes]).view(np.int16)).cuda()
    nc = gpu_lib.Slic.center_count(w, h, 5)[0]
    d_labels = torch.zeros((n, h, w), dtype=torch.float64, device="cuda")
    d_cent = torch.zeros((n, nc, 8), dtype=torch.int32, device="cuda")
    sl = gpu_lib.Slic(max_width=w, max_height=h, max_batch=n)
    torch.cuda.synchronize()
    sl.run_batch_device(d_lab.data_ptr(), d_depth.data_ptr(), w, h, n, d_labels.data_ptr(), d_cent.data_ptr())
    sl.sync()
    torch.cuda.synchronize()
    for f in range(n):
        lo, co = ob.slic(*scenes[f])
        _same(d_labels[f].cpu().numpy(), lo, f"frame {f} labels")
        _same(np.frombuffer(d_cent[f].cpu().numpy().tobytes(), ob.SLIC_CENTER_DTYPE), co, f"frame {f} centres")
