"""cluster::randCent + cluster::kmeans (src/cluster.cc:353-460) with the seeded generator: the oracle against an independent
Python restatement (CPU) and the HIP kernel against the oracle (GPU)."""
import math

import numpy as np
import pytest


def _centres(ob, rng, n, width=640, height=480, zero_depth=0.1):
    c = np.zeros(n, ob.SLIC_CENTER_DTYPE)
    c["x"], c["y"] = rng.integers(0, width, n), rng.integers(0, height, n)
    # a few depth layers + noise, some invalid (0) depths
    layer = rng.integers(0, 4, n)
    c["D"] = np.where(rng.random(n) < zero_depth, 0, 5000 + 6000 * layer + rng.integers(-300, 300, n))
    c["L"], c["A"], c["B"] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    c["label"] = np.arange(1, n + 1)
    c["id"] = -5
    return c


def _python_kmeans(c, k, seed, max_iter=1000):
    """Written from the reference's text (cluster.cc:353-460) with the documented definitions of its undefined behaviours."""
    n = len(c)
    state = seed & 0xffffffff

    def draw():
        nonlocal state
        state = (state * 1103515245 + 12345) & 0xffffffff
        idx = (state & 0x7fffffff) % n + 1
        return 0 if idx >= n else idx

    cent = []
    for _ in range(k):
        idx, tries = draw(), 0
        while c["D"][idx] <= 0 and tries < 4 * n:
            idx, tries = draw(), tries + 1
        cent.append([int(c["x"][idx]), int(c["y"][idx]), int(c["D"][idx])])
    assign = [-1] * n
    xs, ys, ds = c["x"].tolist(), c["y"].tolist(), c["D"].tolist()
    passes = 0
    while True:
        passes += 1
        assert passes <= max_iter
        changed = False
        for i in range(n):
            best, best_d = -1, 2147483647.0
            for j, (cx, cy, cd) in enumerate(cent):
                dist = math.sqrt(float((cx - xs[i]) ** 2 + (cy - ys[i]) ** 2)) / 800.0 + abs(ds[i] - cd) / 20000.0
                if dist < best_d:
                    best, best_d = j, dist
            if assign[i] != best:
                changed, assign[i] = True, best
        for j in range(k):
            members = [i for i in range(n) if assign[i] == j]
            m = len(members)
            cent[j] = [sum(xs[i] for i in members) // m, sum(ys[i] for i in members) // m, sum(ds[i] for i in members) // m] if m else [0, 0, 0]
        if not changed:
            break
    return np.array(assign, np.int32), passes


@pytest.mark.parametrize("n,k,seed", [(300, 5, 1), (777, 15, 42), (50, 3, 7)])
def test_oracle_kmeans_matches_python_restatement(ob, n, k, seed):
    rng = np.random.default_rng(n)
    c = _centres(ob, rng, n)
    got, passes = ob.kmeans(c, k, seed)
    want, want_passes = _python_kmeans(c, k, seed)
    assert passes == want_passes and np.array_equal(got["id"], want)
    assert set(np.unique(got["id"]).tolist()) <= set(range(k))
    for f in ("x", "y", "L", "A", "B", "D", "label"):
        assert np.array_equal(got[f], c[f])  # only the ids are written


def test_oracle_kmeans_definitions_of_the_undefined_cases(ob):
    rng = np.random.default_rng(3)
    c = _centres(ob, rng, 64, zero_depth=1.0)  # every depth zero: the reference would redraw forever
    got, passes = ob.kmeans(c, 4, 9)
    assert passes >= 1 and (got["id"] >= 0).all()
    c = _centres(ob, rng, 200)
    a, _ = ob.kmeans(c, 6, 1)
    b, _ = ob.kmeans(c, 6, 2)
    a2, _ = ob.kmeans(c, 6, 1)
    assert np.array_equal(a["id"], a2["id"]) and not np.array_equal(a["id"], b["id"])  # a function of the seed
    one, p1 = ob.kmeans(c, 1, 5)
    assert (one["id"] == 0).all() and p1 == 2  # k = 1: everything joins cluster 0, the second pass changes nothing
    assert ob.kmeans(c, 6, 1, max_iter=1)[1] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,seed", [(12288, 15, 1), (12288, 15, 99), (3072, 8, 5), (1000, 64, 3), (17, 2, 4)])
def test_gpu_kmeans_vs_oracle(gpu_lib, ob, n, k, seed):
    rng = np.random.default_rng(1000 + n + k)
    w, h = (640, 480)
    c = _centres(ob, rng, n, w, h)
    if n == 12288:  # the SLIC grid of a 640 x 480 frame: 128 x 96 centres, depth in smooth blobs
        gx, gy = np.meshgrid(np.arange(128) * 5 + 2, np.arange(96) * 5 + 2)
        c["x"], c["y"] = gx.ravel(), gy.ravel()
        c["D"] = (8000 + 6000 * np.sin(gx.ravel() / 90.0) * np.cos(gy.ravel() / 70.0) + rng.integers(0, 200, n)).astype(np.int32)
        c["D"][rng.random(n) < 0.05] = 0
    s = gpu_lib.Slic()
    got, passes = s.kmeans(c, k, seed)
    want, want_passes = ob.kmeans(c, k, seed)
    assert passes == want_passes and np.array_equal(got["id"], want["id"])
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_gpu_kmeans_batch_after_slic(gpu_lib, ob):
    """SLIC then k-means for a resident batch (the cluster constructor, cluster.cc:9-43, from the Lab image on)."""
    import torch
    rng = np.random.default_rng(8)
    n_frames, h, w = 3, 240, 320
    lab = rng.integers(0, 256, (n_frames, h, w, 3), dtype=np.uint8)
    lab[:, :, : w // 2] //= 3
    depth = (6000 + 4000 * (np.arange(w)[None, None, :] > w // 3) + rng.integers(0, 100, (n_frames, h, w))).astype(np.uint16)
    s = gpu_lib.Slic(max_width=w, max_height=h, max_batch=n_frames)
    n, _, _ = s.center_count(w, h)
    d_lab, d_depth = torch.from_numpy(lab).cuda(), torch.from_numpy(depth).cuda()
    d_labels = torch.zeros((n_frames, h, w), dtype=torch.float64, device="cuda")
    d_centers = torch.zeros((n_frames, n, 8), dtype=torch.int32, device="cuda")
    d_passes = torch.zeros(n_frames, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    s.run_batch_device(d_lab.data_ptr(), d_depth.data_ptr(), w, h, n_frames, d_labels.data_ptr(), d_centers.data_ptr())
    s.kmeans_batch_device(d_centers.data_ptr(), n, n_frames, k=15, seed=11, d_passes=d_passes.data_ptr())
    s.sync()
    torch.cuda.synchronize()
    got = d_centers.cpu().numpy()
    for f in range(n_frames):
        _, centres = ob.slic(lab[f], depth[f])
        want, passes = ob.kmeans(centres, 15, 11)
        assert int(d_passes[f]) == passes
        assert np.array_equal(got[f].reshape(-1), want.view(np.int32).reshape(-1))


def _python_lab(bgr):
    """OpenCV 4.5 RGB2Lab_b written from its published form, in numpy (independent of the C oracle's code)."""
    x = np.arange(256, dtype=np.float32) / np.float32(255)
    lin = np.where(x <= np.float32(0.04045), x.astype(np.float64) / 12.92, ((x.astype(np.float64) + 0.055) / 1.055) ** 2.4)
    gamma = np.rint(255.0 * 8.0 * lin).astype(np.int64)
    y = np.arange(3072, dtype=np.float32) / np.float32(255 * 8)
    cb = np.rint(32768.0 * np.where(y < np.float32(0.008856), y.astype(np.float64) * 7.787 + 0.13793103448275862, np.cbrt(y.astype(np.float64)))).astype(np.int64)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    coef = np.rint(4096.0 * m / np.array([0.950456, 1.0, 1.088754])[:, None]).astype(np.int64)
    B, G, R = (gamma[bgr[..., c]] for c in range(3))
    f = [cb[(R * coef[i, 0] + G * coef[i, 1] + B * coef[i, 2] + 2048) >> 12] for i in range(3)]
    L = (296 * f[1] - 1336934 + 16384) >> 15
    a = (500 * (f[0] - f[1]) + 128 * 32768 + 16384) >> 15
    b = (200 * (f[1] - f[2]) + 128 * 32768 + 16384) >> 15
    return np.clip(np.stack([L, a, b], -1), 0, 255).astype(np.uint8)


def test_oracle_bgr2lab_known_values_and_numpy(ob):
    """cv::cvtColor(COLOR_BGR2Lab) 8-bit: OpenCV's documented values for the primaries / grays, and an independent numpy form."""
    src = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]], np.uint8)  # BGR
    assert ob.bgr_to_lab(src).tolist() == [[0, 128, 128], [255, 128, 128], [82, 207, 20], [224, 42, 211], [136, 208, 195], [137, 128, 128]]
    assert ob.bgr_to_lab(src[:, ::-1].copy(), rgb_order=True).tolist() == ob.bgr_to_lab(src).tolist()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    assert np.array_equal(ob.bgr_to_lab(img), _python_lab(img))


@pytest.mark.gpu
def test_gpu_bgr2lab_then_slic_then_kmeans(gpu_lib, ob):
    """The cluster constructor (cluster.cc:9-43) on the device from the BGR frame: BGR2Lab -> SLIC -> k-means, each vs the oracle."""
    import torch
    rng = np.random.default_rng(12)
    h, w = 240, 320
    bgr = np.kron(rng.integers(0, 256, (h // 16, w // 16, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    bgr = np.clip(bgr.astype(np.int64) + rng.integers(-20, 21, bgr.shape), 0, 255).astype(np.uint8)
    depth = (6000 + 3000 * (np.arange(w)[None, :] > w // 2) + rng.integers(0, 80, (h, w))).astype(np.uint16)
    s = gpu_lib.Slic(max_width=w, max_height=h, max_batch=1)
    n, _, _ = s.center_count(w, h)
    d_bgr, d_depth = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
    d_lab = torch.zeros_like(d_bgr)
    d_labels = torch.zeros((1, h, w), dtype=torch.float64, device="cuda")
    d_centers = torch.zeros((1, n, 8), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    s.bgr2lab_batch_device(d_bgr.data_ptr(), h * w, d_lab.data_ptr())
    s.run_batch_device(d_lab.data_ptr(), d_depth.data_ptr(), w, h, 1, d_labels.data_ptr(), d_centers.data_ptr())
    s.kmeans_batch_device(d_centers.data_ptr(), n, 1, k=15, seed=3)
    s.sync()
    torch.cuda.synchronize()
    lab = ob.bgr_to_lab(bgr)
    assert np.array_equal(d_lab.cpu().numpy(), lab)
    labels, centres = ob.slic(lab, depth)
    want, _ = ob.kmeans(centres, 15, 3)
    assert np.array_equal(d_labels.cpu().numpy()[0], labels)
    assert np.array_equal(d_centers.cpu().numpy()[0].reshape(-1), want.view(np.int32).reshape(-1))
    # all 2^24 colours: the kernel's table path against the oracle's
    allc = np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 5), indexing="ij"), -1).reshape(-1, 3).astype(np.uint8)
    d_all = torch.from_numpy(allc).cuda()
    d_out = torch.zeros_like(d_all)
    s.bgr2lab_batch_device(d_all.data_ptr(), len(allc), d_out.data_ptr())
    s.sync()
    assert np.array_equal(d_out.cpu().numpy(), ob.bgr_to_lab(allc))
