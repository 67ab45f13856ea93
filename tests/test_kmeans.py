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
