"""CPU: every frame size a handle admits stays inside the handle's allocations (a
1910x1080 frame on a 1920x1080 handle has MORE FAST cells than the largest frame).  Uses the host-only
amos_orb_geometry_probe -- no device is touched."""
import numpy as np
import pytest


def _sweep(pkg, max_w, max_h, sizes, **params):
    worst = np.zeros(6)
    for (w, h) in sizes:
        rc, need, cap = pkg.geometry_probe(max_w, max_h, w, h, **params)
        if rc == -1:
            continue  # a level without a FAST cell: the reference cannot process such a frame either
        assert rc == 0, f"{w}x{h} on a {max_w}x{max_h} handle: need {need.tolist()} cap {cap.tolist()}"
        assert (need <= cap).all()
        worst = np.maximum(worst, need / np.maximum(cap, 1))
    return worst


def test_the_advisors_counter_example(pkg):
    rc, need, cap = pkg.geometry_probe(1920, 1080, 1910, 1080, n_features=4000, n_levels=8)
    rc0, need0, _ = pkg.geometry_probe(1920, 1080, 1920, 1080, n_features=4000, n_levels=8)
    assert rc == 0 and rc0 == 0
    assert need[0] > need0[0], "the smaller frame has more FAST cells than the largest one"
    assert need[0] <= cap[0]


def test_every_width_and_height_of_vga(pkg):
    sizes = [(w, 480) for w in range(200, 641)] + [(640, h) for h in range(200, 481)] + [(w, w * 3 // 4) for w in range(240, 641, 7)]
    worst = _sweep(pkg, 640, 480, sizes)
    assert worst.max() <= 1.0


def test_hd_handle_sampled(pkg):
    rng = np.random.default_rng(5)
    sizes = [(int(w), int(h)) for w, h in zip(rng.integers(700, 1921, 400), rng.integers(500, 1081, 400))]
    sizes += [(w, 1080) for w in range(1850, 1921)] + [(1920, h) for h in range(1000, 1081)]
    worst = _sweep(pkg, 1920, 1080, sizes, n_features=4000, n_levels=12)
    assert worst.max() <= 1.0


@pytest.mark.parametrize("scale,levels", [(1.1, 8), (1.5, 5), (2.0, 4)])
def test_other_scale_factors(pkg, scale, levels):
    rng = np.random.default_rng(int(scale * 10))
    sizes = [(int(w), int(h)) for w, h in zip(rng.integers(400, 753, 150), rng.integers(400, 481, 150))]
    _sweep(pkg, 752, 480, sizes, scale_factor=scale, n_levels=levels)
