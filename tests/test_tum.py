"""The real-sequence input path of bench.py (`--tum-root` / AMOS_TUM_ROOT): associations file as the reference's driver reads it
(Examples/RGB-D/rgbd_tum.cc:182-210), PNG decode, BGR order, gray conversion.  No TUM sequence exists here, so a three-frame fake
sequence is built from the reference's own two sample frames (tests/golden/ref_data)."""
import importlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref_data")
NAMES = ("1341846313.553992", "1341846313.592026", "1341846313.654184")  # timestamps in TUM's style; the first is the reference's file name


@pytest.fixture(scope="module")
def fake_root(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("tum")
    seq = root / "rgbd_dataset_freiburg3_walking_xyz"
    (seq / "rgb").mkdir(parents=True)
    (seq / "depth").mkdir()
    srcs = ("1341846313.553992.png", "122_rgb.png", "1341846313.553992.png")
    lines = []
    for k, (name, src) in enumerate(zip(NAMES, srcs)):
        if k < 2:
            shutil.copy(os.path.join(REF, src), seq / "rgb" / (name + ".png"))
        else:  # third frame: the first one shifted by 4 px (a moving camera)
            im = np.asarray(Image.open(os.path.join(REF, src)).convert("RGB"))
            Image.fromarray(np.roll(im, 4, axis=1)).save(seq / "rgb" / (name + ".png"))
        yy, xx = np.mgrid[0:480, 0:640]
        depth = (5000 * (1.5 + 0.5 * np.sin((xx + 3 * yy + 40 * k) / 97.0))).astype(np.uint16)
        Image.fromarray(depth).save(seq / "depth" / (name + ".png"))
        lines.append(f"{name} rgb/{name}.png {float(name) + 0.01:.6f} depth/{name}.png")
    (seq / "associations.txt").write_text(lines[0] + "\n\n" + lines[1] + "\n" + lines[2] + "\n")  # an empty line, as LoadImages tolerates
    return str(root)


@pytest.fixture(scope="module")
def tum(pkg):
    return importlib.import_module("amos_slam_amd.tum")


def test_associations_as_the_reference_reads_them(tum, fake_root):
    rows = tum.load_associations(os.path.join(fake_root, "rgbd_dataset_freiburg3_walking_xyz", "associations.txt"))
    assert [r[0] for r in rows] == [float(n) for n in NAMES]           # the FIRST timestamp of a line is the frame's
    assert rows[1][1] == f"rgb/{NAMES[1]}.png" and rows[1][2] == f"depth/{NAMES[1]}.png"
    ref_file = "/root/reference/Examples/RGB-D/associations/fr1_xyz.txt"  # the reference's own file, where the reference tree exists
    if os.path.exists(ref_file):
        ref_rows = tum.load_associations(ref_file)
        assert len(ref_rows) == 792 and ref_rows[0] == (1305031102.175304, "rgb/1305031102.175304.png", "depth/1305031102.160407.png")
    with pytest.raises(ValueError):
        bad = os.path.join(fake_root, "bad.txt")
        open(bad, "w").write("1.0 rgb/a.png\n")
        tum.load_associations(bad)


def test_sequence_frames(tum, fake_root, ob):
    from PIL import Image
    seq = tum.load_sequence(fake_root, "fr3_walking_xyz", 5, with_depth=True)   # short name -> directory; 5 frames of 3 wrap around
    assert seq["name"] == "rgbd_dataset_freiburg3_walking_xyz" and seq["frames_in_sequence"] == 3 and seq["wrapped"]
    assert seq["bgr"].shape == (5, 480, 640, 3) and seq["bgr"].dtype == np.uint8 and seq["depth"].shape == (5, 480, 640) and seq["depth"].dtype == np.uint16
    assert np.array_equal(seq["bgr"][3], seq["bgr"][0]) and np.array_equal(seq["bgr"][4], seq["bgr"][1]) and seq["timestamps"][3] == float(NAMES[0])
    rgb = np.asarray(Image.open(os.path.join(REF, "1341846313.553992.png")).convert("RGB"))
    assert np.array_equal(seq["bgr"][0][:, :, ::-1], rgb)                         # B, G, R order like cv::imread
    assert 5000 <= int(seq["depth"][0].min()) and int(seq["depth"][0].max()) <= 10000
    # the gray frame the extractor sees: the TUM yamls set Camera.RGB: 1, so Tracking.cc:311 applies CV_RGB2GRAY to the imread (B, G, R)
    # buffer == the oracle's cvtColor with rgb_order (the default here); Camera.RGB: 0 is the plain BGR2GRAY
    assert tum.CAMERA_RGB == 1
    assert np.array_equal(tum.bgr_to_gray(seq["bgr"][1]), ob.color_to_gray(seq["bgr"][1], rgb_order=True))
    assert np.array_equal(tum.bgr_to_gray(seq["bgr"][1], rgb_flag=0), ob.color_to_gray(seq["bgr"][1], rgb_order=False))
    assert not np.array_equal(tum.bgr_to_gray(seq["bgr"][1], 1), tum.bgr_to_gray(seq["bgr"][1], 0))
    px = np.array([[[200, 10, 30]]], np.uint8)   # B = 200: weighted 0.299 under Camera.RGB = 1, 0.114 under 0
    assert int(tum.bgr_to_gray(px, 1)[0, 0]) == (200 * 9798 + 10 * 19235 + 30 * 3735 + 16384) >> 15 == 69
    assert int(tum.bgr_to_gray(px, 0)[0, 0]) == (200 * 3735 + 10 * 19235 + 30 * 9798 + 16384) >> 15 == 38
    later = tum.load_sequence(fake_root, "rgbd_dataset_freiburg3_walking_xyz", 2, start=1)
    assert np.array_equal(later["bgr"][0], seq["bgr"][1]) and later["depth"] is None
    with pytest.raises(FileNotFoundError):
        tum.load_sequence(fake_root, "fr1_xyz", 1)


def test_bench_cpu_leg_reads_the_sequence(tum, fake_root, ob, synth):
    """bench.py's CPU-baseline leg on the sequence (the GPU legs use the same loader: tests/test_gpu_rccl.py)."""
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS["c2"]
    src = ("tum", fake_root, "fr3_walking_xyz", None, 0, 1)
    frames = bench._source_frames(synth, cfg, src, 3)
    assert len(frames) == 3 and frames[0].shape == (480, 640) and frames[0].dtype == np.uint8
    one = bench.cpu_baseline_one_thread(synth, cfg, 2, src)
    assert one["value"] > 0 and "TUM" in one["sample"]
    granted, quota = bench.granted_cores()
    assert granted >= 1 and (quota is None or quota > 0) and 1 <= bench.all_cores_worker_count() <= granted


@pytest.mark.gpu
def test_bench_on_a_tum_sequence(gpu_lib, fake_root):
    """`bench.py --config c2 --tum-root ...`: real frames resident in HBM, checked against the oracle, `data` says which sequence."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "2", "--warmup", "1", "--batch", "8", "--cpu-frames", "2",
                          "--cpu-cores", "2", "--check", "--tum-root", fake_root], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["data"].startswith("tum:rgbd_dataset_freiburg3_walking_xyz") and "wrapped" in d["data"] and "Camera.RGB=1: CV_RGB2GRAY" in d["data"]
    assert d["oracle_checked_frames"] == 2 and d["value"] > 0 and d["config"]["mean_keypoints_per_frame"] > 500
    assert "TUM" in d["cpu_baseline"]["sample"]
