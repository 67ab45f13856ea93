"""The bench.py output contract (one JSON line with the driver's keys, the roofline and cpu_baseline objects).

Named test_zz_*: a harness test must sort AFTER every parity test, so that under `pytest -x` a harness assertion can never hide them
(tests/conftest.py also orders the collection: parity first, this file last)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _common(d, steps, warmup):
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "frames/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # whichever byte-moving stage was slowest at this toy batch (timing-dependent: nothing below depends on WHICH), its figures
    # must be self-consistent; a zero-byte stage (the latency-bound quad-tree) may never be the roofline kernel
    assert r["kernel"] != "octree" and r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0
    assert r["achieved"] is not None and r["achieved"] > 0 and ("traffic" in r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.02 * r["achieved"] + 0.02
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 2 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0
    assert d["pipeline_roofline"]["frac"] > 0 and d["gated_match"]["queries_per_s"] > 0
    rows = d["digest_per_rank"]["rows"]
    assert len(rows) == 1 and len(rows[0]) % 3 == 0 and rows[0][0] > 100  # (keypoints, descriptor CRC32, matches) per frame
    assert d["oracle_checked_frames"] == 2


@pytest.mark.gpu
def test_default_run_is_the_baseline_metric_with_the_mask(gpu_lib):
    """`python bench.py --gpus 1` times BASELINE.json configs[2] (mask on) and carries the mask-off leg beside it."""
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--leg-steps", "3", "--leg-batch", "64", "--cpu-frames", "3",
              "--cpu-cores", "2", "--check"])
    _common(d, 2, 1)
    assert "+mask" in d["metric"] and "configs[2]" in d["config"]["workload"]
    ious = d["mask_checked_iou_vs_one_frame_pass"]   # --check on a mask-on run: lane 0's first and last mask against a one-frame eager pass
    assert len(ious) == 2 and min(ious) >= 1 - 1e-3
    assert "fp32" in d["dtype"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 0.01      # value = frames / time
    leg = d["extract_match_leg"]
    assert "mask off" in leg["metric"] and leg["value"] > d["value"] and leg["frames_per_step_per_gpu"] == 64 and leg["steps"] == 3
    assert abs(leg["value"] - 64 * 3 / (leg["ms_per_step"] * 3 * 1e-3)) / leg["value"] < 0.01
    m = d["roofline_mask"]
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and m["peak"] == 157.3 and 0 < m["frac"] < 1
    assert 1.0e11 < m["flops_per_frame"] < 1.4e11  # YOLACT-R50 at 550 x 550: 59 G multiply-accumulates, counted as direct convolutions
    assert 0.4 * m["flops_per_frame"] < m["flops_per_frame_executed"] <= m["flops_per_frame"]  # Winograd layers at 1 / 3 (few of them at 8 frames per launch)
    assert abs(m["achieved_direct_equivalent"] / m["achieved"] - m["flops_per_frame"] / m["flops_per_frame_executed"]) < 0.01
    assert d["stage_ms_per_launch"]["mask_pass"] > 0
    lat = d["drop_in_latency"]  # the extra key: one frame per call, host buffers: the C++ drop-in classes themselves, and the ctypes mirror beside them
    assert "error" not in lat
    mir, cxx = lat["ctypes_mirror"], lat["cxx"]
    assert 0 < mir["orb_extract_4arg_ms"] < mir["orb_extract_4arg_and_nxn_match_ms"] < 50 and 0 < mir["mask_pass_one_frame_graph_ms"] < 100
    assert mir["mask_pass_one_frame_graph_ms"] < mir["mask_detect_gate_describe_ms"] <= mir["mask_detect_gate_describe_nxn_match_ms"] * 1.05
    assert "error" not in cxx, cxx
    assert cxx["eval_image_returned_false"] == 0 and cxx["keypoints_last_frame"] > 500 and cxx["matches_last_frame"] > 100
    parts = cxx["eval_image_ms"] + cxx["detect_ms"] + cxx["moving_keypoints_ms"] + cxx["process_desp_ms"] + cxx["search_by_projection_ms"]
    assert abs(parts - cxx["frame_ms"]) < 0.05 * cxx["frame_ms"] + 0.05 and abs(cxx["frames_per_s_one_stream"] - 1e3 / cxx["frame_ms"]) < 1.0
    assert 0.5 < cxx["up_to_descriptors_vs_ctypes_mirror"] < 1.5   # (the round's target is <= 1.10 on an idle box; a test must not be that tight)
    k = m["dominant_kernel"]  # the project's convolution kernel on the largest layer, live: Winograd F(2 x 4) at 8 frames per launch (604 x 4 work-groups)
    assert "k_winograd24_conv" in k["kernel"] and k["bound"] == "mfma" and k["peak"] == 157.3 and 0.2 < k["frac"] < 1.0
    assert abs(k["achieved"] - k["flops_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e12) / k["achieved"] < 0.01
    assert k["flops_per_launch"] == 2 * 24 * k["frames_per_launch"] * 69 * 35 * 256 * 256                     # as executed (tiles of 2 x 4 outputs)
    assert k["direct_convolution_flops_per_launch"] == 2 * k["frames_per_launch"] * 138 * 138 * 256 * 9 * 256   # 2.96 x that


@pytest.mark.gpu
def test_mask_off_config(gpu_lib):
    d = _run(["--gpus", "1", "--config", "c2", "--steps", "4", "--warmup", "1", "--batch", "64", "--cpu-frames", "3", "--cpu-cores", "2", "--check"])
    _common(d, 4, 1)
    assert "mask off" in d["metric"] and d["dtype"] == "u8" and "configs[1]" in d["config"]["workload"]
    assert abs(d["value"] - 64 * 4 / (d["ms_per_step"] * 4 * 1e-3)) / d["value"] < 0.01
    assert "extract_match_leg" not in d and d["stage_ms_per_launch"]["fast"] > 0
