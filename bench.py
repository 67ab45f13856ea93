#!/usr/bin/env python3
"""bench.py -- front-end frames/sec (ORB extract + match + mask) on MI355X.

Default run = BASELINE.json configs[2], the metric as written: a step is one pass of the whole front-end
over one batch of synthetic 640x480 frames already resident in HBM --
    ORB detect (pyramid, FAST, quad-tree, orientation)         [HIP, this repo]
    YOLACT-R50 person mask, fp32                                [PyTorch-ROCm / MIOpen]
    mask gate (31x31 closing + keypoint removal)                [HIP]
    blur + rBRIEF descriptors                                   [HIP]
    N x N best-2 Hamming match of frame k against frame k-1     [HIP]
(Tracking.cc:366 -> Frame.cc:491-496,633 -> ORBmatcher).  `value` is that.  The same JSON line carries the
mask-off sub-leg (`extract_match_leg`, BASELINE configs[1]: extract + match only, the part that is hand-written
HIP), its HBM `roofline` for the dominant kernel, `roofline_mask` (achieved fp32 TFLOP/s of the network against
the MFMA fp32 peak) and `cpu_baseline` (the CPU oracle on all granted host cores, plus one thread).

One process per GPU; rank r works on its own synthetic stream r (frames are independent: no data-path
collective, weak scaling); RCCL is used only for the barrier / max-over-ranks time and the final gather of the
per-frame result digests.  `python bench.py --gpus N` with N > 1 and no launcher environment starts
`python -m torch.distributed.run` itself (before anything touches a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

import numpy as np

# HIP multiplexes streams onto this many hardware queues in creation order; the lanes below create their
# streams as main0, side0, main1, side1, ... so that with 4 queues the main streams of lanes (0, 2) and (1, 3)
# and the side streams of (0, 2) and (1, 3) share a queue each -- measured best.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3   # dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32), same guide; gfx950 has no TF32/xf32

METRIC = "front-end frames/sec (ORB extract+match+mask) at 640x480"

CONFIGS = {
    # BASELINE.json configs[1]: single MI355X, extract + brute-force match, mask disabled
    "c2": dict(width=640, height=480, n_features=1000, n_levels=8, default_batch=512, default_streams=4,
               label="configs[1]: 640x480 L8 N1000 extract+match, mask off"),
    # BASELINE.json configs[2]: full front-end incl. the YOLACT mask (network on PyTorch-ROCm, random weights
    # with a biased class head so that ~100 detections exercise the whole post-processing chain)
    # (two lanes of 64 frames per network forward: 1 496 frames/s against 1 473 for four lanes of 32 -- the larger launches fill the chip's
    # last round of work-groups better; more than 64 frames per forward would exceed the 2 GiB a Winograd launch's input may have)
    "c3": dict(width=640, height=480, n_features=1000, n_levels=8, mask=True, default_batch=128, default_streams=2,
               label="configs[2]: 640x480 L8 N1000 full front-end: YOLACT-R50 fp32 mask + ORB detect + gate + describe + match"),
    # BASELINE.json configs[4]: synthetic HD stream
    "c5": dict(width=1920, height=1080, n_features=4000, n_levels=12, default_batch=128, default_streams=4,
               label="configs[4]: 1920x1080 L12 N4000 extract+match"),
}


def algorithmic_bytes(level_w, level_h, n_kp, width, height):
    """ALGORITHMIC bytes per frame of every kernel (each mandatory pass reads its
    input once and writes its output once)."""
    wh = [int(w) * int(h) for w, h in zip(level_w, level_h)]
    padded = [(int(w) + 38) * (int(h) + 38) for w, h in zip(level_w, level_h)]
    return {
        "import": width * height + padded[0],                  # level 0: frame read, padded plane written
        "pyramid": sum(wh[:-1]) + sum(padded[1:]),             # levels >= 1: level l-1 read, padded level l written
        "fast": sum(wh),
        "octree": 0,  # selection over the candidate list: negligible bytes, latency-bound
        "orient": n_kp * 749,
        "blur": 2 * sum(wh),
        "describe": n_kp * (512 + 32),
        "match": 2 * n_kp * 32 + n_kp * 16,
    }


# --------------------------------------------------------------------------------------------- CPU baseline
def _source_frames(synth, cfg, source, n):
    """n gray frames of a frame source: ("synth", stream) or ("tum", root, sequence, associations, first frame, Camera.RGB flag)."""
    if source[0] == "tum":
        import importlib
        tum = importlib.import_module("amos_slam_amd.tum")
        seq = tum.load_sequence(source[1], source[2], n, associations=source[3], start=source[4])
        return [tum.bgr_to_gray(f, source[5] if len(source) > 5 else tum.CAMERA_RGB) for f in seq["bgr"]]
    return [synth.frame(source[1], k, cfg["height"], cfg["width"]) for k in range(n)]


def _oracle_stream_run(ob, synth, cfg, source, n, barrier=None):
    orc = ob.Oracle(n_features=cfg["n_features"], n_levels=cfg["n_levels"])
    frames = _source_frames(synth, cfg, source, n + 1)
    if barrier is not None:
        barrier.wait()
    t0 = time.perf_counter()
    prev = orc.extract(frames[0])[1]
    for k in range(1, n + 1):
        _, desc = orc.extract(frames[k])
        ob.bruteforce_best2(desc, prev)
        prev = desc
    return time.perf_counter() - t0


def cpu_baseline_one_thread(synth, cfg, n_sample, source=("synth", 0)):
    """The oracle (CPU restatement of src/ORBextractor.cc + ORBmatcher.cc, 1 thread) on a bounded
    sample of the same workload.  Checker only: it is never the thing shipped or measured as GPU."""
    import oracle_binding as ob
    dt = _oracle_stream_run(ob, synth, cfg, source, n_sample)
    what = "the same synthetic stream" if source[0] == "synth" else f"the same TUM sequence ({source[2]})"
    return {"value": round(n_sample / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_sample} frames of {what}: oracle extract + N x N best-2 match, 1 thread"}


def _cpu_worker(root, cfg, source, n, barrier, q):
    """One worker process of the all-cores CPU baseline: frames of its own synthetic stream (or its own stretch of the sequence)."""
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    synth = importlib.import_module("amos_slam_amd.synth")
    import oracle_binding as ob
    q.put(_oracle_stream_run(ob, synth, cfg, source, n, barrier))


def granted_cores():
    """(cores this process may run on, cgroup CPU quota in cores or None): the affinity mask, and cpu.max of cgroup v2 /
    cfs_quota_us of v1 when a quota is set (a quota below the mask means that many cores' worth of time, whatever the mask shows)."""
    granted = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return granted, quota


def all_cores_worker_count():
    """Worker processes of the all-cores CPU baseline: every granted core (north_star: "all host cores stated"), held back only by a
    cgroup quota (more runnable processes than quota just take turns) and by memory (~0.25 GiB per worker: interpreter, numpy, frames)."""
    granted, quota = granted_cores()
    workers = granted if quota is None else max(1, min(granted, int(quota + 0.5)))
    try:
        avail_kb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable:"))
        workers = max(1, min(workers, int(avail_kb / (256 * 1024) * 0.5)))
    except Exception:
        pass
    return workers


def cpu_baseline_all_cores(cfg, workers, n_per_worker, tum_source=None):
    """The same oracle on `workers` processes, one frame stream per process, started together (the reference
    extractor is single-threaded per frame, so all cores = one frame per core)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(workers), ctx.Queue()
    small = {k: cfg[k] for k in ("n_features", "n_levels", "height", "width")}
    sources = [(("synth", 100 + w) if tum_source is None else tum_source[:4] + (tum_source[4] + w * n_per_worker,) + tum_source[5:]) for w in range(workers)]
    procs = [ctx.Process(target=_cpu_worker, args=(ROOT, small, sources[w], n_per_worker, barrier, q)) for w in range(workers)]
    for p in procs:
        p.start()
    times = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    wall = max(times)  # all workers leave the barrier together: the slowest one is the wall time
    what = f"{workers} synthetic streams" if tum_source is None else f"consecutive stretches of the TUM sequence {tum_source[2]}"
    return {"value": round(workers * n_per_worker / wall, 2), "unit": "frames/s", "cores": workers, "kind": "port",
            "per_process_frames_per_s": [round(min(n_per_worker / t for t in times), 2), round(max(n_per_worker / t for t in times), 2)],
            "sample": f"{workers} processes x {n_per_worker} frames of {what}, started together; "
                      f"oracle extract + N x N best-2 match per frame"}


# --------------------------------------------------------------------------------------------- the C++ drop-in path, per frame
def cxx_frame_latency(root, gray_frames, bgr_frames, iters, with_mask, weights=""):
    """amos_host_frame_latency of tests/host/libamos_host_test.so (tests/host/host_capi.cc): per frame, through the C++ classes of
    amos-slam_amd/host/libamos_host.so themselves -- yolact::evalImage, ORBextractor's 3-arg operator(), MovingKeyPoints, ProcessDesp and a
    stack-constructed ORBmatcher's SearchByProjection(CurrentFrame, LastFrame) -- host buffers in and out.  Mean milliseconds per frame."""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(root, "tests", "host", "libamos_host_test.so"))
    lib.amos_host_last_error.restype = C.c_char_p
    gray = np.ascontiguousarray(gray_frames, np.uint8)
    n, h, w = gray.shape
    bgr = np.ascontiguousarray(bgr_frames, np.uint8) if with_mask else None
    py_file = os.path.join(root, "amos-slam_amd", "mask", "yolact_interface.py") if with_mask else ""
    ms, counts = np.zeros(6, np.float64), np.zeros(3, np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    rc = lib.amos_host_frame_latency(py_file.encode(), (weights or "").encode(), ptr(bgr), ptr(gray), C.c_int(n), C.c_int(w), C.c_int(h), C.c_int(5), C.c_int(iters),
                                     C.c_int(-1), ptr(ms), ptr(counts))
    if rc < 0:
        raise RuntimeError("amos_host_frame_latency rc=%d: %s" % (rc, lib.amos_host_last_error().decode()))
    keys = ("eval_image_ms", "detect_ms", "moving_keypoints_ms", "process_desp_ms", "search_by_projection_ms", "frame_ms")
    out = {"what": "ORB_SLAM2::yolact::evalImage -> ORBextractor::operator()(3-arg) -> MovingKeyPoints -> ProcessDesp -> stack ORBmatcher(0.9, true)."
                   "SearchByProjection(CurrentFrame, LastFrame, 15, false): the classes of amos-slam_amd/host/libamos_host.so as Tracking.cc:366,1910 / "
                   "Frame.cc:480-496,633 call them", "frames": iters}
    out.update({k: round(float(v), 4) for k, v in zip(keys, ms)})
    out.update(keypoints_last_frame=int(counts[0]), matches_last_frame=int(counts[1]), eval_image_returned_false=int(counts[2]))
    return out


# --------------------------------------------------------------------------------------------- launcher
def self_launch(args):
    """--gpus N > 1 without a launcher environment: start torch.distributed.run as a CHILD process (this process
    has not imported torch or touched a GPU) and pass its output and return code through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    if proc.returncode != 0:
        print(f"bench.py: torch.distributed.run with {args.gpus} ranks exited with code {proc.returncode}", file=sys.stderr, flush=True)
    return proc.returncode


def dry_run(args):
    """Launcher / collective rehearsal WITHOUT a GPU (tests/test_shard_gloo.py): every rank joins the gloo group,
    pretends to step, and the three collectives of the real path run.  Prints a line with value null."""
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("amos_slam_amd.shard")
    rank, world = shard.init("gloo")
    if os.environ.get("AMOS_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank that dies must fail the whole run
        os._exit(3)
    shard.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, "cpu")
    digest = shard.gather_digests([float(shard.stream_for_rank(rank)), float(zlib.crc32(bytes([rank] * 32))), 0.0], "cpu")
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "dry_run": True, "digest_per_rank": digest}), flush=True)
    shard.finalize()
    return 0


# --------------------------------------------------------------------------------------------- measurement
def count_network_flops(torch, engine, batch, frames_per_forward):
    """FLOPs (2 x multiply-accumulates) of ONE network forward per frame, counted from the shapes the convolutions actually see:
    torch.nn.functional.conv2d is wrapped for one forward with every project kernel switched off, so that every convolution of the
    network goes through it.  Returns (direct, executed): `direct` counts every convolution at kh x kw multiplies per output and channel
    pair (the figure every convolution library is quoted in); `executed` counts the layers that run as Winograd F(2 x 2, 3 x 3) at
    `frames_per_forward` frames per launch (mask/net.py winograd_rule) at the 16 multiplies per 2 x 2 outputs the MFMA units really
    perform (4 / 9 of direct; F(2 x 4, 3 x 3), the default form: 24 per 2 x 4 outputs, 1 / 3 of direct; the transforms' additions and
    the partly empty last tiles of sizes that are not multiples of the tile are not counted)."""
    import importlib
    import torch.nn.functional as F
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    total = {"direct": 0.0, "executed": 0.0, "padding": 0.0}
    # multiplies per output and channel pair of the Winograd form in force over the direct convolution's 9: F(2 x 4): 24 / 8, F(2 x 2): 16 / 4
    wino_share = (24.0 / 8.0 if net_mod.winograd_family() == "24" else 16.0 / 4.0) / 9.0
    real = F.conv2d
    # the merged prediction-head convolution carries 33 all-zero filters (351 -> 384 output channels: a multiple of 64 for the project's
    # kernels, mask/net.py SharedHead.merge_output_layers): they are launch padding, not work of the network -- not counted, neither as
    # direct nor as executed FLOPs (they were in round 3: + 0.8 %)
    head = engine.net.prediction_layers[0]
    merged = getattr(head, "merged", None)
    real_share = 1.0
    if merged is not None:
        n_real = head.bbox_layer.out_channels + head.conf_layer.out_channels + head.mask_layer.out_channels
        real_share = n_real / float(merged.out_channels)

    def counting(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        out = real(x, w, bias, stride, padding, dilation, groups)
        f = 2.0 * w.shape[1] * w.shape[2] * w.shape[3] * out.numel()  # weight [Cout, Cin / groups, kh, kw]
        if merged is not None and w.data_ptr() == merged.weight.data_ptr():
            total["padding"] += f * (1.0 - real_share)
            f *= real_share
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        os.environ["AMOS_MASK_WINOGRAD"] = rule_mode  # the rule as the timed run applies it (this forward itself runs no project kernel)
        try:
            wino = net_mod.winograd_rule(w.shape[1] * groups, w.shape[0], w.shape[2:], pair(stride), pair(padding), pair(dilation), groups,
                                         frames_per_forward, x.shape[2], x.shape[3])
        finally:
            os.environ["AMOS_MASK_WINOGRAD"] = "0"
        total["direct"] += f
        total["executed"] += f * wino_share if wino else f
        return out

    F.conv2d = counting
    keep = {k: os.environ.get(k) for k in ("AMOS_MASK_CONV1X1", "AMOS_MASK_CONV3X3", "AMOS_MASK_WINOGRAD", "AMOS_MASK_STEM")}
    rule_mode = keep["AMOS_MASK_WINOGRAD"] or "1"
    try:
        for k in keep:  # for this one forward every layer goes through F.conv2d (the stem too: its one-kernel form never calls it)
            os.environ[k] = "library" if k == "AMOS_MASK_STEM" else "0"
        with torch.no_grad():
            engine._forward(torch.zeros((batch, 3, 550, 550), device=engine.device))
    finally:
        F.conv2d = real
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return total["direct"] / batch, total["executed"] / batch


def conv_kernel_roofline(torch, amos, dev, frames):
    """The dominant kernel of the mask pass against the fp32 MFMA peak, measured live: the project's convolution kernel on the network's
    largest layer (proto_net[8]: 3 x 3, 256 -> 256 channels at 138 x 138, `frames` frames per launch) through the C ABI, after the timed
    region (the chip is otherwise idle), HIP events on the stream the kernel is launched on.  At the bench's launch sizes that layer runs
    as Winograd F(2 x 4, 3 x 3) (amos::k_winograd24_conv): FLOPs AS EXECUTED = 2 x 24 positions x tiles x cin x cout with tiles =
    frames x 69 x 35 (0.338 of the direct convolution's 2 x output pixels x cout x 9 x cin, reported beside it; F(2 x 2), when
    AMOS_MASK_WINOGRAD_F=22 selects it: 2 x 16 x frames x 69 x 69, 4 / 9); a launch too small
    for the Winograd rule runs the direct implicit GEMM (amos::k_conv_gemm) and the two figures coincide."""
    import importlib
    net_mod = importlib.import_module("amos_slam_amd.mask.net")
    cl = torch.channels_last
    cin = cout = 256
    hw = 138
    x = torch.randn(frames, cin, hw, hw, device=dev).contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / 48.0).contiguous(memory_format=cl)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(frames, cout, hw, hw, device=dev).contiguous(memory_format=cl)
    stream = torch.cuda.current_stream(dev)
    wino = net_mod.winograd_rule(cin, cout, (3, 3), (1, 1), (1, 1), (1, 1), 1, frames, hw, hw)
    if wino:
        f24 = net_mod.winograd_family() == "24"
        make_u, conv = (amos.mask_winograd24_weights, amos.mask_winograd24_conv) if f24 else (amos.mask_winograd_weights, amos.mask_winograd_conv)
        u = torch.empty((24 if f24 else 16) * cin * cout, device=dev)
        make_u(stream.cuda_stream, w.data_ptr(), u.data_ptr(), cin, cout)

        # the pass hands this layer its input channel-blocked ([b][c / 8][h][w][8], net.Blocked) where the chain applies (F(2 x 4), 8+ frames,
        # AMOS_MASK_BLOCKED_CHAIN not 0): measured in that form then
        blocked_in = f24 and frames >= 8 and os.environ.get("AMOS_MASK_BLOCKED_CHAIN", "1") != "0"
        if blocked_in:
            xb = x.permute(0, 2, 3, 1).reshape(frames, hw, hw, cin // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
            del x

            def launch():
                amos.mask_winograd24_conv_layout(stream.cuda_stream, xb.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True, True, False)
        else:
            def launch():
                conv(stream.cuda_stream, x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True)
        name = ("amos::k_winograd24_conv, channel-blocked input" if blocked_in else "amos::k_winograd24_conv") if f24 else "amos::k_winograd_conv"
        # as executed: 2 x positions x tiles x cin x cout (tiles of 2 x 4 outputs, 24 positions; or of 2 x 2 outputs, 16 positions)
        flops = (2.0 * 24 * frames * ((hw + 1) // 2) * ((hw + 3) // 4) * cin * cout) if f24 else (2.0 * 16 * frames * ((hw + 1) // 2) ** 2 * cin * cout)
    else:
        def launch():
            amos.mask_conv(stream.cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, 3, 3, 1, 1, True)
        name = amos.mask_conv_kernel_name(frames, hw, hw, cin, cout, 3, 3, 1, 1)  # the tile shape depends on the launch size
        flops = 2.0 * frames * hw * hw * cout * 9 * cin
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    direct = 2.0 * frames * hw * hw * cout * 9 * cin
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": name + " (proto_net 3x3 256->256 at 138x138: the largest launch of the pass)", "bound": "mfma",
            "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
            "flops_per_launch": int(flops), "direct_convolution_flops_per_launch": int(direct),
            "achieved_direct_equivalent": round(direct / (ms * 1e-3) / 1e12, 2), "frames_per_launch": frames, "avg_launch_ms": round(ms, 4),
            "measured_in": "after the timed region, the kernel alone on the chip, HIP events on the launching stream (10 launches)"}


def use_mask_cfg(args):
    return bool(CONFIGS[args.config].get("mask"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU of the headline config (default 128 = 2 lanes x 64 with the mask; 512 for c2; 128 for c5)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="c3 = BASELINE configs[2] (default, the metric as written); c2 = configs[1]; c5 = configs[4]")
    ap.add_argument("--streams", type=int, default=0, help="independent lanes (handle + HIP streams) the batch is split over")
    ap.add_argument("--leg-steps", type=int, default=100, help="timed steps of the mask-off extract+match sub-leg of a c3 run (0 = skip the leg)")
    ap.add_argument("--leg-batch", type=int, default=512, help="frames per step of that sub-leg (4 lanes)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the 1-thread CPU baseline sample (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-cores", type=int, default=-1, help="processes of the all-cores CPU baseline (default: every granted core, bounded by a cgroup quota and memory; 0 = skip)")
    ap.add_argument("--tum-root", default=os.environ.get("AMOS_TUM_ROOT", ""), help="directory holding TUM RGB-D sequences (default $AMOS_TUM_ROOT): the 640x480 "
                    "configs then run on real frames (BASELINE north_star: fr3/walking_xyz) instead of the synthetic stream")
    ap.add_argument("--tum-sequence", default="", help="sequence(s), comma separated; rank r takes entry r modulo the list (default: fr3_walking_halfsphere for c3, fr3_walking_xyz for c2)")
    ap.add_argument("--tum-associations", default="", help="associations file (`t rgb t depth` per line, rgbd_tum.cc:182-210); default: associations.txt inside the sequence")
    ap.add_argument("--tum-camera-rgb", type=int, choices=[0, 1], default=1, help="Camera.RGB of the settings file (mbRGB, Tracking.cc:308-314): 1 in the reference's "
                    "TUM1/2/3.yaml, i.e. CV_RGB2GRAY applied to the imread B,G,R buffer (default); 0 = CV_BGR2GRAY")
    ap.add_argument("--mask-conv-dtype", choices=["fp32", "fp16", "bf16"], default="fp32",
                    help="precision of the mask network's convolutions (c3 only; fp32 is the parity configuration)")
    ap.add_argument("--mask-chunk", type=int, default=0, help="frames per network forward (default: frames per lane)")
    ap.add_argument("--match-kernel", choices=["auto", "popcount", "mfma"], default="auto", help="brute-force matcher kernel (identical results; A/B switch)")
    ap.add_argument("--latency-frames", type=int, default=30, help="frames of the single-frame drop-in latency measurement reported as the extra key "
                    "`drop_in_latency` (rank 0 of a one-GPU run, after every timed region; 0 = skip); never part of `value`")
    ap.add_argument("--check", action="store_true", help="verify frames of the batch against the oracle")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective rehearsal without a GPU (value null)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        raise SystemExit(dry_run(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks (AMOS_DIST_BACKEND=gloo): ranks share the
    # visible devices and the three collectives run on CPU tensors.  The driver's runs use RCCL, one GPU per rank.
    backend = os.environ.get("AMOS_DIST_BACKEND", "nccl")

    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module("amos_slam_amd.synth")
    shard = importlib.import_module("amos_slam_amd.shard")

    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    shard.init(backend, torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
    coll_dev = dev if backend == "nccl" else "cpu"

    cfg = CONFIGS[args.config]
    W, H = cfg["width"], cfg["height"]
    use_mask = bool(cfg.get("mask"))
    B = args.batch if args.batch > 0 else cfg["default_batch"]
    S = args.streams if args.streams > 0 else cfg["default_streams"]
    S = max(1, min(S, B))
    while B % S:
        S -= 1
    Bl = B // S  # frames per lane and step
    # the mask-off sub-leg of a c3 run uses BASELINE configs[1]'s shape: leg_batch frames over 4 lanes
    leg = use_mask and args.leg_steps > 0
    legS = 4 if leg else 0
    legB = (args.leg_batch // 4) * 4 if leg else 0
    legBl = legB // 4 if leg else 0
    n_lanes = max(S, legS)
    lane_cap_frames = max(Bl, legBl)

    # frames of this rank's stream, resident in HBM before any timed region
    n_frames_dev = max(B, legB)
    tum_source, d_bgr, data_label = None, None, "synthetic"
    if args.tum_root and (W, H) == (640, 480):
        # real frames: rank r takes its own sequence (configs[3]: one sequence per GPU) or, with one sequence, its own stretch of it
        tum = importlib.import_module("amos_slam_amd.tum")
        names = [n for n in (args.tum_sequence or ("fr3_walking_halfsphere" if use_mask_cfg(args) else "fr3_walking_xyz")).split(",") if n]
        seq_name = names[rank % len(names)]
        first = (rank // len(names)) * n_frames_dev
        seq = tum.load_sequence(args.tum_root, seq_name, n_frames_dev, associations=args.tum_associations or None, start=first)
        if seq["bgr"].shape[1:3] != (H, W):
            raise SystemExit(f"TUM frames are {seq['bgr'].shape[2]}x{seq['bgr'].shape[1]}, the config wants {W}x{H}")
        frames_np = tum.bgr_to_gray(seq["bgr"], args.tum_camera_rgb)  # what Tracking.cc:308-321 hands the extractor (the gray-input leg)
        d_bgr = torch.from_numpy(seq["bgr"]).to(dev)  # the colour-input leg reads these (one read: gray pyramid + network input)
        tum_source = ("tum", args.tum_root, seq_name, args.tum_associations or None, first, args.tum_camera_rgb)
        data_label = ("tum:" + seq["name"] + (" (wrapped: %d frames in the sequence)" % seq["frames_in_sequence"] if seq["wrapped"] else "")
                      + " [Camera.RGB=%d: %s on the imread buffer]" % (args.tum_camera_rgb, "CV_RGB2GRAY" if args.tum_camera_rgb else "CV_BGR2GRAY"))
    else:
        frames_np = synth.frames(shard.stream_for_rank(rank), 0, n_frames_dev, H, W)  # one stream per GPU
    d_frames = torch.from_numpy(frames_np).to(dev)
    # gray weights of the colour-input leg: the settings file's Camera.RGB for real frames (the replicated-gray synthetic frames do not care)
    color_rgb_order = bool(args.tum_camera_rgb) if tum_source is not None else False

    # Lanes: extractor + matcher handle with their own HIP streams; kernels of one lane (e.g. the memory-bound
    # pyramid) overlap kernels of another (e.g. the VALU-bound FAST).  Created FIRST: their streams' mapping onto
    # the hardware queues depends on creation order.
    class Lane:
        pass

    lanes = []
    for li in range(n_lanes):
        ln = Lane()
        ln.ext = pkg.OrbExtractor(n_features=cfg["n_features"], n_levels=cfg["n_levels"], max_width=W, max_height=H,
                                  max_batch=lane_cap_frames, device=local_rank)
        ln.matcher = pkg.OrbMatcher(device=local_rank, stream=ln.ext.stream)  # same stream: match follows extract
        ln.matcher.set_bruteforce_kernel(args.match_kernel)
        _, ln.d_desc, ln.d_counts, ln.cap = ln.ext.batch_results_device()
        ln.stream = torch.cuda.ExternalStream(ln.ext.stream, device=local_rank)
        ln.d_match = torch.full((lane_cap_frames, ln.cap, 4), 1 << 30, dtype=torch.int32, device=dev)
        lanes.append(ln)
    cap = lanes[0].cap

    engine, net_flops, net_flops_executed = None, None, None
    chunk = args.mask_chunk if args.mask_chunk > 0 else Bl
    if use_mask:
        mask_mod = importlib.import_module("amos_slam_amd.mask")
        engine = mask_mod.MaskEngine(device=dev, seed=0,
                                     conv_dtype={"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}[args.mask_conv_dtype])
        with torch.no_grad():
            head = engine.net.prediction_layers[0].conf_layer.bias
            b = head.detach().cpu().view(3, 81).clone()
            b[:, 1] += 5.0
            b[1, 3] += 5.5
            head.copy_(b.view(-1).to(head.device))
        engine.prepare()  # fold the batch norms into the convolutions (inference form)
        net_flops, net_flops_executed = count_network_flops(torch, engine, min(chunk, 4), chunk)

    def assign(nl, bl):
        """frames [li * bl, (li + 1) * bl) of the resident stream -> lane li"""
        for li in range(nl):
            ln = lanes[li]
            ln.frames = d_frames[li * bl:(li + 1) * bl]
            ln.bgr_src = d_bgr[li * bl:(li + 1) * bl] if d_bgr is not None else None
            ln.pairs_q = torch.arange(bl, dtype=torch.int32, device=dev)
            ln.pairs_t = (ln.pairs_q - 1) % bl  # frame k against frame k-1 (the lane's first frame against its last)
            ln.bgr = None
        return lanes[:nl]

    def run_leg(active, bl, masked, steps, warmup, alone=False):
        """warmup untimed steps, then exactly `steps` timed steps between barriers; returns (seconds, stage_ms)."""
        match_events, mask_events = [], []
        if masked:
            for ln in active:
                if ln.bgr is None:
                    # the colour frames in HBM: the sequence's own BGR frames, or the synthetic gray stream replicated to three channels
                    ln.bgr = ln.bgr_src.contiguous() if ln.bgr_src is not None else ln.frames.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
                    ln.pre = pkg.MaskPreprocessor(W, H, bl, device=local_rank, stream=ln.ext.stream)
                    ln.net_in = torch.empty((bl, 3, 550, 550), dtype=torch.float32, device=dev)

        def step(timed):
            for li, ln in enumerate(active):
                rec = timed and li == 0  # per-kernel events on lane 0 only
                if masked:
                    # ONE read of the colour frame feeds the gray pyramid (then FAST ... orientation) and the network's input tensor
                    ln.ext.detect_color_with_mask_pre_batch_device(ln.pre, ln.bgr.data_ptr(), H * W * 3, W * 3, W, H, bl, ln.net_in.data_ptr(),
                                                                   rgb_order=color_rgb_order)
                    with torch.cuda.stream(ln.stream):  # the network runs on the lane's stream: ordering is implicit
                        if rec:
                            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            m0.record(ln.stream)
                        ln.masks = engine.eval_net_input_batch(ln.net_in, chunk=chunk, height=H, width=W)  # kept alive until the stream has consumed it
                        if rec:
                            m1.record(ln.stream)
                            mask_events.append((m0, m1))
                        ln.ext.gate_batch_device(ln.masks.data_ptr(), H * W, W)
                        ln.ext.describe_batch_device()
                else:
                    ln.ext.extract_batch_device(ln.frames.data_ptr(), H * W, W, W, H, bl)
                if rec:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ln.stream)
                ln.matcher.bruteforce_best2_batch_device(ln.d_desc, ln.cap * 32, ln.d_counts, ln.pairs_q.data_ptr(), ln.pairs_t.data_ptr(),
                                                         bl, ln.cap, 256, ln.d_match.data_ptr())
                if rec:
                    e1.record(ln.stream)
                    match_events.append((e0, e1))

        def barrier():
            for ln in lanes:
                ln.ext.sync()
            torch.cuda.synchronize()
            if not alone:
                shard.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            step(False)
        active[0].ext.timing_enable(steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        barrier()
        elapsed = time.perf_counter() - t0
        if not alone:
            elapsed = shard.max_over_ranks(elapsed, coll_dev)
        stage_ms, _ = active[0].ext.timing_collect()
        active[0].ext.timing_enable(0)
        stage_ms["match"] = float(np.mean([a.elapsed_time(b) for a, b in match_events])) if match_events else 0.0
        if mask_events:
            stage_ms["mask_pass"] = float(np.mean([a.elapsed_time(b) for a, b in mask_events]))
        return elapsed, stage_ms

    def frame_digest(ln, bl, n=8):
        """per-frame (keypoints, CRC32 of the descriptor bytes, matches within TH_LOW) of the lane's first frames"""
        rows = []
        dm = ln.d_match[:bl].cpu().numpy()
        for f in range(min(bl, n)):
            kps, desc = ln.ext.batch_fetch(f)
            rows += [float(len(kps)), float(zlib.crc32(desc.tobytes())), float(int((dm[f, :len(kps), 1] <= 50).sum()))]
        return rows

    # ---------------------------------------------------------------- headline leg
    active = assign(S, Bl)
    elapsed, stage_ms = run_leg(active, Bl, use_mask, args.steps, args.warmup)
    fps = world * B * args.steps / elapsed
    digest = frame_digest(lanes[0], Bl)
    n_kp = digest[0::3]
    mean_kp = float(np.mean(n_kp))
    digest_all = shard.gather_digests(digest, coll_dev)  # the one collective of the path

    checked, mask_iou = None, None
    if args.check and rank == 0:
        import oracle_binding as ob
        orc = ob.Oracle(n_features=cfg["n_features"], n_levels=cfg["n_levels"])
        checked = 0
        for li, f in ((0, 0), (S - 1, Bl - 1)):
            kg, dg = lanes[li].ext.batch_fetch(f)
            frame = frames_np[li * Bl + f]
            if use_mask:
                orc.detect(frame)
                orc.gate(lanes[li].masks[f].cpu().numpy())
                ko, do = orc.describe()
            else:
                ko, do = orc.extract(frame)
            assert kg.tobytes() == ko.tobytes() and dg.tobytes() == do.tobytes(), f"lane {li} frame {f} differs from the oracle"
            checked += 1
        if use_mask:
            # the masks themselves: lane 0's first and last frame of the LAST timed step (made by the batch launches the headline is made of)
            # against a one-frame eager pass of the same frame through the engine's own entry point (IoU >= 1 - 1e-3, north_star's bound)
            mask_iou = []
            for f in (0, Bl - 1):
                bgr1 = lanes[0].bgr[f].cpu().numpy()
                one = engine.eval_bgr(bgr1)
                got = lanes[0].masks[f].cpu().numpy() > 0
                want = (one.cpu().numpy() > 0) if one is not None else np.zeros_like(got)
                union = int((got | want).sum())
                iou = 1.0 if union == 0 else int((got & want).sum()) / union
                assert iou >= 1 - 1e-3, f"lane 0 frame {f}: mask of the {Bl}-frame launch vs the one-frame pass: IoU {iou}"
                mask_iou.append(round(iou, 6))

    # ---------------------------------------------------------------- mask-off extract+match leg (c3 runs) or the same leg's extras
    em = None  # dict describing the extract+match measurement the roofline refers to
    if use_mask and leg:
        em_active = assign(legS, legBl)
        em_elapsed, em_stage = run_leg(em_active, legBl, False, args.leg_steps, 3)
        em = dict(S=legS, Bl=legBl, B=legB, steps=args.leg_steps, elapsed=em_elapsed, stage_ms=em_stage,
                  fps=world * legB * args.leg_steps / em_elapsed, mean_kp=float(np.mean(frame_digest(lanes[0], legBl)[0::3])))
    elif not use_mask:
        em = dict(S=S, Bl=Bl, B=B, steps=args.steps, elapsed=elapsed, stage_ms=stage_ms, fps=fps, mean_kp=mean_kp)
    alone_ms = None
    if em and em["S"] > 1:
        # the same per-launch times with lane 0 ALONE on the chip (the other lanes idle)
        _, alone_ms = run_leg(lanes[:1], em["Bl"], False, 4, 0, alone=True)

    # SURVEY 8d match workload (ii): window-gated search (SearchByProjection radius 15 * scale, levels
    # octave-1..octave+1) of frame k's keypoints in frame k-1, everything resident: grid cells, CSR grid,
    # window best-2.  Measured after the timed regions on lane 0; not part of `value`.
    gated = None
    if em:
        ln, bl = lanes[0], em["Bl"]
        d_kps = ln.ext.batch_results_device()[0]
        d_cell = torch.zeros((bl, ln.cap), dtype=torch.int32, device=dev)
        d_start = torch.zeros((bl, 64 * 48 + 1), dtype=torch.int32, device=dev)
        d_items = torch.zeros((bl, ln.cap), dtype=torch.int32, device=dev)
        d_win = torch.zeros((bl, ln.cap, 4), dtype=torch.int32, device=dev)
        bounds = (0.0, float(W), 0.0, float(H))
        sf = ln.ext.tables()["scale"]
        torch.cuda.synchronize()
        evs = []
        for it in range(6):
            with torch.cuda.stream(ln.stream):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record(ln.stream)
                ln.ext.rgbd_glue_batch_device(None, False, 1.0, 0, 0, 0.0, bounds, None, None, d_cell.data_ptr())
                ev[1].record(ln.stream)
                ln.matcher.grid_build_batch_device(d_cell.data_ptr(), ln.d_counts, bl, ln.cap, d_start.data_ptr(), d_items.data_ptr())
                ev[2].record(ln.stream)
                ln.matcher.window_best2_batch_device(d_kps, ln.d_desc, ln.d_counts, d_start.data_ptr(), d_items.data_ptr(),
                                                     ln.pairs_q.data_ptr(), ln.pairs_t.data_ptr(), bl, ln.cap, sf, 15.0, d_win.data_ptr(),
                                                     mode=0, bounds=bounds)
                ev[3].record(ln.stream)
            if it > 0:
                evs.append(ev)
        ln.ext.sync()
        torch.cuda.synchronize()
        g_ms = [float(np.mean([e[k].elapsed_time(e[k + 1]) for e in evs])) for k in range(3)]
        gated = {"workload": "frame k keypoints searched in frame k-1: window 15*scale, levels octave-1..octave+1, best-2",
                 "frames_per_launch": bl, "grid_cells_ms": round(g_ms[0], 4), "grid_build_ms": round(g_ms[1], 4),
                 "window_best2_ms": round(g_ms[2], 4), "queries_per_s": round(em["mean_kp"] * bl / (sum(g_ms) * 1e-3), 1),
                 "matched_within_TH_HIGH": int((d_win[:, :, 1] <= 100).sum().item())}

    # The path Tracking.cc:366 / Frame.cc:480-496 / Tracking.cc:1910 really take: ONE frame at a time, host buffers in and out (PCIe inside the
    # figure).  An extra key for the reader, measured after every timed region; the headline is the resident batch rate above, never this.
    #   cxx           -- the C++ drop-in classes themselves (amos-slam_amd/host/libamos_host.so through the harness tests/host/libamos_host_test.so):
    #                    yolact::evalImage -> ORBextractor 3-arg operator() -> MovingKeyPoints -> ProcessDesp -> a STACK-constructed
    #                    ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, LastFrame, 15, false) per frame
    #   ctypes_mirror -- the same C ABI driven from Python with persistent handles (amos-slam_amd/__init__.py; NOT what the reference's C++ links):
    #                    the same call sequence up to the descriptors (mask session, detect, gate, describe), then an N x N best-2 match
    latency = None
    if rank == 0 and world == 1 and args.latency_frames > 0 and (W, H) == (640, 480):
        try:
            n = args.latency_frames
            lat_ext = pkg.OrbExtractor(n_features=cfg["n_features"], n_levels=cfg["n_levels"], max_width=W, max_height=H, max_batch=1, device=local_rank)
            lat_m = pkg.OrbMatcher(device=local_rank)
            n_src = min(len(frames_np), 16)
            fr = [np.ascontiguousarray(frames_np[k % n_src]) for k in range(n + 5)]
            bgr_np = (d_bgr[:n_src].cpu().numpy() if d_bgr is not None else np.repeat(frames_np[:n_src, :, :, None], 3, axis=3)) if use_mask else None
            zero_mask = np.zeros((H, W), np.uint8)
            session = engine.frame_session(H, W) if use_mask else None  # pre-processing + network + detection + mask assembly of one frame as ONE HIP graph

            def one_frame(k, match_prev):
                """mask session -> detect -> gate -> describe (-> N x N match) through the ctypes mirror; returns the descriptors"""
                m = zero_mask
                if session is not None:
                    session.frame_in.numpy()[...] = bgr_np[k % n_src]
                    m = session.mask_out.numpy() if session.run() else zero_mask
                lat_ext.detect(fr[k])
                lat_ext.gate(m)
                _, dsc = lat_ext.describe()
                if match_prev is not None and len(dsc) and len(match_prev):
                    lat_m.bruteforce_best2(dsc, match_prev)
                return dsc

            for k in range(5):
                lat_ext.extract(fr[k])
                one_frame(k, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in fr[5:]:
                lat_ext.extract(f)
            t1 = time.perf_counter()
            prev = None
            for f in fr[5:]:
                _, dsc = lat_ext.extract(f)
                if prev is not None:
                    lat_m.bruteforce_best2(dsc, prev)
                prev = dsc
            t2 = time.perf_counter()
            for k in range(5, n + 5):
                one_frame(k, None)
            t3 = time.perf_counter()
            prev = None
            for k in range(5, n + 5):
                prev = one_frame(k, prev if prev is not None else np.zeros((0, 32), np.uint8))
            t4 = time.perf_counter()
            mirror = {"what": "persistent handles through the ctypes mirror of the C ABI (amos-slam_amd/__init__.py), one 640x480 frame per call, host buffers in "
                              "and out: NOT the C++ classes the reference's call sites link -- those are `cxx` below",
                      "frames": n, "orb_extract_4arg_ms": round((t1 - t0) / n * 1e3, 3), "orb_extract_4arg_and_nxn_match_ms": round((t2 - t1) / n * 1e3, 3),
                      "mask_detect_gate_describe_ms": round((t3 - t2) / n * 1e3, 3), "mask_detect_gate_describe_nxn_match_ms": round((t4 - t3) / n * 1e3, 3)}
            if session is not None:
                t5 = time.perf_counter()
                for _ in range(n):
                    session.run()
                mirror["mask_pass_one_frame_graph_ms"] = round((time.perf_counter() - t5) / n * 1e3, 3)
            mirror["frames_per_s_one_stream"] = round(1e3 / mirror["mask_detect_gate_describe_nxn_match_ms"], 1)
            latency = {"what": "one 640x480 frame per call, host buffers in and out (PCIe inside), everything on one host thread; never `value`",
                       "ctypes_mirror": mirror}
            try:
                wpath = ""
                if use_mask:
                    # the C++ class loads its weights from a .pth like the reference's (System.cc:107): the same seeded network with the same
                    # class-head bias as the engine above, saved in the checkpoint's key layout (before folding)
                    import tempfile
                    raw = mask_mod.MaskEngine(device="cpu", seed=0)
                    with torch.no_grad():
                        hb_ = raw.net.prediction_layers[0].conf_layer.bias
                        bb = hb_.detach().view(3, 81).clone()
                        bb[:, 1] += 5.0
                        bb[1, 3] += 5.5
                        hb_.copy_(bb.view(-1))
                    wpath = os.path.join(tempfile.mkdtemp(prefix="amos_bench_"), "yolact_seed0.pth")
                    torch.save(raw.net.state_dict(), wpath)
                    os.environ["AMOS_MASK_DEVICE"] = dev
                latency["cxx"] = cxx_frame_latency(ROOT, frames_np[:n_src], bgr_np, n, use_mask, wpath)
                c = latency["cxx"]
                c["frames_per_s_one_stream"] = round(1e3 / c["frame_ms"], 1)
                # the shared part of the two (up to the descriptors): the C++ classes against the mirror
                c["up_to_descriptors_ms"] = round(c["eval_image_ms"] + c["detect_ms"] + c["moving_keypoints_ms"] + c["process_desp_ms"], 4)
                c["up_to_descriptors_vs_ctypes_mirror"] = round(c["up_to_descriptors_ms"] / mirror["mask_detect_gate_describe_ms"], 3)
            except Exception as exc:
                latency["cxx"] = {"error": repr(exc)[:300]}
        except Exception as exc:  # an extra: never fail the bench line over it
            latency = {"error": repr(exc)[:300]}

    if rank == 0:
        lw, lh = lanes[0].ext.level_sizes(W, H)
        out = {
            "metric": METRIC if use_mask else "front-end frames/sec (ORB extract+match, mask off) at %dx%d" % (W, H),
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if not use_mask else "u8 (ORB, gate, matcher) + %s (mask network)" % args.mask_conv_dtype,
            "data": data_label,
            "config": {"workload": cfg["label"], "frames_per_step_per_gpu": B, "lanes_per_gpu": S, "frames_per_launch": Bl, "width": W, "height": H,
                       "n_features": cfg["n_features"], "n_levels": cfg["n_levels"], "ini_th_fast": 20, "min_th_fast": 7,
                       "mean_keypoints_per_frame": round(mean_kp, 1), "match": "frame k vs k-1, N x N best-2",
                       "frames": "resident in HBM before the timed region (offline replay: %.0f MB/s of frames, no host link in the figure)" % (fps / world * W * H / 1e6),
                       "synthetic_stream_seed": ("stream = rank, frame k seeded 1000*rank+k (amos-slam_amd/synth.py)" if tum_source is None
                                                 else "n/a: frames of " + data_label + " from frame %d on" % tum_source[4]),
                       "parallelism": f"frames sharded {world} ways, one process per GPU, no data-path collective"},
            "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "digest_per_rank": {"layout": "per frame of the rank's first lane: keypoints, CRC32 of descriptor bytes, matches <= TH_LOW", "rows": digest_all},
        }
        if checked is not None:
            out["oracle_checked_frames"] = checked
            if use_mask:
                out["mask_checked_iou_vs_one_frame_pass"] = mask_iou
        import torch.distributed as dist
        if dist.is_initialized():  # under a launcher, also with one rank: the collectives above ran through this group
            out["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if use_mask:
            out["config"]["mask"] = {"network": "YOLACT-R50-FPN 550x550, %s, batch-norm folded, NHWC, random weights (no checkpoint offline)" % args.mask_conv_dtype,
                                     "frames_per_forward": chunk}
            n_fwd = (Bl + chunk - 1) // chunk
            net_ms = stage_ms.get("mask_pass", 0.0)
            step_s = elapsed / args.steps
            tf = net_flops_executed * B / step_s / 1e12  # every lane's forwards of one step over the step's wall time: the whole chip's rate
            tf_direct = net_flops * B / step_s / 1e12
            out["roofline_mask"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                    "frac": round(tf / MFMA_F32_PEAK_TF, 4), "flops_per_frame": int(net_flops),
                                    "flops_per_frame_executed": int(net_flops_executed), "achieved_direct_equivalent": round(tf_direct, 2),
                                    "frames_per_step": B, "lanes": S, "lane_pass_ms": round(net_ms, 3), "frames_per_lane_pass": Bl,
                                    "forwards_per_lane_pass": n_fwd,
                                    "note": "achieved / frac: convolution FLOPs AS EXECUTED on the MFMA units (2 x MAC; the stride-1 3 x 3 layers that run as "
                                            "Winograd F(2x4,3x3) counted at 24 instead of 72 multiplies per 2 x 4 outputs: flops_per_frame_executed) of all frames "
                                            "of a step / the step's wall time, i.e. a lower bound of the convolution kernels' own rate: the step also holds the "
                                            "pre / post-processing, the ORB kernels and the match.  flops_per_frame / achieved_direct_equivalent: the same "
                                            "layers counted as direct convolutions (the figure comparable with a direct-convolution implementation; it may "
                                            "exceed the peak).  The 33 all-zero filters that pad the merged prediction head to 384 channels are counted in neither figure.  "
                                            "lane_pass_ms = one lane's whole mask pass (events on its stream), lanes overlap."}
            if args.mask_conv_dtype == "fp32":
                out["roofline_mask"]["dominant_kernel"] = conv_kernel_roofline(torch, pkg, torch.device(dev), chunk)
        if em:
            st = em["stage_ms"]
            alg = algorithmic_bytes(lw, lh, em["mean_kp"], W, H)
            total_alg = sum(alg.values())
            # the dominant KERNEL: largest total time per pass; "pyramid" is the resize launches of one pass,
            # so its per-launch figures are the averages over those launches (what rocprofv3 --stats reports)
            launches = {k: 1 for k in st}
            launches["pyramid"] = lanes[0].ext.pyramid_launches()
            # only byte-moving stages can be the HBM-roofline kernel: the quad-tree selection (alg 0, latency-bound,
            # ~constant time per launch) may out-last FAST at toy batches but has no bandwidth figure to report
            movers = [k for k in st if alg.get(k, 0) > 0 and st[k] > 0]
            dominant = max(movers, key=lambda k: st[k]) if movers else "fast"
            dom_bytes = alg[dominant] * em["Bl"] / launches.get(dominant, 1)  # algorithmic bytes of one (average) launch
            dom_ms = st.get(dominant, 0.0) / launches.get(dominant, 1)
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    ent = tj.get("c2" if W == 640 else args.config, {}).get(dominant)
                    if ent and ent.get("batch") == em["Bl"]:
                        traffic = ent["hbm_bytes_per_launch"] / launches[dominant]
                        traffic_src = tj.get("source", "profiles/traffic.json") + " (PMC pass of an earlier run of this command, not this run)"
                except Exception:
                    traffic = None
            em_fps = em["fps"]
            out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": (round(achieved, 2) if achieved is not None else None),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (round(achieved / HBM_PEAK_GBS, 5) if achieved is not None else None), "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                               "launches_per_pass": launches[dominant],
                               "avg_launch_ms_is": "a HIP-event INTERVAL on the launching stream (lane 0) inside the timed region: with several lanes running it "
                                                   "includes the time the launch waits behind / shares CUs with the other lanes' kernels, so it is longer than the "
                                                   "kernel duration rocprofv3 reports for the same launches (profiles/*_steady_mask_off_kernel_stats.csv); the same "
                                                   "launch alone on the chip is roofline_lane_alone",
                               "measured_in": "extract_match_leg timed region (HIP events on lane 0's streams)" if use_mask else "the timed region (HIP events on lane 0's streams)"}
            pipeline = {"algorithmic_bytes_per_frame": int(total_alg), "achieved_GBs": round(total_alg * em_fps / world / 1e9, 2),
                        "frac": round(total_alg * em_fps / world / 1e9 / HBM_PEAK_GBS, 5)}
            stage_frac = {k: (round(alg[k] * em["Bl"] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if v > 0 and k in alg else None) for k, v in st.items()}
            alone = None
            if alone_ms and alone_ms.get(dominant, 0) > 0:
                a_ms = alone_ms[dominant] / launches[dominant]
                alone = {"kernel": dominant, "note": "same launch, measured after the timed region with the other lanes idle",
                         "avg_launch_ms": round(a_ms, 4), "achieved": round(dom_bytes / (a_ms * 1e-3) / 1e9, 2),
                         "frac": round(dom_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            leg_obj = {"metric": "front-end frames/sec (ORB extract+match, mask off) at %dx%d" % (W, H), "value": round(em_fps, 1), "unit": "frames/s",
                       "workload": CONFIGS["c2"]["label"] if W == 640 else cfg["label"], "steps": em["steps"],
                       "ms_per_step": round(em["elapsed"] / em["steps"] * 1e3, 4), "frames_per_step_per_gpu": em["B"], "lanes_per_gpu": em["S"],
                       "frames_per_launch": em["Bl"], "mean_keypoints_per_frame": round(em["mean_kp"], 1), "pipeline_roofline": pipeline,
                       "stage_ms_per_launch": {k: round(v, 4) for k, v in st.items()}, "stage_frac_of_hbm_peak": stage_frac,
                       "stage_ms_per_launch_lane_alone": ({k: round(v, 4) for k, v in alone_ms.items()} if alone_ms else None),
                       "roofline_lane_alone": alone,
                       # Hamming throughput of the N x N match as N_q * N_t * 256 bit comparisons per second
                       "hamming_bitops_per_s": round(em_fps * em["mean_kp"] * em["mean_kp"] * 256.0, 1)}
            if use_mask:
                out["extract_match_leg"] = leg_obj
            else:
                out.update({k: v for k, v in leg_obj.items() if k not in ("metric", "value", "unit", "steps", "ms_per_step", "workload")})
            out["pipeline_roofline"] = pipeline
        if gated:
            out["gated_match"] = gated
        if latency:
            out["drop_in_latency"] = latency
        n_cpu = args.cpu_frames if args.cpu_frames >= 0 else (100 if W == 640 else 12)
        if world == 1 and n_cpu > 0:
            granted, quota = granted_cores()
            cores = args.cpu_cores if args.cpu_cores >= 0 else all_cores_worker_count()
            one = cpu_baseline_one_thread(synth, cfg, n_cpu, tum_source or ("synth", 0))
            if cores > 1:
                out["cpu_baseline"] = cpu_baseline_all_cores(cfg, cores, 48 if W == 640 else 4, tum_source)
                out["cpu_baseline"]["single_thread"] = one
            else:
                out["cpu_baseline"] = one
            out["cpu_baseline"]["host_cores_visible"] = os.cpu_count()
            out["cpu_baseline"]["host_cores_granted"] = granted
            out["cpu_baseline"]["cgroup_cpu_quota_cores"] = quota
            out["cpu_baseline"]["workload"] = "ORB extract + N x N best-2 match (the reference's CPU path, src/ORBextractor.cc + ORBmatcher.cc; its mask network runs on a GPU in the reference too)"
        print(json.dumps(out), flush=True)

    shard.finalize()


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:  # a failed rank must not sit in the process group's teardown
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise
        if isinstance(exc, SystemExit) and isinstance(exc.code, int):
            raise
        if isinstance(exc, SystemExit):
            print(exc.code, file=sys.stderr, flush=True)
            os._exit(2)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
