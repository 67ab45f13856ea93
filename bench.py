#!/usr/bin/env python3
"""bench.py -- front-end frames/sec (ORB extract + brute-force Hamming match) on MI355X.

A step = one pass of the hot path over one batch of B synthetic 640x480 frames that are already
resident in HBM: ORB extraction of every frame (pyramid, FAST, quad-tree, orientation, blur, rBRIEF)
followed by the N x N best-2 Hamming match of frame k against frame k-1.  One process per GPU; rank r
works on its own synthetic stream r (frames are independent: no data-path collective, weak scaling);
RCCL is used only for the barrier / max-over-ranks time and the final gather of result digests.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP multiplexes streams onto this many hardware queues in creation order; the lanes below create their
# streams as main0, side0, main1, side1, ... so that with 4 queues the main streams of lanes (0, 2) and (1, 3)
# and the side streams of (0, 2) and (1, 3) share a queue each -- measured best.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # BASELINE.json configs[1]: single MI355X, extract + brute-force match, mask disabled
    "c2": dict(width=640, height=480, n_features=1000, n_levels=8, default_batch=512, default_streams=4, label="configs[1]: 640x480 L8 N1000 extract+match, mask off"),
    # BASELINE.json configs[2]: full front-end incl. the YOLACT mask (network on PyTorch-ROCm, random weights
    # with a biased class head so that ~100 detections exercise the whole post-processing chain)
    "c3": dict(width=640, height=480, n_features=1000, n_levels=8, mask=True, default_batch=32,
               label="configs[2]: 640x480 L8 N1000 YOLACT-R50 mask + extract + gate + match"),
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
        "mask_net": 0,  # MFMA-bound convolutions on PyTorch: timed, not part of the HBM roofline
    }


def cpu_baseline(synth, cfg, n_sample):
    """The oracle (CPU restatement of src/ORBextractor.cc + ORBmatcher.cc, 1 thread) on a bounded
    sample of the same workload.  Checker only: it is never the thing shipped or measured as GPU."""
    import oracle_binding as ob
    orc = ob.Oracle(n_features=cfg["n_features"], n_levels=cfg["n_levels"])
    frames = [synth.frame(0, k, cfg["height"], cfg["width"]) for k in range(n_sample + 1)]
    t0 = time.perf_counter()
    prev = orc.extract(frames[0])[1]
    for k in range(1, n_sample + 1):
        _, desc = orc.extract(frames[k])
        ob.bruteforce_best2(desc, prev)
        prev = desc
    dt = time.perf_counter() - t0
    return {"value": round(n_sample / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_sample} frames of the same synthetic stream: oracle extract + N x N best-2 match, 1 thread"}


def _cpu_worker(job):
    """One worker process of the all-cores CPU baseline: frames of its own synthetic stream."""
    root, cfg, stream, n = job
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    synth = importlib.import_module("amos_slam_amd.synth")
    import oracle_binding as ob
    orc = ob.Oracle(n_features=cfg["n_features"], n_levels=cfg["n_levels"])
    frames = [synth.frame(stream, k, cfg["height"], cfg["width"]) for k in range(n + 1)]
    t0 = time.perf_counter()
    prev = orc.extract(frames[0])[1]
    for k in range(1, n + 1):
        _, desc = orc.extract(frames[k])
        ob.bruteforce_best2(desc, prev)
        prev = desc
    return time.perf_counter() - t0


def cpu_baseline_all_cores(cfg, workers, n_per_worker):
    """The same oracle, one frame stream per process, `workers` processes (the GPU box grants 16 host
    cores per GPU).  Extra information beside the contract's single-thread cpu_baseline."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(ROOT, {k: cfg[k] for k in ("n_features", "n_levels", "height", "width")}, 100 + w, n_per_worker)
                               for w in range(workers)])
        dt = time.perf_counter() - t0  # includes process start-up and frame synthesis: a lower bound on fps
    return {"value": round(workers * n_per_worker / dt, 2), "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{workers} processes x {n_per_worker} frames (wall time incl. start-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default 512; 32 with the mask; 128 for c5)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--streams", type=int, default=0, help="independent lanes (handle + HIP streams) the batch is split over (default 4; 2 with the mask)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--mask-conv-dtype", choices=["fp32", "fp16", "bf16"], default="fp32",
                    help="precision of the mask network's convolutions (c3 only; fp32 is the parity configuration)")
    ap.add_argument("--check", action="store_true", help="verify one frame of the batch against the oracle")
    ap.add_argument("--cpu-all-cores", type=int, default=0, help="also time the oracle on this many processes (0 = off)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks (AMOS_DIST_BACKEND=gloo): ranks share the
    # visible devices and the three collectives run on CPU tensors.  The driver's runs use RCCL, one GPU per rank.
    backend = os.environ.get("AMOS_DIST_BACKEND", "nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module("amos_slam_amd.synth")
    shard = importlib.import_module("amos_slam_amd.shard")

    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    shard.init(backend, torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
    coll_dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"

    cfg = CONFIGS[args.config]
    if args.batch <= 0:
        args.batch = cfg.get("default_batch", 256)
    if args.streams <= 0:
        args.streams = cfg.get("default_streams", 2)
    W, H, B = cfg["width"], cfg["height"], args.batch
    use_mask = bool(cfg.get("mask"))
    frames_np = synth.frames(shard.stream_for_rank(rank), 0, B, H, W)  # one stream per GPU
    d_frames = torch.from_numpy(frames_np).cuda(local_rank)

    # The batch is split over S independent lanes (extractor + matcher handle, own HIP streams): kernels of
    # one lane (e.g. the memory-bound pyramid) overlap kernels of the other (e.g. the VALU-bound FAST).
    S = max(1, min(args.streams, B))
    while B % S:
        S -= 1
    Bl = B // S  # frames per lane and step

    class Lane:
        pass

    engine = None
    if use_mask:
        mask_mod = importlib.import_module("amos_slam_amd.mask")
        engine = mask_mod.MaskEngine(device=f"cuda:{local_rank}", seed=0,
                                     conv_dtype={"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}[args.mask_conv_dtype])
        with torch.no_grad():
            head = engine.net.prediction_layers[0].conf_layer.bias
            b = head.detach().cpu().view(3, 81).clone()
            b[:, 1] += 5.0
            b[1, 3] += 5.5
            head.copy_(b.view(-1).to(head.device))
    lanes = []
    for li in range(S):
        ln = Lane()
        ln.ext = pkg.OrbExtractor(n_features=cfg["n_features"], n_levels=cfg["n_levels"], max_width=W, max_height=H,
                                  max_batch=Bl, device=local_rank)
        ln.matcher = pkg.OrbMatcher(device=local_rank, stream=ln.ext.stream)  # same stream: match follows extract
        _, ln.d_desc, ln.d_counts, ln.cap = ln.ext.batch_results_device()
        ln.frames = d_frames[li * Bl:(li + 1) * Bl]
        ln.pairs_q = torch.arange(Bl, dtype=torch.int32, device=f"cuda:{local_rank}")
        ln.pairs_t = (ln.pairs_q - 1) % Bl  # frame k against frame k-1 (the lane's first frame against its last)
        ln.d_match = torch.full((Bl, ln.cap, 4), 1 << 30, dtype=torch.int32, device=f"cuda:{local_rank}")
        ln.stream = torch.cuda.ExternalStream(ln.ext.stream, device=local_rank)
        ln.bgr = ln.frames.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous() if use_mask else None  # gray as BGR, in HBM
        lanes.append(ln)
    ext, cap = lanes[0].ext, lanes[0].cap
    torch.cuda.synchronize()
    match_events, mask_events = [], []

    def step(timed):
        for li, ln in enumerate(lanes):
            rec = timed and li == 0  # per-kernel events on lane 0 only
            if use_mask:
                ln.ext.detect_batch_device(ln.frames.data_ptr(), H * W, W, W, H, Bl)
                with torch.cuda.stream(ln.stream):  # the network runs on the lane's stream: ordering is implicit
                    if rec:
                        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        m0.record(ln.stream)
                    ln.masks = engine.eval_bgr_batch(ln.bgr)  # kept alive until the stream has consumed it
                    if rec:
                        m1.record(ln.stream)
                        mask_events.append((m0, m1))
                    ln.ext.gate_batch_device(ln.masks.data_ptr(), H * W, W)
                    ln.ext.describe_batch_device()
            else:
                ln.ext.extract_batch_device(ln.frames.data_ptr(), H * W, W, W, H, Bl)
            if rec:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(ln.stream)
            ln.matcher.bruteforce_best2_batch_device(ln.d_desc, ln.cap * 32, ln.d_counts, ln.pairs_q.data_ptr(), ln.pairs_t.data_ptr(),
                                                     Bl, ln.cap, 256, ln.d_match.data_ptr())
            if rec:
                e1.record(ln.stream)
                match_events.append((e0, e1))

    def barrier():
        for ln in lanes:
            ln.ext.sync()
        torch.cuda.synchronize()
        shard.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    ext.timing_enable(args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, coll_dev)

    stage_ms, n_rec = ext.timing_collect()
    stage_ms["match"] = float(np.mean([a.elapsed_time(b) for a, b in match_events])) if match_events else 0.0
    if mask_events:
        stage_ms["mask_net"] = float(np.mean([a.elapsed_time(b) for a, b in mask_events]))
    # The same per-launch times with lane 0 ALONE on the chip (after the timed region; the other lanes idle): what
    # one launch of each kernel costs when it does not share the CUs with three other lanes' kernels.
    alone_ms = None
    if S > 1 and not use_mask:
        barrier()
        ext.timing_enable(4)
        ev_alone = []
        for _ in range(4):
            ln = lanes[0]
            ln.ext.extract_batch_device(ln.frames.data_ptr(), H * W, W, W, H, Bl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ln.stream)
            ln.matcher.bruteforce_best2_batch_device(ln.d_desc, ln.cap * 32, ln.d_counts, ln.pairs_q.data_ptr(), ln.pairs_t.data_ptr(),
                                                     Bl, ln.cap, 256, ln.d_match.data_ptr())
            e1.record(ln.stream)
            ev_alone.append((e0, e1))
        barrier()
        alone_ms, _ = ext.timing_collect()
        alone_ms["match"] = float(np.mean([a.elapsed_time(b) for a, b in ev_alone]))
    # result digest: keypoint counts + number of matches within TH_LOW (final gather over RCCL)
    n_kp = [len(ext.batch_fetch(f)[0]) for f in range(min(Bl, 8))]
    mean_kp = float(np.mean(n_kp))
    good = int(sum(int((ln.d_match[:, :, 1] <= 50).sum().item()) for ln in lanes))
    # the one collective of the path: final gather of the per-rank digests
    digest_all = shard.gather_digests([float(sum(n_kp)), float(good)], coll_dev)

    # SURVEY 8d match workload (ii): window-gated search (SearchByProjection radius 15 * scale, levels
    # octave-1..octave+1) of frame k's keypoints in frame k-1, everything resident: grid cells, CSR grid,
    # window best-2.  Measured after the timed region on lane 0; not part of `value`.
    gated = None
    if not use_mask:
        ln = lanes[0]
        dev = f"cuda:{local_rank}"
        d_kps = ln.ext.batch_results_device()[0]
        d_cell = torch.zeros((Bl, ln.cap), dtype=torch.int32, device=dev)
        d_start = torch.zeros((Bl, 64 * 48 + 1), dtype=torch.int32, device=dev)
        d_items = torch.zeros((Bl, ln.cap), dtype=torch.int32, device=dev)
        d_win = torch.zeros((Bl, ln.cap, 4), dtype=torch.int32, device=dev)
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
                ln.matcher.grid_build_batch_device(d_cell.data_ptr(), ln.d_counts, Bl, ln.cap, d_start.data_ptr(), d_items.data_ptr())
                ev[2].record(ln.stream)
                ln.matcher.window_best2_batch_device(d_kps, ln.d_desc, ln.d_counts, d_start.data_ptr(), d_items.data_ptr(),
                                                     ln.pairs_q.data_ptr(), ln.pairs_t.data_ptr(), Bl, ln.cap, sf, 15.0, d_win.data_ptr(),
                                                     mode=0, bounds=bounds)
                ev[3].record(ln.stream)
            if it > 0:
                evs.append(ev)
        ln.ext.sync()
        torch.cuda.synchronize()
        g_ms = [float(np.mean([e[k].elapsed_time(e[k + 1]) for e in evs])) for k in range(3)]
        nq = mean_kp * Bl
        gated = {"workload": "frame k keypoints searched in frame k-1: window 15*scale, levels octave-1..octave+1, best-2",
                 "frames_per_launch": Bl, "grid_cells_ms": round(g_ms[0], 4), "grid_build_ms": round(g_ms[1], 4),
                 "window_best2_ms": round(g_ms[2], 4), "queries_per_s": round(nq / (sum(g_ms) * 1e-3), 1),
                 "matched_within_TH_HIGH": int((d_win[:, :, 1] <= 100).sum().item())}

    if args.check and rank == 0:
        import oracle_binding as ob
        orc = ob.Oracle(n_features=cfg["n_features"], n_levels=cfg["n_levels"])
        for li, f in ((0, 0), (S - 1, Bl - 1)):
            kg, dg = lanes[li].ext.batch_fetch(f)
            ko, do = orc.extract(frames_np[li * Bl + f])
            assert kg.tobytes() == ko.tobytes() and dg.tobytes() == do.tobytes(), f"frame {f} differs from the oracle"

    if rank == 0:
        lw, lh = ext.level_sizes(W, H)
        alg = algorithmic_bytes(lw, lh, mean_kp, W, H)
        total_alg = sum(alg.values())
        # the dominant KERNEL: largest total time per pass; "pyramid" is n_levels - 1 launches of one kernel,
        # so its per-launch figures are the averages over those launches (what rocprofv3 --stats reports)
        launches = {k: 1 for k in stage_ms}
        launches["pyramid"] = cfg["n_levels"] - 1
        dominant = max((k for k in stage_ms if k != "mask_net"), key=lambda k: stage_ms[k])
        dom_bytes = alg[dominant] * Bl / launches[dominant]  # algorithmic bytes of one (average) launch
        dom_ms = stage_ms[dominant] / launches[dominant]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(args.config, {}).get(dominant)
                if ent and ent.get("batch") == Bl:
                    traffic = ent["hbm_bytes_per_launch"] / launches[dominant]
            except Exception:
                traffic = None
        fps = world * B * args.steps / elapsed
        out = {
            "metric": "front-end frames/sec (ORB extract+match%s) at %dx%d" % ("+mask" if use_mask else ", mask off", W, H),
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if not use_mask else "u8 (ORB, matcher) + %s (mask network)" % args.mask_conv_dtype,
            "data": "synthetic",
            "config": {"workload": cfg["label"], "frames_per_step_per_gpu": B, "lanes_per_gpu": S, "frames_per_launch": Bl, "width": W, "height": H,
                       "n_features": cfg["n_features"], "n_levels": cfg["n_levels"], "ini_th_fast": 20, "min_th_fast": 7,
                       "mean_keypoints_per_frame": round(mean_kp, 1), "match": "frame k vs k-1, N x N best-2",
                       "synthetic_stream_seed": "stream = rank, frame k seeded 1000*rank+k (amos-slam_amd/synth.py)",
                       "parallelism": f"frames sharded {world} ways, one process per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                         "launches_per_pass": launches[dominant]},
            "pipeline_roofline": {"algorithmic_bytes_per_frame": int(total_alg),
                                  "achieved_GBs": round(total_alg * fps / world / 1e9, 2),
                                  "frac": round(total_alg * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_frac_of_hbm_peak": {k: (round(alg[k] * Bl / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if v > 0 else None)
                                       for k, v in stage_ms.items()},
            "digest_per_rank": digest_all,
            "stage_ms_per_launch_lane_alone": ({k: round(v, 4) for k, v in alone_ms.items()} if alone_ms else None),
            "roofline_lane_alone": ({"kernel": dominant, "note": "same launch, measured after the timed region with the other lanes idle",
                                     "avg_launch_ms": round(alone_ms[dominant] / launches[dominant], 4),
                                     "achieved": round(dom_bytes / (alone_ms[dominant] / launches[dominant] * 1e-3) / 1e9, 2),
                                     "frac": round(dom_bytes / (alone_ms[dominant] / launches[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                    if alone_ms and alone_ms.get(dominant, 0) > 0 else None),
            # Hamming throughput of the N x N match as N_q * N_t * 256 bit comparisons per second
            "hamming_bitops_per_s": round(fps * mean_kp * mean_kp * 256.0, 1),
        }
        if gated:
            out["gated_match"] = gated
        n_cpu = args.cpu_frames if args.cpu_frames >= 0 else (100 if args.config == "c2" else 12)
        if world == 1 and n_cpu > 0:
            out["cpu_baseline"] = cpu_baseline(synth, cfg, n_cpu)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            if args.cpu_all_cores > 0:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, args.cpu_all_cores, 24 if args.config != "c5" else 3)
        print(json.dumps(out), flush=True)

    shard.finalize()


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:  # a failed rank must not sit in the process group's teardown
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
