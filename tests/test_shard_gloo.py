"""CPU: the N > 1 path of bench.py (stream-per-rank sharding, max-over-ranks time, final gather)
with two gloo ranks.  Each rank runs the CPU oracle on its own small stream as a stand-in for the
GPU path (the sharding logic is what is under test)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("amos_slam_amd.shard")
    synth = importlib.import_module("amos_slam_amd.synth")
    import oracle_binding as ob
    r, w = shard.init("gloo")
    assert (r, w) == (rank, world)
    stream = shard.stream_for_rank(r)
    orc = ob.Oracle(n_features=300, n_levels=4)
    n_kp = 0
    for k in range(2):
        kps, desc = orc.extract(synth.frame(stream, k, 240, 320))
        n_kp += len(kps)
    shard.barrier()
    t = shard.max_over_ranks(1.0 + rank)  # rank 1 is "slower"
    digests = shard.gather_digests([float(stream), float(n_kp)])
    q.put((rank, t, digests))
    shard.finalize()


def test_two_rank_shard_and_gather(ob, synth):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = []
    for stream in (0, 1):  # what a single process computes for both streams
        orc = ob.Oracle(n_features=300, n_levels=4)
        expect.append([float(stream), float(sum(len(orc.extract(synth.frame(stream, k, 240, 320))[0]) for k in range(2)))])
    for rank, t, digests in results:
        assert t == 2.0  # max over ranks
        assert digests == expect  # rank order, identical on every rank
    assert expect[0][1] != expect[1][1] or expect[0][0] != expect[1][0]


def _bench(args, extra_env=None, timeout=300):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment spawns torch.distributed.run itself (before torch or a
    GPU is touched) and relays rank 0's single JSON line; --dry-run swaps the GPU work for a sleep so that the
    launcher, the barrier, the max-over-ranks time and the digest gather run here on two gloo ranks."""
    out, lines = _bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True and d["value"] is None
    assert len(d["digest_per_rank"]) == 2 and d["digest_per_rank"][0] != d["digest_per_rank"][1]
    assert [row[0] for row in d["digest_per_rank"]] == [0.0, 1.0]  # stream r on rank r, gathered in rank order
    assert d["ms_per_step"] >= 2.0  # the slowest rank (2 ms per step) sets the time


def test_bench_reports_a_failed_rank():
    out, lines = _bench(["--gpus", "2", "--dry-run", "--steps", "2"], {"AMOS_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0 and not lines
    assert "exited with code" in out.stderr


def test_bench_rejects_a_mismatched_launcher_environment():
    out, lines = _bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and not lines


def test_single_process_helpers(pkg):
    import importlib
    shard = importlib.import_module("amos_slam_amd.shard")
    assert shard.max_over_ranks(0.5) == 0.5
    assert shard.gather_digests([1.0, 2.0]) == [[1.0, 2.0]]
    assert shard.stream_for_rank(3) == 3
