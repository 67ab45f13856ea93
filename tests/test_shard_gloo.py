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


def test_single_process_helpers(pkg):
    import importlib
    shard = importlib.import_module("amos_slam_amd.shard")
    assert shard.max_over_ranks(0.5) == 0.5
    assert shard.gather_digests([1.0, 2.0]) == [[1.0, 2.0]]
    assert shard.stream_for_rank(3) == 3
