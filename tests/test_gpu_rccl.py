"""The RCCL code path of the sharded bench on the ONE GPU a test box has: a one-rank "nccl" (= RCCL on ROCm) process group runs the
same barrier / all_reduce(MAX) / all_gather calls on device tensors as the N-rank job (amos-slam_amd/shard.py), first inside this
process, then as `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (a child process: nothing that has touched
the GPU is ever replaced by another program)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_collectives(gpu_lib, monkeypatch):
    import torch
    import torch.distributed as dist
    shard = importlib.import_module("amos_slam_amd.shard")
    assert not dist.is_initialized()
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port()))):
        monkeypatch.setenv(k, v)
    assert shard.launched()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        assert shard.init("nccl", dev) == (0, 1)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        shard.barrier()
        assert shard.max_over_ranks(1.25, "cuda:0") == 1.25                      # all_reduce(MAX) on a device tensor
        digest = [3.0, 1000.0, 4242424242.0, 512.0]
        assert shard.gather_digests(digest, "cuda:0") == [digest]                # all_gather on device tensors
        t = torch.arange(8, dtype=torch.float32, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert t.tolist() == list(range(8))
    finally:
        shard.finalize()
    assert not dist.is_initialized()


@pytest.mark.gpu
def test_bench_under_the_launcher_with_one_rank(gpu_lib):
    """What the driver runs for N > 1, at N = 1: the launcher environment makes bench.py join an RCCL group."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "c2", "--steps", "3", "--warmup", "1", "--batch", "32", "--cpu-frames", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 1 and d["value"] > 0 and len(d["digest_per_rank"]["rows"]) == 1
    assert d["process_group"] == {"backend": "nccl", "world_size": 1}
