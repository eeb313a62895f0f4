"""Rehearsal of bench.py's N > 1 path on ONE GPU: two (three) ranks share cuda:0, the gather goes through
gloo with host staging instead of RCCL, everything else -- interleaved partition, one batched launch per
step per rank, ring of in-flight gathers, un-permute on rank 0 -- is the production code path.  The JSON
line's `parity` field compares frames assembled on rank 0 with the oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,gather", [(2, "f32"), (3, "u8")])
def test_bench_multirank_rehearsal(torch_gpu, world, gather):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--laps", "1",
           "--backend", "gloo", "--gather", gather]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["steps"] == 2
    assert d["parity"]["pixels_not_bit_identical"] == 0 and d["parity"]["max_abs_drgba_vs_oracle"] == 0.0
    assert d["value"] > 0 and d["roofline"]["launches"] == 2


@pytest.mark.parametrize("gather,reserve", [("u8", 32), ("f32", 0)])
def test_bench_rccl_path_single_rank(torch_gpu, gather, reserve):
    """The RCCL calls of the N > 1 path executed for real (a one-GPU box allows one rank): process group on "nccl", partition with
    world 1, march on the CU-masked stream, dist.gather on the side stream, un-permute, all_reduce, barrier.  Assembled frames are
    compared with the oracle inside bench.py."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1", "--laps", "1", "--gather", gather,
           "--reserve-cus", str(reserve)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["reserved_cus"] == reserve and "RCCL gather" in d["config"]["partition"]
    assert d["parity"]["pixels_not_bit_identical"] == 0 and d["parity"]["frames_checked"] == 2
    assert d["value"] > 0 and d["roofline"]["launches"] == 3
