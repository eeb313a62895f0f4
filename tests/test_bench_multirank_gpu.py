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

import hooks

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


def test_bench_starts_its_own_ranks(torch_gpu):
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py starts the two ranks itself (fresh child processes, the parent never
    touches the GPU) and relays rank 0's line; a line that reports another rank count is an error, not a one-GPU number."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--laps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["pixels_not_bit_identical"] == 0 and d["parity"]["frames_checked"] >= 3


def test_bench_launcher_repeats_a_failed_run_conservatively_and_kills_a_wedged_one(torch_gpu, fake_rccl):
    """The launcher inside bench.py walks a ladder: (i) ranks that fail with reserved compute units (test hook) are started again
    WITHOUT the reservation and the C-ABI gather then works -- the second rung; (ii) ranks whose C-ABI gather cannot bind its library
    at all fall through the abi rungs to torch.distributed's gather -- the last rung; the line names every failed attempt; (iii) ranks
    that do not finish within --launch-timeout are killed -- their process group, nothing else -- and the exit code is not 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MNV_LIB_PATH"] = hooks.HOOKS_LIB
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--laps", "1"]
    # (i) the two-step ladder over the stand-in
    e1 = dict(env, MNV_RCCL_LIBRARY=fake_rccl)
    r = subprocess.run(cmd + ["--test-fail-reserved"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e1)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and "[default] ended with code" in d["launch"] and "repeated with --reserve-cus 0 --one-march-stream" in d["launch"]
    assert "mnv_gather_tiles" in d["config"]["partition"] and d["config"]["reserved_cus"] == 0 and d["parity"]["pixels_not_bit_identical"] == 0
    # (ii) down to the last rung
    e2 = dict(env, MNV_RCCL_LIBRARY="/nonexistent/librccl.so")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=e2)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["launch"].count("ended with code") == 3 and "repeated with --gather-via torch" in d["launch"] and "gloo" in d["config"]["partition"]
    assert d["parity"]["pixels_not_bit_identical"] == 0
    # (iii) the watchdog
    r = subprocess.run(cmd + ["--launch-timeout", "3"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)   # (no rung can finish in 3 s)
    assert r.returncode != 0 and "was killed" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_multirank_through_the_c_abi_gather(torch_gpu, fake_rccl, world):
    """bench.py's N > 1 path with the C-ABI gather (mnv.Comm -> mnv_gather_tiles) instead of the gloo staging: CU-masked march streams,
    one stream per ring slot, root-relieving partition, per-slot side streams, un-permute on rank 0.  The ranks share the GPU and
    tests/shim/fake_rccl.cpp stands in for RCCL's transport (MNV_RCCL_LIBRARY); bench.py starts the ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MNV_RCCL_LIBRARY"] = fake_rccl
    env["MNV_LIB_PATH"] = hooks.HOOKS_LIB
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "3", "--warmup", "1", "--laps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and "mnv_gather_tiles" in d["config"]["partition"] and d["config"]["march_streams"] == 2
    assert d["parity"]["pixels_not_bit_identical"] == 0 and d["parity"]["frames_checked"] >= 3 and d["roofline"]["approximate"] is True


def test_bench_refuses_a_rank_count_it_cannot_have(torch_gpu):
    """Over RCCL every rank needs its own GPU: on a one-GPU box `--gpus 2` must fail, not print a one-GPU line."""
    if torch_gpu.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr


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
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["reserved_cus"] == reserve and "RCCL (mnv_gather_tiles) gather" in d["config"]["partition"]
    assert d["parity"]["pixels_not_bit_identical"] == 0 and d["parity"]["frames_checked"] >= 3
    assert d["value"] > 0 and d["roofline"]["launches"] == 3
    # the first-contact report: what every rank measured, next to the step the one-GPU emulation predicts for the real world size
    (pr,) = d["per_rank"]
    assert pr["rank"] == 0 and pr["march_ms"] > 0 and pr["gather_ms"] > 0 and pr["unpermute_ms"] > 0 and pr["rccl_version"] > 0
    assert pr["reserved_cus"] == reserve and pr["cu_mask_in_effect"] == (reserve > 0) and (reserve == 0 or pr["enabled_cus"] < 256)
    assert "predicted" in d      # None here: the emulation has no row for one rank


def test_scale_preflight_with_one_rank(torch_gpu):
    """tools/scale_preflight.py -- peer access, communicator, gathers at the bench's message sizes, all-gather between all pairs, CU-masked
    stream under N processes -- on the one GPU of this box: every item passes over real RCCL (self send / receive)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_preflight.py"), "--gpus", "1", "--json"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] and d["ranks"] == 1
    names = [it["name"] for it in d["results"][0]["items"]]
    assert names == ["peer access", "communicator", "gather 4.1 MB", "gather 16.6 MB", "all-gather", "masked stream"]
