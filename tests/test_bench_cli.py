"""bench.py's launcher logic, without a GPU: a rank count that cannot be had is an error, never a silently smaller run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(over)
    return env


def test_gpus_n_without_launcher_needs_n_devices_over_rccl():
    import torch

    if torch.cuda.device_count() >= 2:
        return  # a multi-GPU machine would really start the ranks
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_equal_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "refusing" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "refusing" in r.stderr


def test_kernel_source_sha_is_stable_and_matches_the_binding():
    sys.path.insert(0, ROOT)
    import bench

    a, b = bench.kernel_source_sha(), bench.kernel_source_sha()
    assert a == b and len(a) == 16
