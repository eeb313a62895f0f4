"""bench.py's launcher logic, without a GPU: a rank count that cannot be had is an error, never a silently smaller run."""
import os
import subprocess
import sys

import pytest

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


def test_roofline_fractions_keep_one_meaning_each():
    """bench.py: `algorithmic_over_peak` is always achieved / peak with the ALGORITHMIC bytes of SURVEY 8(d) (a yardstick against the
    reference's algorithm that may exceed 1 -- cfg4: neighbouring 4K rays share lines, the model counts every ray's loads); `frac` repeats it
    where it is a fraction (<= 1) and is null where it is not, so that nothing called a fraction of the roofline exceeds 1; `frac_footprint`
    (unique lines a launch touches + its output, / time / peak) is the bound; `frac_l2_fabric` is the counter traffic's fraction, or null
    without a committed pass.  None switches meaning with its value."""
    sys.path.insert(0, ROOT)
    import bench

    f = bench.roofline_fractions(7200.0, 78.3e9, 15.9)          # algorithmic below the peak
    assert f["frac"] == f["algorithmic_over_peak"] == 0.9 and 0.6 < f["frac_l2_fabric"] < 0.63 and f["frac_footprint"] is None
    f = bench.roofline_fractions(9365.0, 70.3e9, 22.19)         # cfg4_n1: algorithmic above the peak
    assert f["frac"] is None and f["algorithmic_over_peak"] > 1.0 and 0.39 < f["frac_l2_fabric"] < 0.41
    f = bench.roofline_fractions(9365.0, None, 22.19)           # ... and no counters
    assert f["frac"] is None and f["algorithmic_over_peak"] > 1.0 and f["frac_l2_fabric"] is None
    fp = {"grid2_lines": 100_000_000, "rows_lines": 150_000_000, "footprint_bytes": 250_000_000 * 128 + 2_000_000_000}
    f = bench.roofline_fractions(9365.0, 70.3e9, 22.19, fp)
    assert f["footprint_bytes"] == fp["footprint_bytes"] and 0.18 < f["frac_footprint"] < 0.20 and f["footprint_lines_by_array"] == {"grid2": 100_000_000, "rows": 150_000_000}
    assert f["footprint_frame_by_frame_bytes"] is None and f["frac_footprint_frame_by_frame"] is None   # no frame-by-frame pass for this workload
    assert all(v is None or v <= 1.0 for k, v in f.items() if k.startswith("frac"))
    fp["footprint_frame_by_frame_bytes"] = 4 * fp["footprint_bytes"]   # every line needed by four frames on average: between the two other figures
    f = bench.roofline_fractions(9365.0, 70.3e9 * 4, 22.19 * 4, fp)
    assert f["frac_footprint"] < f["frac_footprint_frame_by_frame"] < f["frac_l2_fabric"] <= 1.0


def test_committed_counters_cover_all_poses_of_every_bench_workload():
    """The numerators of the three rooflines: 16 poses each, the algorithmic-bytes field equal to the formula applied to the counters."""
    sys.path.insert(0, ROOT)
    import bench

    for wl, rays in (("cfg2", 1920 * 1080), ("cfg3", 1920 * 1080), ("cfg4", 3840 * 2160), ("fog", 1920 * 1080)):
        c = bench.load_counters(wl)
        assert c is not None and sorted(c["poses"], key=int) == [str(i) for i in range(16)], wl
        for p in c["poses"].values():
            assert p["rays"] == rays and p["algorithmic_bytes"] == bench.alg_bytes(p), wl
    # cfg3's rays are 1.8 x as long as cfg2's: what puts it at 0.54 x the headline's ray rate at the same efficiency per step
    steps = {wl: sum(p["steps"] for p in bench.load_counters(wl)["poses"].values()) / (16 * 1920 * 1080) for wl in ("cfg2", "cfg3")}
    assert 1.7 < steps["cfg3"] / steps["cfg2"] < 2.0
    # fog: long dense runs -- more than 30 dense samples per ray (cfg2: 4.5)
    fog = bench.load_counters("fog")["poses"].values()
    assert sum(p["hits"] for p in fog) / (16 * 1920 * 1080) > 30


def test_committed_traffic_belongs_to_one_kernel_source_and_launch_shape(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import json

    import bench

    fake = tmp_path / "profiles"
    fake.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "f" * 16)
    (fake / "r09_traffic_cfg3.json").write_text(json.dumps({"kernel_source_sha": "0" * 16, "frames_per_launch": 16, "hbm_bytes_per_launch": 1}))
    assert bench.committed_traffic("cfg3", 16) == (None, None)          # another source
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "0" * 16)
    assert bench.committed_traffic("cfg3", 16) == (1, os.path.join("profiles", "r09_traffic_cfg3.json"))
    assert bench.committed_traffic("cfg3", 64) == (None, None)          # another launch shape
    assert bench.committed_traffic("cfg4", 16) == (None, None)          # another workload


@pytest.mark.gpu
def test_footprint_pass_counts_the_lines_a_launch_touches(torch_gpu):
    """tools/footprint.py (bench.py's `footprint_bytes`): one launch of the diagnostics instantiation on the test-hook build with a line bitmap.
    Deterministic, at least the output, at most every line of the arrays a plain cfg2 launch reads (the level-9 grid and the rows), and no line
    of an array such a launch does not read (node words, records, the voxel grids: the inline cell words answer)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import footprint

    a = footprint.measure(["cfg2_small"])["cfg2_small"]
    b = footprint.measure(["cfg2_small"])["cfg2_small"]
    assert a == b
    rays = 4 * 640 * 360
    assert a["output_bytes"] == rays * 16 and a["footprint_bytes"] == a["output_bytes"] + 128 * sum(v for k, v in a.items() if k.endswith("_lines"))
    assert 10_000 < a["grid2_lines"] <= (512 ** 3 * 4) // 128 and 50_000 < a["rows_lines"] <= 1_499_569 * 8 * 64 // 128
    assert a["nodes_lines"] == 0 and a["records_lines"] == 0 and a["grid2_vox_lines"] == 0
    # frame by frame (an accel per pose): a frame touches no more lines than the launch of all four, the four frames together at least as many as
    # the launch (a line two poses see is counted twice) and less than four times as many
    lines = sum(v for k, v in a.items() if k.endswith("_lines"))
    assert 0 < a["frame_lines_min"] <= a["frame_lines_max"] <= lines
    assert a["footprint_bytes"] <= a["footprint_frame_by_frame_bytes"] < 4 * a["footprint_bytes"]
