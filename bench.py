#!/usr/bin/env python3
"""bench.py -- Mrays/s of the N3Tree march on BASELINE.json's headline configuration.

Workload (BASELINE.json configs[1]): synthetic depth-10 SH-9 N3Tree
(spherical shell, 1,499,569 chunks), 1920x1080, fx = fy = 1600, a 16-pose orbit at radius 2.6 /
elevation 20 deg, CLI-default RenderOptions.  One STEP = one pass over that batch of 16 poses
(33,177,600 primary rays).  Inputs (tree, cameras) are resident in HBM before the timed region.

N > 1 (one process per GPU, torch.distributed over RCCL): the tree is replicated, each frame is
cut into interleaved macro tiles (rank = tile % world), every rank renders its tiles into a
compact buffer with ONE batched launch per step, and the tiles are gathered to rank 0 over xGMI
(dist.gather -> RCCL send/recv) and un-permuted into the frames there.  The gather of step k
overlaps the render of step k+1 (ring of 2 buffers).  Total work per step is fixed, so scaling is
"strong".  What is gathered is, by default, the reference's own output format (RGBA8): at 8 GPUs the
root receives 7/8 of every frame over 7 xGMI links, and 33 MB float frames would make the step
link-bound (about 1.3 ms against 0.6 ms of rendering); --gather f32 selects that variant.  After the
timed region rank 0 checks assembled frames against the oracle (u8 frames byte for byte).

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline      algorithmic bytes of the dominant kernel / its HIP-event launch time vs 8 TB/s
  cpu_baseline  the CPU oracle (a port of the reference's march; the reference has no CPU
                renderer) timed on this host's cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

# the CPU baseline's OpenMP threads: one per physical core, pinned and spread (must be in the environment before an OpenMP runtime loads).
# Once a runtime has loaded, the main thread is bound to ITS place and sched_getaffinity no longer tells how many CPUs the process may use:
# the count is taken here and handed to the oracle's thread-count logic (oracle/mnv_oracle.py).
# ONLY in a one-rank run: with proc-bind every process's main thread -- the one that issues the launches -- is bound to the FIRST place, and N
# ranks of a multi-GPU run would share one core.


def _single_rank_run():
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        return False
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            return sys.argv[i + 1] == "1"
        if a.startswith("--gpus="):
            return a.split("=", 1)[1] == "1"
    return True


if _single_rank_run():
    if hasattr(os, "sched_getaffinity"):
        os.environ.setdefault("MNV_ORACLE_CPUS", str(len(os.sched_getaffinity(0))))
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W, H, FX = 1920, 1080, 1600.0
N_POSES = 16
N_FRAMES = 16  # frames per step: N_POSES x --laps
MACRO_W, MACRO_H = 64, 24            # 30 x 45 = 1350 macro tiles: per-rank launch times within 3 % of each other at world 8
                                     # (tools/partition_balance.py: 128x120 tiles leave the slowest rank 19 % behind)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8.0 TB/s spec
COUNTERS_JSON = os.path.join(ROOT, "tests", "golden", "cfg2_counters.json")
COUNTERS3_JSON = os.path.join(ROOT, "tests", "golden", "cfg3_counters.json")   # sections "cfg3" (1920x1080) and "cfg4" (3840x2160)


def _latest_traffic_json(workload="cfg2"):
    """profiles/rNN_traffic[_cfg3|_cfg4].json of the latest round: HBM bytes per launch from the committed PMC passes (valid for one kernel
    source hash and one launch shape)."""
    import glob

    suffix = "" if workload == "cfg2" else "_" + workload
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic{suffix}.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", f"r02_traffic{suffix}.json")


TRAFFIC_JSON = _latest_traffic_json()


def committed_traffic(workload, frames_per_launch):
    """(HBM bytes per launch, file) from the committed PMC passes if they belong to THIS kernel source and launch shape, else (None, None).
    PMC counters cannot be read from inside this process."""
    path = _latest_traffic_json(workload)
    if not os.path.exists(path):
        return None, None
    tj = json.load(open(path))
    if tj.get("frames_per_launch") == frames_per_launch and tj.get("kernel_source_sha") == kernel_source_sha():
        return tj["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def roofline_fractions(achieved_gbs, traffic, avg_ms, footprint=None):
    """The fractions of a roofline object, each with ONE meaning in every object of the line (DESIGN.md section 7 says which is the bound):
    `algorithmic_over_peak` = `achieved` / `peak` with `achieved` = the ALGORITHMIC bytes of SURVEY 8(d) / kernel time -- what the reference's
    algorithm would have to move for these frames (every ray's loads counted, its root-restart child words included, no sharing between
    neighbouring rays).  This kernel does not execute those loads (lookup grids, inline cell words, brick records): a yardstick against the
    reference's algorithm, NOT a bound -- it can exceed 1.  `frac` repeats it where it is below 1 (the headline) and is null where it is not:
    no field called a fraction of the roofline exceeds 1.
    `frac_footprint` = `footprint_bytes` / kernel time / peak: the unique 128-byte lines the launch's loads touch (tools/footprint.py: one pass
    of the diagnostics instantiation with a line bitmap) + its output.  Every such line has to arrive from memory at least once when the working
    set is far beyond L2 + Infinity Cache, so this IS a floor under the HBM bytes and a bound: <= 1.
    `frac_l2_fabric` = `traffic` / kernel time / peak with `traffic` = 2 x FETCH_SIZE + WRITE_SIZE of the committed PMC passes for these kernel
    sources: requests the L2 sent to the fabric.  On gfx950 those include Infinity-Cache (MALL) hits, so this is an UPPER estimate of the DRAM
    bytes, bracketed from below by the footprint; null when no committed pass belongs to this source and launch shape.
    `frac_footprint_frame_by_frame` (cfg2, cfg3) = the same with the lines counted FRAME BY FRAME (each frame's unique lines + its output, summed
    over the launch's frames): what must come in when nothing survives in the caches from one frame to the next -- true of the 32 MiB of L2
    (a frame touches ~0.2 GB), not necessarily of the 256 MiB Infinity Cache; between `frac_footprint` and `frac_l2_fabric`."""
    over = achieved_gbs / HBM_PEAK_GBS
    fabric = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
    out = {"frac": round(over, 5) if over <= 1.0 else None, "algorithmic_over_peak": round(over, 5),
           "frac_l2_fabric": round(fabric, 5) if fabric is not None else None,
           "traffic_is": "L2-to-fabric bytes (2 x FETCH_SIZE + WRITE_SIZE): includes Infinity-Cache hits, an upper estimate of DRAM bytes"}
    if footprint:
        fb = footprint["footprint_bytes"]
        out.update({"footprint_bytes": int(fb), "frac_footprint": round(fb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "footprint_lines_by_array": {k[:-6]: v for k, v in footprint.items() if k.endswith("_lines")}})
        ff = footprint.get("footprint_frame_by_frame_bytes")
        out.update({"footprint_frame_by_frame_bytes": int(ff) if ff else None,
                    "frac_footprint_frame_by_frame": round(ff / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ff else None})
    else:
        out.update({"footprint_bytes": None, "frac_footprint": None})
    return out


def measured_footprints(workloads):
    """tools/footprint.py in a child process on the test-hook build (the shipped library has no diagnostics hook): {workload: dict} or {}."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import footprint as fp_tool
        return fp_tool.measure(workloads)
    except Exception as e:  # a diagnostic pass must not cost the line
        print(f"bench.py: footprint pass skipped: {type(e).__name__}: {e}", file=sys.stderr)
        return {}


def load_counters(workload="cfg2"):
    """Per-pose integer work counters of the CPU restatement, committed with the fixtures."""
    path = COUNTERS_JSON if workload == "cfg2" else os.path.join(ROOT, "tests", "golden", "fog_counters.json") if workload == "fog" else COUNTERS3_JSON
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    return d if workload in ("cfg2", "fog") else d.get(workload)


def kernel_source_sha():
    """Identity of the device code a profile belongs to -- the .hip / .h files of csrc/ (host-only .cpp files such as the RCCL binding do
    not enter) -- stored with the PMC numbers by tools/make_traffic_json.py."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "mega-nerf-viewer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def alg_bytes(c, basis_dim=9):
    # sum_rays [16 + sum_steps (4 d_s + 2 + hit_s * 6 * basis_dim)]
    return 16 * c["rays"] + 4 * c["levels"] + 2 * c["steps"] + 6 * basis_dim * c["hits"]


def gpu_count_without_hip():
    """GPUs this process can use, from the KFD topology in sysfs, or None when that cannot be read: nodes with SIMDs whose properties are
    readable (a container that is given one GPU of eight sees ten nodes and may read three) and whose render node exists in /dev/dri, at most
    as many as HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list.  The launcher below must never initialise HIP (it starts the ranks as children
    and this pool forbids a GPU-initialised process to exec): torch.cuda.device_count() may fall back to hipGetDeviceCount on builds without
    amdsmi, so it is not asked."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):
        return 0  # no amdgpu compute driver on this machine: no GPUs
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "0"))
        except (OSError, ValueError):
            continue  # not ours to read: not ours to use
        if minor > 0 and os.path.isdir("/dev/dri") and not os.path.exists(f"/dev/dri/renderD{minor}"):
            continue
        n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n, backend, timeout_s):
    """One process per GPU via torch.distributed.run, started from a parent that never initialises HIP.  The ranks run in their own
    process group under a watchdog: a run that exceeds `timeout_s` (a wedged collective) is killed -- that group, by id -- and, like a run
    that failed, repeated down a short ladder of more conservative configurations (no CU reservation; float tiles; torch.distributed's
    gather instead of mnv_gather_tiles); the line then carries what failed and what it ran with in `launch`."""
    import signal
    import socket
    import subprocess

    have = gpu_count_without_hip()
    if backend == "nccl" and have is not None and have < n:
        print(f"bench.py: --gpus {n} over RCCL needs {n} GPUs, this node shows {have} "
              f"(--backend gloo rehearses the {n}-rank path on fewer GPUs)", file=sys.stderr)
        return 2

    def attempt(extra):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
            rc = p.returncode
        except subprocess.TimeoutExpired:
            # the launcher, its ranks and whatever they started: the descendants of the process started above, listed BEFORE anything dies
            # (a rank whose launcher is gone is re-parented and can no longer be found), each ended by its own pid, then the session's group
            victims = []
            try:
                import psutil

                victims = psutil.Process(p.pid).children(recursive=True)
            except Exception:
                pass
            for v in victims:
                try:
                    v.kill()
                except Exception:
                    pass
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            try:
                out, _ = p.communicate(timeout=15)
            except subprocess.TimeoutExpired:   # something still holds the pipe: do not wait for it
                p.stdout.close()
                out = ""
            rc = -signal.SIGKILL
            print(f"bench.py: the {n}-rank run did not finish within {timeout_s} s and was killed", file=sys.stderr)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        for ln in out.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        return rc, lines

    # The ladder: every rung changes ONE thing a first contact with an N-GPU node can trip over, the last rung leaves libmnv's gather altogether.
    # A rung is skipped when the command line already pins the option it would change.
    ladder = [([], "as asked"),
              (["--reserve-cus", "0", "--one-march-stream"], "no compute units reserved for RCCL, one march stream"),
              (["--gather", "f32", "--reserve-cus", "0", "--one-march-stream"], "float RGBA tiles instead of RGBA8, no reservation, one march stream"),
              (["--gather-via", "torch", "--one-march-stream"], "torch.distributed's gather instead of mnv_gather_tiles, one march stream")]
    failed = []
    rc, lines = 4, []
    for extra, what in ladder:
        if extra and any(o in sys.argv for o in extra if o.startswith("--") and o != "--one-march-stream"):
            continue
        rc, lines = attempt(extra)
        if rc == 0 and lines:
            break
        failed.append(f"[{' '.join(extra) or 'default'}] ended with code {rc}")
        print(f"bench.py: attempt {len(failed)} ({what}) {failed[-1]}", file=sys.stderr)
    note = None
    if failed and rc == 0 and lines:
        note = "; ".join(failed) + f"; this line: repeated with {' '.join(extra)} ({what})"
    if rc == 0 and lines:
        d = json.loads(lines[-1])
        if d.get("n_gpus") != n:
            print(f"bench.py: asked for {n} ranks, the line reports {d.get('n_gpus')}", file=sys.stderr)
            return 3
        if note:
            d["launch"] = note
        print(json.dumps(d), flush=True)
        return 0
    return rc or 4


MFMA_PEAK_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak (no sparsity)


def extras_cfg3_cfg4(mnv, cases, orc, torch, dev, opt, steps, footprints=None):
    """Secondary numbers of the default run: BASELINE.json configs[2] (merged-octree stand-in, 1920x1080) and configs[3] on ONE GPU
    (the same tree at 3840x2160) -- the 7.2 M-chunk terrain of SURVEY.md 8(d), 16 oblique poses per launch.  Each with its own roofline
    (algorithmic bytes from the oracle's counters of two poses rendered in this run) and a bit-exact check of those frames."""
    t_setup = time.time()
    tree = cases.make_tree(mnv, cases.CFG3_FULL)
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    info = mnv.accel_info(tree.accel)
    setup_s = time.time() - t_setup
    out = {}
    for name, (w, h) in (("cfg3", (1920, 1080)), ("cfg4_n1", (3840, 2160))):
        cams = [cases.cfg3_camera(mnv, pose, w, h, fx=1400.0 * w / 1920) for pose in range(N_POSES)]
        frames = torch.empty((N_POSES, h, w, 4), dtype=torch.float32, device=dev)
        for _ in range(2):  # two untimed launches: the first one of a 5 GB accel runs cold (page tables, MALL)
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=frames)
        torch.cuda.synchronize(dev)
        mnv.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=frames)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        kern_ms, launches = mnv.take_timing()
        mnv.set_timing(False)
        chk = (0, 5)
        committed = load_counters("cfg3" if name == "cfg3" else "cfg4")
        fresh_bytes, n_bad, max_d, stale = [], 0, 0.0, []
        for i in chk:
            r = orc.render(ot, cams[i].c, opt)
            c = r["counters"].as_dict()
            fresh_bytes.append(alg_bytes(c))
            if committed and any(committed["poses"].get(str(i), {}).get(k) != x for k, x in c.items()):
                stale.append(i)
            gpu = frames[i].cpu().numpy()
            n_bad += int((gpu.view(np.uint32) != r["rgba"].view(np.uint32)).any(axis=-1).sum())
            max_d = max(max_d, float(np.abs(gpu - r["rgba"]).max()))
        if stale:
            raise SystemExit(f"bench.py: tests/golden/cfg3_counters.json [{name}] disagrees with the oracle's counters for poses {stale}")
        # numerator: all 16 poses of the launch (committed counters, two of them re-derived above); without the file, the two fresh ones
        per_launch = (float(np.sum([alg_bytes(c) for c in committed["poses"].values()])) if committed and len(committed["poses"]) == N_POSES
                      else float(np.mean(fresh_bytes)) * N_POSES)
        avg_ms = kern_ms / max(1, launches)
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = committed_traffic("cfg3" if name == "cfg3" else "cfg4", N_POSES)
        rl = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        rl.update(roofline_fractions(achieved, traffic, avg_ms, (footprints or {}).get("cfg3" if name == "cfg3" else "cfg4")))
        rl.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": "march_accel_kernel<9,256,0,true>", "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                   "algorithmic_bytes_per_launch": int(per_launch), "counter_poses": N_POSES if committed else list(chk), "counters_rechecked_poses": list(chk) if committed else []})
        out[name] = {"value": round(N_POSES * w * h * steps / el / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
                     "resolution": f"{w}x{h}", "frames_per_launch": N_POSES, "roofline": rl,
                     "parity": {"frames_checked": len(chk), "pixels_not_bit_identical": n_bad, "max_abs_drgba_vs_oracle": max_d}}
        del frames
    workload = (f"depth-11 SH9 anisotropic 4x2-brick terrain N3Tree ({tree.capacity:,} chunks, {tree.capacity * 8 * 28 * 2 / 1e9:.2f} GB of voxel rows), "
                f"packed accel {info['device_bytes'] / 1e9:.2f} GB with a level-{info['grid2_level']} lookup grid, 16 oblique poses per launch")
    for v in out.values():
        v["workload"] = workload
    out["cfg3"]["setup_s"] = round(setup_s, 2)
    try:
        out["cfg5_cfg3"] = extras_cfg5_cfg3(mnv, cases, torch, dev, tree)
    except Exception as e:  # a secondary number must not cost the line
        out["cfg5_cfg3"] = {"error": f"{type(e).__name__}: {e}"}
    del tree
    torch.cuda.empty_cache()
    return out


FOG_COUNTERS_JSON = os.path.join(ROOT, "tests", "golden", "fog_counters.json")


def extras_fog(mnv, cases, orc, torch, dev, opt, steps, footprints=None):
    """A second distribution for the headline's kernel (long dense runs): cases.FOG_TREE -- the cfg2 generator with a thick shell of thin
    density, 33 dense samples in 44 steps per ray against cfg2's 4.5 in 21 -- under the cfg2 cameras at 1920x1080, 16 poses per launch.
    Its own roofline (numerator: the committed 16-pose counters, one pose re-derived here) and its own bit-exact frame."""
    t_setup = time.time()
    tree = cases.make_tree(mnv, cases.FOG_TREE)
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    info = mnv.accel_info(tree.accel)
    setup_s = time.time() - t_setup
    w, h = W, H
    cams = [cases.cfg2_camera(mnv, pose, w, h, FX) for pose in range(N_POSES)]
    frames = torch.empty((N_POSES, h, w, 4), dtype=torch.float32, device=dev)
    for _ in range(2):
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=frames)
    torch.cuda.synchronize(dev)
    mnv.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=frames)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    kern_ms, launches = mnv.take_timing()
    mnv.set_timing(False)
    committed = json.load(open(FOG_COUNTERS_JSON)) if os.path.exists(FOG_COUNTERS_JSON) else None
    pose = 3
    r = orc.render(ot, cams[pose].c, opt)
    c = r["counters"].as_dict()
    if committed and any(committed["poses"].get(str(pose), {}).get(k) != x for k, x in c.items()):
        raise SystemExit(f"bench.py: tests/golden/fog_counters.json disagrees with the oracle's counters for pose {pose}")
    gpu = frames[pose].cpu().numpy()
    n_bad = int((gpu.view(np.uint32) != r["rgba"].view(np.uint32)).any(axis=-1).sum())
    max_d = float(np.abs(gpu - r["rgba"]).max())
    have_all = bool(committed) and len(committed["poses"]) == N_POSES
    per_launch = float(np.sum([alg_bytes(p) for p in committed["poses"].values()])) if have_all else float(alg_bytes(c)) * N_POSES
    dense_per_ray = (sum(p["hits"] for p in committed["poses"].values()) / sum(p["rays"] for p in committed["poses"].values())) if have_all else c["hits"] / c["rays"]
    steps_per_ray = (sum(p["steps"] for p in committed["poses"].values()) / sum(p["rays"] for p in committed["poses"].values())) if have_all else c["steps"] / c["rays"]
    avg_ms = kern_ms / max(1, launches)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9
    traffic, traffic_source = committed_traffic("fog", N_POSES)
    rl = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    rl.update(roofline_fractions(achieved, traffic, avg_ms, (footprints or {}).get("fog")))
    rl.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": "march_accel_kernel<9,256,0,true>", "avg_launch_ms": round(avg_ms, 5), "launches": launches,
               "algorithmic_bytes_per_launch": int(per_launch), "counter_poses": N_POSES if have_all else [pose], "counters_rechecked_poses": [pose] if committed else []})
    value = N_POSES * w * h * steps / el / 1e6
    out = {"value": round(value, 2), "unit": "Mrays/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps, "resolution": f"{w}x{h}", "frames_per_launch": N_POSES,
           "dense_samples_per_ray": round(dense_per_ray, 2), "steps_per_ray": round(steps_per_ray, 2),
           "dense_samples_per_s": round(value * 1e6 * dense_per_ray, 0), "roofline": rl,
           "parity": {"frames_checked": 1, "pixels_not_bit_identical": n_bad, "max_abs_drgba_vs_oracle": max_d},
           "workload": (f"fog: depth-9 SH9 thick shell N3Tree ({tree.capacity:,} chunks, {tree.capacity * 8 * 28 * 2 / 1e9:.2f} GB of voxel rows; half-thickness 0.06, sigma U(5,40)), "
                        f"packed accel {info['device_bytes'] / 1e9:.2f} GB with a level-{info['grid2_level']} lookup grid, the cfg2 orbit, 16 poses per launch"),
           "setup_s": round(setup_s, 2)}
    del frames, tree
    torch.cuda.empty_cache()
    return out


def extras_cfg5(mnv, cases, torch, dev, tree, frames_each=6):
    """Secondary numbers for BASELINE.json configs[4] (dynamic refinement + guided sampling with the per-sample network fused into the march)
    on the headline tree at 1920x1080: the guided-sampling frame as one kernel (device time per frame from HIP events, bit-exact against the
    four-step path it replaces), and whole frames with BOTH switches on through the renderer (wall time per frame, tree edits included)."""
    import mlp_cases

    w, h = W, H
    v = tree.host_view()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 32
    desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 4, 2
    for i in range(3):
        g.min_position[i], g.range[i] = -1.0, 2.0
    cams = [cases.cfg2_camera(mnv, p, w, h, FX) for p in range(8)]
    out = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for c in cams[:2]:
        mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=out)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        for c in cams:
            mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=out)
    e1.record()
    torch.cuda.synchronize(dev)
    guided_ms = e0.elapsed_time(e1) / (reps * len(cams))
    # the same frames with three in flight on HIP streams (the entry point is re-entrant across streams, like the plain march): a frame's
    # tail -- workgroups that have run dry while the longest tiles finish -- runs under the next frame's start.  Throughput, not latency.
    in_flight = 3
    sts = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
    outs = torch.empty((len(cams), h, w, 4), dtype=torch.float32, device=dev)
    one_stream = torch.empty_like(outs)
    for i, c in enumerate(cams):
        mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=one_stream[i])
    torch.cuda.synchronize(dev)

    def lap():
        for i, c in enumerate(cams):
            mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=outs[i], stream=sts[i % in_flight].cuda_stream)

    lap()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        lap()
    torch.cuda.synchronize(dev)
    guided_flight_ms = (time.perf_counter() - t0) / (reps * len(cams)) * 1e3
    flight_bad = int((outs.view(torch.int32) != one_stream.view(torch.int32)).any(dim=-1).sum().item())
    del outs, one_stream
    # the same frame through the four kernels the fused one replaces: sample march -> compaction -> network -> composite
    n_px, dd = w * h, v.data_dim
    num = torch.zeros(n_px, dtype=torch.int16, device=dev)
    guided = torch.zeros((n_px, 32, 4), dtype=torch.float32, device=dev)
    clusters = torch.zeros((n_px, 32), dtype=torch.int16, device=dev)
    offsets = torch.empty(n_px, dtype=torch.int64, device=dev)
    cap = 16_000_000
    z = torch.empty(cap, dtype=torch.float32, device=dev)
    rows = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    rcl = torch.empty(cap, dtype=torch.int16, device=dev)
    values = torch.empty((cap, dd + 1), dtype=torch.float32, device=dev)
    ref = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    cam = cams[3]
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref)
    out.fill_(float("nan"))
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out, sample_counter=counter)
    torch.cuda.synchronize(dev)
    n_bad = int((out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1).sum().item())
    evals = int(counter.item())
    flops_per_eval = 2 * (32 * 64 + 64 * 64 + 64 * 32)   # the padded 32 -> 64 -> 64 -> 32 network the matrix cores run (27 inputs, 29 outputs used)
    del num, guided, clusters, offsets, z, rows, rcl, values, ref
    # both switches through the renderer: march + networks + composite + trackers in one kernel, vote, 4096 splits x 8 corners x 8 network samples, accel patch
    tree2 = cases.make_tree(mnv, cases.CFG2_TREE)
    r = mnv.Renderer()
    r.resize(w, h)
    r.set(tree2, tree2.capacity + 1_000_000)
    desc6 = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    from test_renderer_refine_gpu import make_grid
    r.set_model(desc6, mlp_cases.make_params(mnv, desc6, seed=21), make_grid(mnv))
    r.set_seed(7)
    o = r.options
    o.use_splitting, o.use_guided_sampling, o.max_depth, o.split_batch_size, o.samples_per_corner, o.max_guided_samples = True, True, 12, 4096, 8, 32
    ts, st = [], None
    for f in range(frames_each + 2):
        c = cases.cfg2_camera(mnv, f % N_POSES)
        m = c.c2w
        r.set_camera(tuple(m[9:12]), tuple(m[6:9]), fx=FX)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        st = r.render()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    both_ms = float(np.median(ts[2:]))
    res = {"guided_ms_per_frame": round(guided_ms, 4), "guided_Mrays_per_s": round(w * h / guided_ms / 1e3, 1), "network_evals_per_frame": evals,
           "network_evals_per_s": round(evals / (guided_ms * 1e-3), 0), "mfma_frac": round(evals * flops_per_eval / (guided_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 5),
           "mfma_note": "network flops of the padded 32-64-64-32 tiles / whole-kernel time / 2.5 PFLOP/s: the kernel is the march AND the network",
           "pixels_not_bit_identical_vs_four_step": n_bad,
           "guided_in_flight": {"frames_in_flight": in_flight, "ms_per_frame": round(guided_flight_ms, 4), "Mrays_per_s": round(w * h / guided_flight_ms / 1e3, 1),
                                "pixels_not_bit_identical_vs_one_stream": flight_bad},
           "both_ms_per_frame": round(both_ms, 4), "both_network_evals_per_frame": int(st["guided_samples"]), "both_added_per_frame": int(st["added"]), "both_fused": int(st["fused"]),
           "what": "cfg2 tree at 1920x1080: guided = mnv_render_guided_fused (one kernel per frame, 8 sub-modules, 64x2 network, quota 32), HIP events over 24 frames; "
                   "both = VolumeRenderer::render with use_splitting + use_guided_sampling (4096 splits x 8 corners x 8 samples per frame), wall time per frame"}
    del r, tree2
    torch.cuda.empty_cache()
    return res


def extras_cfg5_cfg3(mnv, cases, torch, dev, tree):
    """BASELINE.json configs[4] as SURVEY.md 8(d) defines it: cfg3's tree (the 7.2 M-chunk merged-octree stand-in) + the tiny MLP,
    max_guided_samples 128, samples_per_corner 8, 1920x1080.  (The `cfg5` object is the same frame kinds on the cfg2 shell at quota 32.)
    guided = mnv_render_guided_fused, one kernel per frame, eight sub-modules on the 4 x 2 grid over world y, z; checked bit for bit against the
    four kernels it replaces; the producers' share of time spent waiting for ring space from the kernel's own clocks; both = both switches
    through VolumeRenderer::render.  `tree` is moved to the device again with room to grow (it is not used afterwards)."""
    import mlp_cases
    w, h = W, H
    v = tree.host_view()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 128
    desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=4)
    mlp = mnv.Mlp(desc, params)
    g = cases.cfg3_cluster_grid(mnv)
    cams = [cases.cfg3_camera(mnv, p, w, h, fx=1400.0) for p in range(0, N_POSES, 2)]
    out = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for c in cams[:2]:
        mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=out)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 2
    e0.record()
    for _ in range(reps):
        for c in cams:
            mnv.render_guided_fused(tree.accel, c, opt, mlp, g, rgba=out)
    e1.record()
    torch.cuda.synchronize(dev)
    guided_ms = e0.elapsed_time(e1) / (reps * len(cams))
    # one frame through the four kernels the fused one replaces, bit for bit; the same frame with the kernel's clocks on
    cam = cams[2]
    n_px, dd = w * h, v.data_dim
    num = torch.zeros(n_px, dtype=torch.int16, device=dev)
    guided = torch.zeros((n_px, 128, 4), dtype=torch.float32, device=dev)
    clusters = torch.zeros((n_px, 128), dtype=torch.int16, device=dev)
    offsets = torch.empty(n_px, dtype=torch.int64, device=dev)
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total_guess = int(num.to(torch.int64).sum().item())
    z = torch.empty(total_guess + 1, dtype=torch.float32, device=dev)
    rows = torch.empty((total_guess + 1, 3), dtype=torch.float32, device=dev)
    rcl = torch.empty(total_guess + 1, dtype=torch.int16, device=dev)
    values = torch.empty((total_guess + 1, dd + 1), dtype=torch.float32, device=dev)
    ref = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref)
    at_quota = int((num >= 128).sum().item())
    out.fill_(float("nan"))
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out, sample_counter=counter)
    torch.cuda.synchronize(dev)
    n_bad = int((out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1).sum().item())
    evals = int(counter.item())
    del num, guided, clusters, offsets, z, rows, rcl, values, ref
    diag = torch.zeros(32, dtype=torch.int64, device=dev)
    mnv.accel_set_fused_diag(tree.accel, diag)
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out)
    torch.cuda.synchronize(dev)
    mnv.accel_set_fused_diag(tree.accel, None)
    d = [int(x) for x in diag.tolist()]
    faults = mnv.accel_fused_faults(tree.accel)
    ring_wait = d[9] / max(1, d[8])        # F2Diag: kProdRingWait / kProdTotal (csrc/mnv_guided_fused2.h)
    cons_busy = d[6] / max(1, d[7])        # kConsBusy / kConsTotal
    # both switches through the renderer at the survey's parameters
    r = mnv.Renderer()
    r.resize(w, h)
    cap0 = tree.capacity
    r.set(tree, cap0 + 1_000_000)
    r.set_model(desc, params, g)
    r.set_seed(7)
    o = r.options
    o.use_splitting, o.use_guided_sampling, o.max_depth, o.split_batch_size, o.samples_per_corner, o.max_guided_samples = True, True, 13, 4096, 8, 128
    ts, st = [], None
    for f in range(10):
        c = cases.cfg3_camera(mnv, f % N_POSES, w, h, fx=1400.0)
        m = c.c2w
        r.set_camera(tuple(m[9:12]), tuple(m[6:9]), up=(1.0, 0.0, 0.0), fx=1400.0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        st = r.render()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    res = {"guided_ms_per_frame": round(guided_ms, 4), "guided_Mrays_per_s": round(w * h / guided_ms / 1e3, 1), "network_evals_per_frame": evals,
           "rays_at_the_quota_of_128": at_quota, "pixels_not_bit_identical_vs_four_step": n_bad, "samples_equal_four_step": bool(evals == total), "watchdog_faults": int(faults),
           "producer_ring_wait_share": round(ring_wait, 4), "consumer_busy_share": round(cons_busy, 4),
           "both_ms_per_frame": round(float(np.median(ts[2:])), 4), "both_network_evals_per_frame": int(st["guided_samples"]), "both_added_per_frame": int(st["added"]),
           "both_fused": int(st["fused"]),
           "what": "SURVEY 8(d) cfg5: the cfg3 tree (7.2 M chunks, depth 11) at 1920x1080, eight 64x2 sub-modules on the 4 x 2 grid over world y, z, max_guided_samples 128, "
                   "samples_per_corner 8, split_batch_size 4096; guided = mnv_render_guided_fused (HIP events over 16 frames, 8 oblique poses), both = VolumeRenderer::render with "
                   "use_splitting + use_guided_sampling, wall time per frame; ring wait = share of the producers' time spent waiting for ring space (the kernel's own clocks)"}
    del r
    torch.cuda.empty_cache()
    return res


def extras_live_call(mnv, cases, orc, torch, dev, tree, cams, opt):
    """The reference's LITERAL per-frame call (src/renderer/cuda_renderer.cpp:68-142 without the GL shell): the image attachment cleared to the
    background and the depth attachment to 1e9 (:70-77), split_tracker.fill_(-1), sample_tracker.fill_(-1) (:97-98), then
    render_voxels(tree, cam, opt, image, depth, stream, to_split, to_sample, visited, track_visit = false, offscreen = false) (:141-142) --
    every frame passes the two tracker tensors, the depth image and the image, so what runs is the tracker instantiation of the march, not
    the plain one the `per_frame` / `ref_layout` objects time.  Three servers of the same call: mnv_render_voxels_ex on the reference's arrays
    (stateless), the same with mnv_set_tree_cache(1), and mnv_render_voxels_accel_visit_ex on the packed accel; each on one stream with a wait
    per frame (the viewer's glFinish, main.cpp:614), on one stream back to back, and with three frames in flight (three sets of buffers)."""
    import ctypes as C
    v = tree.host_view()
    cap = v.capacity
    w, h = W, H
    counts = torch.full((cap, 8), 8, dtype=torch.int16, device=dev)   # (the reference leaves this array uninitialised, n3tree.cpp:235-241)
    dv = tree.device_view()
    tv = mnv.TreeView()
    C.memmove(C.byref(tv), C.byref(dv), C.sizeof(tv))
    tv.sample_counts = counts.data_ptr()
    o = mnv.RenderOptions()
    C.memmove(C.byref(o), C.byref(opt), C.sizeof(o))
    bg8 = int(o.background_brightness * 255)
    clear_word = (255 << 24 | bg8 << 16 | bg8 << 8 | bg8) - (1 << 32)   # RGBA8 bytes (bg, bg, bg, 255) as one little-endian int32
    K = 3
    sets = [dict(image=torch.empty((h, w, 4), dtype=torch.uint8, device=dev), depth=torch.empty((h, w), dtype=torch.float32, device=dev),
                 split=torch.empty((h * w, 3), dtype=torch.float32, device=dev), sample=torch.empty((h * w, 3), dtype=torch.float32, device=dev)) for _ in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]

    def frame(server, i, j, st):
        b = sets[j]
        with torch.cuda.stream(st):
            b["image"].view(torch.int32).fill_(clear_word)   # glClearNamedFramebufferfv(..., GL_COLOR, 0, {bg, bg, bg, 1})
            b["depth"].fill_(1e9)
            b["split"].fill_(-1)
            b["sample"].fill_(-1)
        if server == "accel":
            mnv.render_voxels_accel_visit(tree.accel, cams[i], o, None, None, rgba8=b["image"], split_track=b["split"], sample_track=b["sample"], sample_counts=counts,
                                          stream=st.cuda_stream, tmax_px=b["depth"], rgba8_init=b["image"])
        else:
            mnv.render_voxels(tv, cams[i], o, rgba8=b["image"], split_track=b["split"], sample_track=b["sample"], stream=st.cuda_stream, tmax_px=b["depth"],
                              rgba8_init=b["image"])

    def timed(server, mode, laps=2):
        def lap():
            for i in range(N_POSES):
                if mode == "wait_each":
                    frame(server, i, 0, streams[0])
                    streams[0].synchronize()
                elif mode == "one_stream":
                    frame(server, i, 0, streams[0])
                else:
                    frame(server, i, i % K, streams[i % K])
        lap()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(laps):
            lap()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / laps / N_POSES * 1e3
        return {"ms_per_frame": round(ms, 4), "Mrays_per_s": round(w * h / ms / 1e3, 1)}

    res = {}
    for server in ("ref_arrays", "ref_arrays_tree_cache", "accel"):
        mnv.set_tree_cache(server == "ref_arrays_tree_cache")
        res[server] = {mode: timed(server, mode) for mode in ("wait_each", "one_stream", "three_in_flight")}
    # one frame of every server against the oracle's tracker frame with the same two attachments: RGBA8 and both tracker arrays
    pose = 5
    counts_host = np.full((cap, 8), 8, np.int16)   # (kept alive: the oracle's tree view holds a pointer into it)
    ot = orc.tree_from_view(v, sample_counts=counts_host)
    want = orc.render(ot, cams[pose].c, o, want_rgba8=True, want_trackers=True, tmax_px=np.full((h, w), 1e9, np.float32),
                      rgba8_init=np.tile(np.array([bg8, bg8, bg8, 255], np.uint8), (h, w, 1)))
    bad = {}
    for server in ("ref_arrays", "ref_arrays_tree_cache", "accel"):
        mnv.set_tree_cache(server == "ref_arrays_tree_cache")
        frame(server, pose, 0, streams[0])
        torch.cuda.synchronize(dev)
        b = sets[0]
        bad[server] = {"rgba8_bytes_differing": int((b["image"].cpu().numpy() != want["rgba8"]).sum()),
                       "tracker_values_differing": int((b["split"].cpu().numpy().reshape(h, w, 3) != want["split"]).sum() + (b["sample"].cpu().numpy().reshape(h, w, 3) != want["sample"]).sum())}
    mnv.set_tree_cache(False)
    res["checked_against_oracle"] = bad
    res["what"] = ("the reference's per-frame call, literally (cuda_renderer.cpp:68-77,97-98,141-142): clear image + depth, fill_(-1) both tracker tensors, render_voxels(..., "
                   "to_split, to_sample, visited, track_visit = false, offscreen = false) on the cfg2 tree at 1920x1080; ref_arrays = mnv_render_voxels_ex on the reference's own "
                   "arrays, ref_arrays_tree_cache = the same after mnv_set_tree_cache(1), accel = mnv_render_voxels_accel_visit_ex; wait_each = one stream with a wait per frame "
                   "(the viewer's glFinish), one_stream = back to back, three_in_flight = three sets of buffers on three streams; wall time incl. the four fills")
    return res


def predicted_step(world, root_period, n_frames, workload):
    """ms per step that the one-GPU emulation of one rank of `world` measured for this partition (the latest profiles/rNN_root_emulation.jsonl, tools/root_emulation.py: rank 0's
    march beside a device copy of the incoming tiles and the un-permute; 64 frames of cfg2 per step), so that the line of a real N-GPU run
    carries the model it is to be held against.  None when no such row exists."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_root_emulation.jsonl")))
    path = found[-1] if found else ""
    if workload != "cfg2" or n_frames != 64 or not os.path.exists(path):
        return None
    rows = [json.loads(ln) for ln in open(path) if ln.strip()]
    rows = [r for r in rows if r.get("world") == world]
    if not rows:
        return None
    best = min(rows, key=lambda r: abs(r.get("root_period", 0) - root_period))
    return {"ms_per_step": best["step_ms"], "rank0_march_only_ms": best["rank0_march_only_ms"], "other_ranks_march_ms": best["rank1_march_only_ms"],
            "root_period_of_the_emulation": best["root_period"], "source": os.path.relpath(path, ROOT), "kernel_source_sha_of_the_emulation": best.get("kernel_source_sha"),
            "stale": best.get("kernel_source_sha") != kernel_source_sha(),  # true: measured with other kernel sources than this run's -- read as an order of magnitude
            "assumes": ["RCCL's receive costs rank 0 no more than a device copy of the same bytes", "a CU-masked stream keeps the reserved units free under N processes",
                        "the peers' sends arrive while rank 0 marches (xGMI point-to-point links are not the bound)"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gather", choices=["f32", "u8"], default="u8",
                    help="pixel format rendered and gathered to rank 0 when N > 1: u8 = the reference's own output format "
                         "(RGBA8, renderer_kernel.cu:237; 8.3 MB per frame), f32 = the float RGBA the parity tests compare (33.2 MB per frame)")
    ap.add_argument("--cpu-poses", type=int, default=16, help="poses rendered by the CPU baseline and compared bit for bit with the GPU frames (default: the whole 16-pose orbit, about 5 s on 128 host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", choices=["accel", "ref_layout"], default="accel")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1; gloo (host-staged gather, all ranks may share one GPU) is the single-GPU rehearsal of the multi-GPU path")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "fog"], default="cfg2",
                    help="cfg2 = BASELINE.json's headline config (default); cfg3 = merged-Mega-NeRF stand-in (anisotropic terrain, 7.2 M chunks); "
                         "cfg4 = cfg3 at 3840x2160 (configs[3], meant for --gpus 8)")
    ap.add_argument("--per-frame", action="store_true", help="one launch per pose instead of one batched launch per step")
    ap.add_argument("--frame-streams", type=int, default=3,
                    help="--per-frame: launches rotate over this many HIP streams (frames in flight; 1 = the frames of a step run back to back on one stream)")
    ap.add_argument("--fast-colour", action="store_true",
                    help="mnv_accel_set_colour_math(accel, 1): hardware exp2 / rcp in the colour sigmoid (alpha and control flow stay exact; colours move ~1e-7)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="N > 1: compute units the march leaves free for the RCCL kernels of the tile gather (the march runs on a CU-masked "
                         "stream).  Default 32 when N > 1 (one unit per shader engine and XCD; measured cost of the march: 10 %%), 0 when N = 1. "
                         "Without it the gather cannot overlap the next step: RCCL's workgroups do not fit beside the persistent march workgroups "
                         "(tools/cumask/probe.py)")
    ap.add_argument("--one-march-stream", action="store_true", help="N > 1: launch every step on the same stream (default: one stream per ring slot)")
    ap.add_argument("--force-dist", action="store_true",
                    help="test hook: with one process, run the N > 1 code path (process group, partition with world 1, reserved stream, "
                         "tile gather, un-permute) -- the only way to execute the RCCL calls on a one-GPU box")
    ap.add_argument("--root-period", type=int, default=-1,
                    help="N > 1: mnv_partition.root_period -- every M-th round of the tile deal leaves rank 0 out, because rank 0 also takes in "
                         "the gather and un-permutes the frames (tools/root_emulation.py: +13 %% on its march at N = 8).  Default round(64 / N) "
                         "(8 at N = 8: rank 0 renders 7/8 of a plain share); 0 = plain round robin")
    ap.add_argument("--gather-via", choices=["abi", "torch"], default="abi",
                    help="N > 1: abi = mnv_gather_tiles (libmnv's own RCCL gather, the product path); torch = torch.distributed's gather "
                         "(the launcher's second attempt if the first one fails or wedges)")
    ap.add_argument("--launch-timeout", type=float, default=600.0, help="--gpus N > 1 without a launcher: watchdog for the ranks this process starts (s)")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the secondary numbers for the other BASELINE configs (cfg3, cfg4_n1, cfg5)")
    ap.add_argument("--test-fail-reserved", action="store_true", help=argparse.SUPPRESS)   # rehearsal of the launcher's ladder; reads no environment
    ap.add_argument("--laps", type=int, default=4, help="the step walks the 16-pose orbit this many times (16 x laps frames in one launch, <= 64)")
    args = ap.parse_args()
    global W, H, N_FRAMES
    if args.workload == "cfg4":
        W, H = 3840, 2160
    if not 1 <= args.laps <= 4:
        raise SystemExit("--laps must be 1 .. 4 (MNV_MAX_BATCH = 64 frames per launch)")
    N_FRAMES = N_POSES * args.laps

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves as FRESH child processes (this process has not
        # touched the GPU: the device count comes from sysfs, gpu_count_without_hip), relay rank 0's JSON line, exit with their code
        sys.exit(self_launch(args.gpus, args.backend, args.launch_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as a {args.gpus}-GPU number")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist   # the partition / gather code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    rccl_ranks = dist.get_world_size() if multi and args.backend == "nccl" else None
    import __graft_entry__ as g
    g.build_if_missing()
    import mega_nerf_viewer_amd as mnv
    import cases

    dev = torch.device("cuda", local_rank)
    t_setup = time.time()
    if args.workload == "cfg2":
        tree = cases.make_tree(mnv, cases.CFG2_TREE)
        cams = [cases.cfg2_camera(mnv, pose % N_POSES, W, H, FX) for pose in range(N_FRAMES)]
        workload = "cfg2: depth-10 SH9 shell N3Tree (1,499,569 chunks), 1920x1080, 16-pose orbit per step" + (f" x {args.laps} laps" if args.laps > 1 else "")
    elif args.workload == "fog":
        tree = cases.make_tree(mnv, cases.FOG_TREE)
        cams = [cases.cfg2_camera(mnv, pose % N_POSES, W, H, FX) for pose in range(N_FRAMES)]
        workload = f"fog: depth-9 SH9 thick shell N3Tree ({tree.capacity:,} chunks; half-thickness 0.06, sigma U(5,40)), {W}x{H}, the cfg2 orbit, 16 poses per step"
    else:
        tree = cases.make_tree(mnv, cases.CFG3_FULL)
        cams = [cases.cfg3_camera(mnv, pose % N_POSES, W, H, fx=1400.0 * W / 1920) for pose in range(N_FRAMES)]
        workload = f"{args.workload}: depth-11 SH9 anisotropic 4x2-brick terrain N3Tree ({tree.capacity:,} chunks), {W}x{H}, 16 oblique poses per step"
    tree.move_to_device()
    if args.fast_colour:
        mnv.accel_set_colour_math(tree.accel, 1)
    if args.workload != "cfg2":
        info = mnv.accel_info(tree.accel)
        workload += f", packed accel {info['device_bytes'] / 1e9:.2f} GB with a level-{info['grid2_level']} lookup grid"
    opt = mnv.RenderOptions.cli_defaults()
    setup_s = time.time() - t_setup

    RING = 2
    reserve = args.reserve_cus if args.reserve_cus >= 0 else (32 if multi else 0)
    if multi and reserve > 0 and args.test_fail_reserved:
        # hidden test flag (tests/test_bench_multirank_gpu.py): stands in for a node on which the CU-masked stream does not work
        print("bench.py: --test-fail-reserved: failing the run with reserved compute units", file=sys.stderr)
        sys.exit(7)
    n_march_streams = RING if multi and not args.one_march_stream else 1
    march_streams = None
    enabled_cus = None   # compute units the march stream may use (None: an ordinary stream, all of them)
    if reserve > 0 or n_march_streams > 1:
        # the march (and what follows it in stream order) runs on streams that cannot use `reserve` compute units; RCCL's own stream
        # and the gatherer's side stream can.  Two such streams, one per ring slot: the launch of step k + 1 fills the wave slots
        # the last wavefronts of step k leave (tools/rank_solo.py: 2.46 -> 2.33 ms per step for one rank of eight).
        march_streams = []
        for _ in range(n_march_streams):
            try:
                handle, enabled = mnv.stream_create_reserved(reserve)
            except mnv.MnvError as e:  # no CU masking on this system: run unmasked (the gather then waits for each march to drain)
                print(f"[bench] rank {rank}: {e}; continuing without reserved compute units", file=sys.stderr)
                reserve = 0
                handle, enabled = mnv.stream_create_reserved(0)
            march_streams.append(torch.cuda.ExternalStream(handle, device=dev))
        enabled_cus = int(enabled)
        mnv.accel_set_cu_budget(tree.accel, enabled)
        torch.cuda.set_stream(march_streams[0])
    stream = torch.cuda.current_stream(dev).cuda_stream
    if not multi:
        # one launch per step: the 16 poses as a batch (frame f at frames[slot][f])
        frames = [torch.empty((N_FRAMES, H, W, 4), dtype=torch.float32, device=dev) for _ in range(RING)]
        dv = tree.device_view() if args.kernel == "ref_layout" else None
        counter = [0]

        # --per-frame: the reference's call pattern (one render_voxels call per frame, cuda_renderer.cpp:141-142), with
        # --frame-streams frames in flight: frame i goes to stream i % K, so the tail of one launch (a few wavefronts finishing
        # the longest rays) overlaps the next launches instead of idling the device
        pf_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.frame_streams))] if args.per_frame and args.frame_streams > 1 else None

        def step():
            slot = counter[0] % RING
            counter[0] += 1
            if args.kernel == "accel" and not args.per_frame:
                mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=frames[slot], stream=stream)
            else:
                for i in range(N_FRAMES):
                    render_pose(i, frames[slot][i], pf_streams[i % len(pf_streams)].cuda_stream if pf_streams else stream)

        def render_pose(i, out, st=None):
            st = stream if st is None else st
            if args.kernel == "accel":
                mnv.render_voxels_accel(tree.accel, cams[i], opt, rgba=out, stream=st)
            else:
                mnv.render_voxels(dv, cams[i], opt, rgba=out, stream=st)
    else:
        from mega_nerf_viewer_amd.multigpu import TileGatherer, TilePartition

        root_period = args.root_period if args.root_period >= 0 else (max(2, round(64 / world)) if world > 1 else 0)
        part = TilePartition(W, H, world, MACRO_W, MACRO_H, root_period)
        n_local = part.local_tiles(rank)
        assert n_local == mnv.partition_local_tiles((0, 0, W, H), rank, world, MACRO_W, MACRO_H, part.root_period)
        dt = torch.float32 if args.gather == "f32" else torch.uint8
        # one launch + one gather per step; the gather of step k overlaps the launch of step k + 1
        comm = None
        # MNV_RCCL_LIBRARY (test hook of mnv_comm.cpp: a transport stand-in that lets several ranks share one GPU) brings the C-ABI gather
        # into the gloo rehearsal as well, so that everything but RCCL's own transport runs as on N GPUs
        if args.gather_via == "abi" and (args.backend == "nccl" or os.environ.get("MNV_RCCL_LIBRARY")):
            # the data path's collective is libmnv's own RCCL gather (mnv_gather_tiles, C ABI); torch.distributed only carries the
            # 128-byte id to the other ranks and the barrier / max-over-ranks of the timing
            box = [mnv.comm_get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = mnv.Comm(box[0], world, rank)
        tg = TileGatherer(part, rank, dev, dtype=dt, depth=RING, frames=N_FRAMES, stage_on_host=args.backend == "gloo" and comm is None, comm=comm, timing=True)
        frames = tg._frames if rank == 0 else None
        counter = [0]

        def step():
            slot = counter[0] % RING
            counter[0] += 1
            st = march_streams[slot % n_march_streams] if march_streams else torch.cuda.current_stream(dev)
            with torch.cuda.stream(st):
                tg.finish(slot)  # orders this stream after the slot's previous gather: its buffer is about to be overwritten
                kw = dict(rgba=tg.local(slot)) if args.gather == "f32" else dict(rgba8=tg.local(slot))
                if n_local > 0:
                    mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=part.part(rank), stream=st.cuda_stream, **kw)
                tg.submit(slot)

        def render_pose(i, out):
            mnv.render_voxels_accel(tree.accel, cams[i], opt, rgba=out, stream=stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def all_max(x):
        t_ = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    if multi:
        # create the RCCL point-to-point channels outside the timed region even with --warmup 0
        tg.submit(0)
        tg.finish(0)
    for _ in range(args.warmup):
        step()
    if multi:
        tg.finish_all()
    sync_all()
    if multi:
        tg.take_timings()   # discard the channel-creating gather and the warm-up steps: per_rank reports the timed steps only
    mnv.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if multi:
        tg.finish_all()
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = mnv.take_timing()
    mnv.set_timing(False)
    per_rank = None
    if multi:
        elapsed = all_max(elapsed)
        # what every rank saw, gathered to rank 0 for the line: an N-GPU run that is slower than the one-GPU emulation predicted says where
        tgt = tg.take_timings()
        mine = {"rank": rank, "march_ms": round(kern_ms / max(1, launches), 4), "march_launches": launches, "local_tiles": int(n_local),
                "gather_ms": round(tgt["gather_ms"], 4) if "gather_ms" in tgt else None, "unpermute_ms": round(tgt["unpermute_ms"], 4) if "unpermute_ms" in tgt else None,
                "reserved_cus": reserve, "enabled_cus": enabled_cus, "cu_mask_in_effect": bool(reserve > 0 and enabled_cus is not None and enabled_cus < torch.cuda.get_device_properties(dev).multi_processor_count),
                "rccl_version": mnv.rccl_version() if comm is not None else None, "device": torch.cuda.get_device_name(dev), "device_index": local_rank}
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box

    rays_per_step = N_FRAMES * W * H
    value = rays_per_step * args.steps / elapsed / 1e6

    # secondary number: the reference's call pattern -- ONE launch per frame -- with --frame-streams frames in flight
    # (what VolumeRenderer::render does); not the headline, reported beside it
    per_frame = None
    if not multi and args.kernel == "accel" and not args.per_frame and args.frame_streams >= 1:
        k = args.frame_streams
        sts = [torch.cuda.Stream(device=dev) for _ in range(k)]
        pf_out = torch.empty_like(frames[0])   # own buffer: the batched frames of the last timed step stay intact for the parity check

        def pf_step():
            for i in range(N_FRAMES):
                mnv.render_voxels_accel(tree.accel, cams[i], opt, rgba=pf_out[i], stream=sts[i % k].cuda_stream)

        pf_step()
        torch.cuda.synchronize(dev)
        pf_steps = max(1, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(pf_steps):
            pf_step()
        torch.cuda.synchronize(dev)
        pf_el = time.perf_counter() - t1
        per_frame = {"value": round(rays_per_step * pf_steps / pf_el / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(pf_el / pf_steps / N_FRAMES * 1e3, 5),
                     "launches": pf_steps * N_FRAMES, "frames_in_flight": k,
                     "what": "one mnv_render_voxels_accel call per 1920x1080 frame (the reference's call pattern, cuda_renderer.cpp:141-142), launches rotating over HIP streams"}

    # second secondary number: the batched launch with mnv_accel_set_colour_math(accel, 1) -- hardware exp2 / rcp in the colour sigmoid only; opacity,
    # transmittance, step sequence and every branch stay on the exact path.  Within the north star's 1e-4 (measured here against the exact
    # frames of the last timed step, which the parity leg below compares bit for bit with the oracle); the headline stays the exact mode.
    fast_colour = None
    if not multi and args.kernel == "accel" and not args.per_frame and not args.fast_colour and args.frame_streams >= 1:
        exact = frames[(counter[0] - 1) % RING]
        fc_out = pf_out
        mnv.accel_set_colour_math(tree.accel, 1)
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=fc_out, stream=stream)
        torch.cuda.synchronize(dev)
        fc_steps = max(1, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(fc_steps):
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=fc_out, stream=stream)
        torch.cuda.synchronize(dev)
        fc_el = time.perf_counter() - t1
        mnv.accel_set_colour_math(tree.accel, 0)
        d = (fc_out - exact).abs()
        fast_colour = {"value": round(rays_per_step * fc_steps / fc_el / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(fc_el / fc_steps * 1e3, 4),
                       "max_abs_drgba_vs_exact": float(d.max().item()), "alpha_not_bit_identical": int((fc_out[..., 3].view(torch.int32) != exact[..., 3].view(torch.int32)).sum().item()),
                       "what": "mnv_accel_set_colour_math(accel, 1): v_exp_f32 / v_rcp_f32 in the colour sigmoid, everything that feeds a branch exact"}
        del d

    # third secondary number: mnv_render_voxels itself -- the literal drop-in on the reference's own arrays (no accel, nothing kept between
    # calls), one call per frame, on one stream and with two frames in flight; bit-identical to the frames above
    ref_layout = None
    if not multi and args.kernel == "accel" and not args.per_frame and args.frame_streams >= 1:
        dv_ref = tree.device_view()
        exact = frames[(counter[0] - 1) % RING]
        rl_out = pf_out
        rl = {}
        for k_rl in (1, 2, -1, -3):   # negative: the same calls with mnv_set_tree_cache(1) on |k| streams
            mnv.set_tree_cache(k_rl < 0)
            sts_rl = [torch.cuda.Stream(device=dev) for _ in range(abs(k_rl))]

            def rl_step():
                for i in range(N_POSES):
                    mnv.render_voxels(dv_ref, cams[i], opt, rgba=rl_out[i], stream=sts_rl[i % abs(k_rl)].cuda_stream)

            rl_step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                rl_step()
            torch.cuda.synchronize(dev)
            rl[k_rl] = (time.perf_counter() - t1) / 3 / N_POSES
            if k_rl == 2:
                rl_bad = int((rl_out[:N_POSES].view(torch.int32) != exact[:N_POSES].view(torch.int32)).any(dim=-1).sum().item())
        rl_bad_cached = int((rl_out[:N_POSES].view(torch.int32) != exact[:N_POSES].view(torch.int32)).any(dim=-1).sum().item())
        mnv.set_tree_cache(False)
        ref_layout = {"value": round(W * H / rl[2] / 1e6, 2), "unit": "Mrays/s", "frames_in_flight": 2, "ms_per_frame": round(rl[2] * 1e3, 5),
                      "one_stream": {"value": round(W * H / rl[1] / 1e6, 2), "ms_per_frame": round(rl[1] * 1e3, 5)},
                      "pixels_not_bit_identical": rl_bad,
                      "what": "one mnv_render_voxels call per 1920x1080 frame on the reference's own arrays (renderer_kernel.hpp:23-34): per-launch level-7 lookup table + walking kernel",
                      "with_tree_cache": {"one_stream": {"value": round(W * H / rl[-1] / 1e6, 2), "ms_per_frame": round(rl[-1] * 1e3, 5)},
                                          "three_in_flight": {"value": round(W * H / rl[-3] / 1e6, 2), "ms_per_frame": round(rl[-3] * 1e3, 5)},
                                          "pixels_not_bit_identical": rl_bad_cached,
                                          "what": "the same calls after mnv_set_tree_cache(1): the stateless entry point keeps the packed re-layout of the tree it saw (include/mnv.h; the caller "
                                                  "invalidates after editing the arrays)"}}

    # ---- roofline of the dominant kernel (the march): algorithmic bytes per launch / launch time
    counters = load_counters(args.workload)
    counters_checked = 0
    cpu_baseline = None
    parity = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        import mnv_oracle as orc
        ot = orc.tree_from_view(tree.host_view())
        n_cpu = max(1, min(args.cpu_poses, N_POSES))
        t_cpu, max_diff, n_bad, n_alpha = 0.0, 0.0, 0, 0
        fresh = {}
        # one thread per physical core this process can run: SMT siblings share the core's ports (no gain for this pointer chase), and a
        # cgroup CPU quota below the core count throttles whatever is started beyond it (tools/cpu_ladder.py: 128 threads under a quota of
        # 16 CPUs run at 8 Mrays/s, 16 threads at 9.4)
        n_thr, phys, hw_threads, quota = orc.baseline_threads()
        # the tree was built by one thread (all its pages on that thread's NUMA node): the baseline marches a copy whose pages its own
        # threads touched first, spread over the nodes
        ot_cpu = orc.copy_first_touch(ot, n_thr)
        for i in range(n_cpu):
            tc = time.perf_counter()
            r = orc.render(ot_cpu, cams[i].c, opt, n_threads=n_thr)
            t_cpu += time.perf_counter() - tc
            fresh[str(i)] = r["counters"].as_dict()
            scratch = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
            render_pose(i, scratch)
            torch.cuda.synchronize(dev)
            gpu = scratch.cpu().numpy()
            if not multi and not args.per_frame and args.kernel == "accel":  # the batched launch wrote the same frame
                assert np.array_equal(frames[(counter[0] - 1) % RING][i].cpu().numpy().view(np.uint32), gpu.view(np.uint32))
            d = np.abs(gpu - r["rgba"])
            max_diff = max(max_diff, float(d.max()))
            n_bad += int((gpu.view(np.uint32) != r["rgba"].view(np.uint32)).any(axis=-1).sum())
            n_alpha += int((gpu[..., 3].view(np.uint32) != r["rgba"][..., 3].view(np.uint32)).sum())
        # a second, untimed-for-parity pass over the same poses: the run-to-run spread of the baseline itself
        t_again = 0.0
        for i in range(n_cpu):
            tc = time.perf_counter()
            orc.render(ot_cpu, cams[i].c, opt, n_threads=n_thr)
            t_again += time.perf_counter() - tc
        orc.free_copy(ot_cpu)
        rates = [n_cpu * W * H / t / 1e6 for t in (t_cpu, t_again)]
        cpu_value = sum(rates) / 2
        cpu_baseline = {"value": round(cpu_value, 4), "unit": "Mrays/s", "cores": n_thr, "host_physical_cores": phys, "hardware_threads": hw_threads,
                        "cgroup_cpu_quota": quota,
                        "per_core": round(cpu_value / n_thr, 4), "passes": [round(x, 4) for x in rates],
                        "spread": round(abs(rates[0] - rates[1]) / cpu_value, 4),
                        "kind": "port", "sample": f"poses 0..{n_cpu - 1} of the 16-pose orbit, full 1920x1080 frames, twice; gcc -O3, OpenMP (OMP_PROC_BIND=spread, OMP_PLACES=cores), "
                                                  "16x16-pixel tiles handed out one at a time, one thread per physical core the process may run (cgroup CPU quota respected), "
                                                  "tree pages first touched by the marching threads"}
        parity = {"max_abs_drgba_vs_oracle": max_diff, "pixels_not_bit_identical": n_bad, "frames_checked": n_cpu,
                  "alpha_not_bit_identical": n_alpha, "colour_math": "fast" if args.fast_colour else "exact"}
        if counters is None:
            counters = {"poses": fresh, "partial": True}
        else:
            # the committed counters (the roofline's numerator) must be the ones this tree and these cameras produce today
            stale = [k for k, v in fresh.items() if any(counters["poses"].get(k, {}).get(n) != x for n, x in v.items())]
            if stale:
                raise SystemExit(f"bench.py: tests/golden/{args.workload if args.workload in ('cfg2', 'fog') else 'cfg3'}_counters.json disagrees with the oracle's counters for poses {stale}: "
                                 f"e.g. committed {counters['poses'].get(stale[0])} vs fresh {fresh[stale[0]]}")
            counters_checked = len(fresh)
    if rank == 0 and multi and not args.no_cpu_baseline:
        # the assembled frames of the last step against the oracle (not timed): validates partition + gather + un-permute
        import mnv_oracle as orc
        ot = orc.tree_from_view(tree.host_view())
        last = frames[(counter[0] - 1) % RING]
        max_diff, n_bad = 0.0, 0
        chk = sorted({0, 1, N_FRAMES // 2, N_FRAMES - 1})   # first, second, a middle and the last frame of the batch (frame stride, ragged rounds)
        n_chk = len(chk)
        for i in chk:
            r = orc.render(ot, cams[i].c, opt, want_rgba8=True)
            gpu = last[i].cpu().numpy()
            if args.gather == "f32":
                max_diff = max(max_diff, float(np.abs(gpu - r["rgba"]).max()))
                n_bad += int((gpu.view(np.uint32) != r["rgba"].view(np.uint32)).any(axis=-1).sum())
            else:  # RGBA8 frames: byte-exact against the oracle's pack, difference reported in float units
                max_diff = max(max_diff, float(np.abs(gpu.astype(np.int32) - r["rgba8"].astype(np.int32)).max()) / 255.0)
                n_bad += int((gpu != r["rgba8"]).any(axis=-1).sum())
        parity = {"max_abs_drgba_vs_oracle": max_diff, "pixels_not_bit_identical": n_bad, "frames_checked": n_chk,
                  "frames": chk, "what": "frames assembled on rank 0 after the gather"}
    footprints = {}
    extras_on = rank == 0 and not multi and args.kernel == "accel" and not args.per_frame and args.workload == "cfg2" and not args.no_extras and not args.no_cpu_baseline
    if extras_on and N_FRAMES == 64:
        # one pass of the diagnostics instantiation per workload in a child process on the test-hook build: the unique 128-byte lines a launch touches
        footprints = measured_footprints(["cfg2", "cfg3", "cfg4", "fog"])
    roofline = None
    if counters is not None and launches > 0:
        poses = counters["poses"]
        mean_bytes = float(np.mean([alg_bytes(c) for c in poses.values()]))
        frames_per_launch = 1 if (args.per_frame or args.kernel != "accel") and not multi else N_FRAMES
        per_launch = mean_bytes * frames_per_launch / world   # each rank's launch covers about 1/world of its frames (rank 0 a little less, see --root-period)
        avg_ms = kern_ms / launches
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = (None, None)
        if not multi and args.kernel == "accel":
            # `traffic` is what the committed rocprofv3 --pmc passes measured for THIS kernel source (sha over csrc/), workload and
            # launch shape; any other source or shape reports null
            traffic, traffic_source = committed_traffic(args.workload, frames_per_launch)
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roofline.update(roofline_fractions(achieved, traffic, avg_ms, footprints.get("cfg2") if frames_per_launch == 64 else None))
        roofline.update({"traffic": traffic, "traffic_source": traffic_source,
                    "counters_rechecked_poses": counters_checked,
                    "kernel": "march_accel_kernel<9,256,0,true>" if args.kernel == "accel" else "march_ref_layout_kernel<9>",
                    "frames_per_launch": frames_per_launch,
                    "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                    "algorithmic_bytes_per_launch": int(per_launch)})
        if multi:
            roofline["approximate"] = True
            roofline["note"] = ("rank 0's launches only; algorithmic bytes = the frame's bytes / world (rank 0 owns a little less, see root_period)"
                                + ("; consecutive launches run on two streams and overlap: each launch's event interval includes the share of the "
                                   "device it left to its neighbour" if n_march_streams > 1 else ""))

    cfg345 = {}
    if rank == 0 and not multi and args.kernel == "accel" and not args.per_frame and args.workload == "cfg2" and not args.no_extras and not args.no_cpu_baseline:
        import mnv_oracle as orc
        try:
            cfg345["live_call"] = extras_live_call(mnv, cases, orc, torch, dev, tree, cams, opt)
        except Exception as e:  # a secondary number must not cost the headline line
            cfg345["live_call"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            cfg345["cfg5"] = extras_cfg5(mnv, cases, torch, dev, tree)
        except Exception as e:  # a secondary number must not cost the headline line
            cfg345["cfg5"] = {"error": f"{type(e).__name__}: {e}"}
        del frames
        tree = None
        torch.cuda.empty_cache()
        try:
            cfg345.update(extras_cfg3_cfg4(mnv, cases, orc, torch, dev, opt, max(1, min(args.steps, 5)), footprints))
        except Exception as e:
            cfg345["cfg3"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            cfg345["fog"] = extras_fog(mnv, cases, orc, torch, dev, opt, max(1, min(args.steps, 5)), footprints)
        except Exception as e:
            cfg345["fog"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        line = {
            "metric": "Mrays/sec at 1920x1080 on depth-10 SH-9 N3Tree; max|dRGBA| vs ref",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload,
                       "rays_per_step": rays_per_step, "kernel": args.kernel, "launches_per_step": 1 if not args.per_frame and args.kernel == "accel" else N_FRAMES,
                       "partition": "none" if not multi else f"interleaved {MACRO_W}x{MACRO_H} macro tiles, {args.gather} " + (
                           ("RCCL (mnv_gather_tiles)" if comm is not None else "RCCL (torch.distributed)") if args.backend == "nccl" else ("mnv_gather_tiles over a transport stand-in (MNV_RCCL_LIBRARY)" if comm is not None
                                                                  else "gloo (host-staged rehearsal)")) + " gather to rank 0",
                       "reserved_cus": reserve, "march_streams": n_march_streams, "root_period": part.root_period if multi else 0},
            "roofline": roofline,
            "per_frame": per_frame,
            "fast_colour": fast_colour,
            "ref_layout": ref_layout,
            "cpu_baseline": cpu_baseline,
            "parity": parity,
            "setup_s": round(setup_s, 2),
        }
        line.update(cfg345)
        if multi:
            line["per_rank"] = per_rank
            line["predicted"] = predicted_step(world, part.root_period, N_FRAMES, args.workload)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL printf()s its library path into C stdio's buffer, which a pipe only sees at exit: flush it now so that the JSON line
        # is the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stderr.flush()
        print(json.dumps(line), flush=True)



if __name__ == "__main__":
    main()
