// mnv_render -- offline batch renderer: the reference's `nerf-viewer` command line without the window.
//
// Flag names and defaults follow the reference (src/opts.cpp:17-32 common flags, main.cpp:491-505
// viewer flags); --grid (wireframe overlay) is accepted and ignored.  --model_path names a model container
// (an .npz, see VolumeRenderer::load_model) and enables the refinement flags, which the reference exposes as
// window check boxes (main.cpp:270-300) rather than flags:
//   --use_splitting          grow / resample / prune the tree from the per-ray trackers while rendering
//   --use_guided_sampling    composite per-sample network outputs instead of the tree's colours
//   --max_depth D  --max_sample_count C  --seed S
//   --save_tree FILE.npz     write the refined tree after the last frame
// Added for batch use:
//   --out PREFIX     write PREFIX_%04d.ppm (RGB from the RGBA8 output) per frame
//   --raw            also write PREFIX_%04d.f32 (float RGBA, row-major, little endian)
//   --frames N       render N frames; with --orbit DEG the camera is rotated about --origin around
//                    --world_up by DEG degrees per frame (default 0: N identical frames, for timing)
//   --gpu ID         HIP device
//   --gpus N         one process per GPU (forked before any HIP call), devices --gpu .. --gpu + N - 1: the tree is replicated, every
//                    frame is cut into interleaved 64x24 macro tiles (mnv_partition), each rank renders its tiles of up to 64
//                    frames with one launch, the compact buffers are gathered to rank 0 over RCCL (mnv_gather_tiles) and
//                    un-permuted there (mnv_assemble_tiles); rank 0 writes the frames.  --gpus 1 runs the same path with one
//                    rank.  --reserve_cus R (default 32 when N > 1) keeps R compute units free for the RCCL kernels.
//                    With --model_path + --use_guided_sampling every rank runs the fused guided-sampling kernel on its tiles
//                    (mnv_render_guided_fused_part: the networks read the tree only).  With --use_splitting the ranks refine the
//                    scene in lock step, one frame at a time (VolumeRenderer::set_ranks: tracker rows and visit marks are
//                    all-gathered, every rank applies the same tree edits to its replica).
//   --in_flight K    plain frames in flight (default 3; VolumeRenderer::frames_in_flight): frame k is downloaded and written
//                    after frames k+1 .. k+K-1 have been issued
//   --guided_in_flight   guided-sampling frames without splitting rotate over the slots as well (VolumeRenderer::guided_in_flight); their
//                    sample counts are printed when the frame is written
#include <hip/hip_runtime_api.h>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <csignal>
#include <deque>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "volume_renderer.hpp"

namespace {

struct Args {
    std::map<std::string, std::string> kv;
    std::string file;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string get(const std::string &k, const std::string &d) const { return has(k) ? kv.at(k) : d; }
    float f(const std::string &k, float d) const { return has(k) ? std::strtof(kv.at(k).c_str(), nullptr) : d; }
    long l(const std::string &k, long d) const { return has(k) ? std::strtol(kv.at(k).c_str(), nullptr, 10) : d; }
    std::vector<float> vec(const std::string &k, std::vector<float> d) const {
        if (!has(k)) return d;
        std::vector<float> out;
        const std::string s = kv.at(k);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            out.push_back(std::strtof(s.substr(i, j - i).c_str(), nullptr));
            i = j + 1;
        }
        return out;
    }
};

Args parse(int argc, char **argv) {
    static const std::map<std::string, std::string> shorts = {
        {"s", "step_size"}, {"e", "stop_thresh"}, {"a", "sigma_thresh"}, {"c", "max_tree_capacity"}, {"x", "split_batch_size"},
        {"n", "nerf_batch_size"}, {"v", "samples_per_voxel"}, {"b", "bounds_only"}, {"y", "appearance_embedding"},
        {"z", "max_guided_samples"}, {"w", "width"}, {"h", "height"}};
    static const char *flags[] = {"bounds_only", "raw", "help", "use_splitting", "use_guided_sampling", "guided_in_flight"};
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string t = argv[i];
        if (t.size() > 1 && t[0] == '-' && !(t[1] >= '0' && t[1] <= '9') && t[1] != '.') {
            std::string name = t.substr(t[1] == '-' ? 2 : 1), val;
            const size_t eq = name.find('=');
            bool has_val = false;
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            if (shorts.count(name)) name = shorts.at(name);
            bool is_flag = false;
            for (const char *f : flags) is_flag |= name == f;
            if (!has_val && !is_flag) {
                if (i + 1 >= argc) throw std::runtime_error("missing value for --" + name);
                val = argv[++i];
            }
            a.kv[name] = is_flag && !has_val ? "1" : val;
        } else {
            a.file = t;
        }
    }
    if (a.has("file")) a.file = a.kv["file"];
    return a;
}

void usage() {
    std::puts("usage: mnv_render npz_file [--bg 0.0] [-s step_size] [-e stop_thresh] [-a sigma_thresh] [-c max_tree_capacity]\n"
              "                  [-w width] [-h height] [--fx 1111] [--fy -1] [--cx -1] [--cy -1] [--center x,y,z] [--back x,y,z]\n"
              "                  [--origin x,y,z] [--world_up x,y,z] [-b] [--out PREFIX] [--raw] [--frames N] [--orbit DEG] [--gpu ID]\n"
              "                  [--in_flight K] [--guided_in_flight] [--gpus N [--reserve_cus R] [--root_period M]]\n"
              "                  [--model_path MODEL.npz [--use_splitting] [--use_guided_sampling] [-x split_batch_size] [-v samples_per_voxel]\n"
              "                   [-y appearance_embedding] [-z max_guided_samples] [--max_depth D] [--max_sample_count C] [--seed S]\n"
              "                   [--save_tree FILE.npz]]");
}

// rotate v about unit axis k by angle (Rodrigues), double precision
void rotate(float v[3], const float k[3], double ang) {
    const double c = std::cos(ang), s = std::sin(ang);
    const double x = v[0], y = v[1], z = v[2], kx = k[0], ky = k[1], kz = k[2];
    const double dot = kx * x + ky * y + kz * z;
    v[0] = (float)(x * c + (ky * z - kz * y) * s + kx * dot * (1 - c));
    v[1] = (float)(y * c + (kz * x - kx * z) * s + ky * dot * (1 - c));
    v[2] = (float)(z * c + (kx * y - ky * x) * s + kz * dot * (1 - c));
}

}  // namespace


namespace {

void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void mnv_ok(int rc, const char *what) {
    if (rc != MNV_OK) throw std::runtime_error(std::string(what) + ": " + mnv_last_error());
}

// the options / camera part of the command line (render_options_from_args src/opts.cpp:49-67, camera flags main.cpp:550-582)
void configure(const Args &args, viewer::VolumeRenderer &rend, int width, int height) {
    rend.options.background_brightness = args.f("bg", 0.0f);
    rend.options.step_size = args.f("step_size", 1e-4f);
    rend.options.stop_thresh = args.f("stop_thresh", 1e-2f);
    rend.options.sigma_thresh = args.f("sigma_thresh", 1e-2f);
    rend.options.split_batch_size = (int)args.l("split_batch_size", 4096);
    rend.options.nerf_batch_size = (int)args.l("nerf_batch_size", 4096);
    rend.options.samples_per_corner = (int)args.l("samples_per_voxel", 8);
    rend.options.appearance_embedding = (int)args.l("appearance_embedding", -1);
    rend.options.max_guided_samples = (int)args.l("max_guided_samples", 128);
    rend.camera = viewer::Camera(width, height, args.f("fx", 1111.f), args.f("fy", -1.f), args.f("cx", -1.f), args.f("cy", -1.f));
    const std::vector<float> center = args.vec("center", {-3.5f, 0.f, 3.5f}), back = args.vec("back", {-0.7071068f, 0.f, 0.7071068f}),
                             origin = args.vec("origin", {0.f, 0.f, 0.f}), up = args.vec("world_up", {0.f, 0.f, 1.f});
    if (center.size() != 3 || back.size() != 3 || origin.size() != 3 || up.size() != 3) throw std::runtime_error("vector flags need 3 components");
    rend.camera.center = {center[0], center[1], center[2]};
    rend.camera.v_back = {back[0], back[1], back[2]};
    rend.camera.origin = {origin[0], origin[1], origin[2]};
    rend.camera.v_world_up = {up[0], up[1], up[2]};
}

// --orbit: rotate the camera about `origin` around world_up by `ang`
void orbit_step(viewer::Camera &cam, double ang) {
    float axis[3] = {cam.v_world_up.x, cam.v_world_up.y, cam.v_world_up.z};
    const float an = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    for (float &v : axis) v /= an;
    float c[3] = {cam.center.x - cam.origin.x, cam.center.y - cam.origin.y, cam.center.z - cam.origin.z};
    float b[3] = {cam.v_back.x, cam.v_back.y, cam.v_back.z};
    rotate(c, axis, ang);
    rotate(b, axis, ang);
    cam.center = {c[0] + cam.origin.x, c[1] + cam.origin.y, c[2] + cam.origin.z};
    cam.v_back = {b[0], b[1], b[2]};
}

void write_frame(const std::string &out, long f, int width, int height, const uint8_t *rgba8, const float *rgba) {
    char name[4096];
    std::snprintf(name, sizeof(name), "%s_%04ld.ppm", out.c_str(), f);
    if (std::FILE *fp = std::fopen(name, "wb")) {
        std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
        static thread_local std::vector<uint8_t> rgb;
        rgb.resize((size_t)width * height * 3);
        for (size_t p = 0; p < (size_t)width * height; ++p) {
            rgb[p * 3 + 0] = rgba8[p * 4 + 0];
            rgb[p * 3 + 1] = rgba8[p * 4 + 1];
            rgb[p * 3 + 2] = rgba8[p * 4 + 2];
        }
        std::fwrite(rgb.data(), 1, rgb.size(), fp);
        std::fclose(fp);
    } else {
        throw std::runtime_error(std::string("cannot write ") + name);
    }
    if (rgba) {
        std::snprintf(name, sizeof(name), "%s_%04ld.f32", out.c_str(), f);
        std::FILE *fp = std::fopen(name, "wb");
        if (!fp) throw std::runtime_error(std::string("cannot write ") + name);
        std::fwrite(rgba, sizeof(float), (size_t)width * height * 4, fp);
        std::fclose(fp);
    }
}

// ---- multi-GPU mode: one process per GPU --------------------------------------------------------------------------------

struct Rendezvous {  // anonymous shared mapping created before the fork
    std::atomic<int> id_ready;
    std::atomic<int> failed;
    char id[MNV_COMM_ID_BYTES];
};

constexpr int kMacroW = 64, kMacroH = 24;  // per-rank launch times within 3 % of each other at world 8

// communicator: rank 0 draws the id, the others read it from the shared page
mnv_comm *join_ranks(int rank, int world, Rendezvous *rv) {
    if (rank == 0) {
        mnv_ok(mnv_comm_get_unique_id(rv->id), "mnv_comm_get_unique_id");
        rv->id_ready.store(1, std::memory_order_release);
    } else {
        while (!rv->id_ready.load(std::memory_order_acquire)) {
            if (rv->failed.load()) throw std::runtime_error("another rank failed before the rendezvous");
            usleep(1000);
        }
    }
    mnv_comm *comm = nullptr;
    mnv_ok(mnv_comm_init_rank(rv->id, world, rank, &comm), "mnv_comm_init_rank");
    return comm;
}

void print_refine_stats(long f, const viewer::VolumeRenderer::FrameStats &st) {
    std::printf("frame %ld: capacity %ld", f, st.capacity);
    if (st.split_candidates || st.added) std::printf("  split candidates %d, added %d%s", st.split_candidates, st.added, st.full ? " (full)" : "");
    if (st.sample_candidates) std::printf("  sample candidates %d, resampled %d", st.sample_candidates, st.resampled);
    if (st.pruned) std::printf("  pruned %d", st.pruned > 0 ? st.pruned : 0);
    if (st.guided_samples > 0) std::printf("  guided samples %ld", st.guided_samples);
    if (st.guided_samples < 0) std::printf("  guided samples: in flight");
    std::printf("\n");
}

// --gpus N with --use_splitting: the ranks refine ONE scene in lock step (VolumeRenderer::set_ranks) -- every rank marches its macro
// tiles, the tracker rows are all-gathered, every rank applies the same splits / resamples / prunes to its replica of the tree, and
// rank 0 assembles and writes the frames.  One frame at a time: a frame reads the tree the previous one left.
// Test hook, compiled only into the -DMNV_TEST_HOOKS build (testhooks/mnv_render): MNV_RANKS_SHARE_GPU=1 puts every rank on device --gpu
// (with a transport stand-in for RCCL, which refuses two ranks on one device: tests/shim/fake_rccl.cpp) so that the world > 1 paths can
// run on a one-GPU machine.  The shipped binary gives rank r device --gpu + r, always.
static bool save_every_rank() {  // test hook as well: every rank writes its replica of the refined tree
#ifdef MNV_TEST_HOOKS
    return std::getenv("MNV_SAVE_EVERY_RANK") != nullptr;
#else
    return false;
#endif
}
static bool ranks_share_gpu() {
#ifdef MNV_TEST_HOOKS
    return std::getenv("MNV_RANKS_SHARE_GPU") != nullptr;
#else
    return false;
#endif
}

int run_rank_refine(const Args &args, int rank, int world, Rendezvous *rv) {
    const bool share = ranks_share_gpu();
    if (hipSetDevice((int)args.l("gpu", 0) + (share ? 0 : rank)) != hipSuccess)
        throw std::runtime_error("rank " + std::to_string(rank) + ": no usable HIP device");
    viewer::N3Tree tree(args.file);
    if (tree.N <= 0) throw std::runtime_error("--gpus needs a tree (N > 0)");
    const int width = (int)args.l("width", 800), height = (int)args.l("height", 800);
    viewer::VolumeRenderer rend;
    configure(args, rend, width, height);
    const long max_capacity = std::max<long>(tree.capacity, args.l("max_tree_capacity", 20000000));
    rend.set(tree, max_capacity);
    rend.resize(width, height);
    rend.load_model(args.get("model_path", ""));
    rend.options.use_splitting = args.has("use_splitting");
    rend.options.use_guided_sampling = args.has("use_guided_sampling");
    rend.options.max_depth = (int)args.l("max_depth", rend.options.max_depth);
    rend.options.max_sample_count = (int)args.l("max_sample_count", rend.options.max_sample_count);
    rend.seed = (uint64_t)args.l("seed", 0);
    mnv_comm *comm = join_ranks(rank, world, rv);
    rend.set_ranks(comm, kMacroW, kMacroH);

    const long frames = args.l("frames", 1);
    const double orbit = args.f("orbit", 0.f) * M_PI / 180.0;
    const std::string out = args.get("out", "");
    std::vector<float> rgba;
    std::vector<uint8_t> rgba8;
    const auto wall0 = std::chrono::steady_clock::now();
    for (long f = 0; f < frames; ++f) {
        rend.render();
        if (rank == 0) {
            print_refine_stats(f, rend.stats);
            if (!out.empty()) {
                rend.download(args.has("raw") ? &rgba : nullptr, &rgba8);
                write_frame(out, f, width, height, rgba8.data(), args.has("raw") ? rgba.data() : nullptr);
            }
        }
        if (orbit != 0.0) orbit_step(rend.camera, orbit);
    }
    rend.sync_tree_streams();
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (args.has("save_tree")) {  // every rank holds the same tree; rank r > 0 writes <name>.rank<r> when asked to (MNV_SAVE_EVERY_RANK: tests)
        const std::string name = args.get("save_tree", "");
        if (rank == 0 || save_every_rank()) {
            rend.sync_tree();
            tree.save_npz(rank == 0 ? name : name + ".rank" + std::to_string(rank) + ".npz");
        }
    }
    if (rank == 0)
        std::printf("%s x %d (RCCL %d): %ld refinement frame(s) %dx%d in lock step, interleaved %dx%d macro tiles, %.3f ms/frame wall%s\n", rend.get_backend(),
                    world, (int)mnv_comm_rccl_version(), frames, width, height, kMacroW, kMacroH, wall_ms / std::max<long>(frames, 1),
                    out.empty() ? "" : " incl. download + file output");
    rend.set_ranks(nullptr);
    mnv_comm_destroy(comm);
    return 0;
}

int run_rank(const Args &args, int rank, int world, Rendezvous *rv) {
    const bool share = ranks_share_gpu();
    if (hipSetDevice((int)args.l("gpu", 0) + (share ? 0 : rank)) != hipSuccess)
        throw std::runtime_error("rank " + std::to_string(rank) + ": no usable HIP device");
    viewer::N3Tree tree(args.file);
    if (tree.N <= 0) throw std::runtime_error("--gpus needs a tree (N > 0)");
    const int width = (int)args.l("width", 800), height = (int)args.l("height", 800);
    viewer::VolumeRenderer rend;
    configure(args, rend, width, height);
    rend.set(tree, tree.capacity);  // every rank holds the whole tree: the march needs no exchange
    rend.resize(width, height);
    if (!tree.device.accel) throw std::runtime_error("--gpus needs the packed accel (N == 2, RGBA or SH1/4/9/16/25 rows)");
    const bool guided = args.has("model_path") && args.has("use_guided_sampling");
    if (guided) {
        rend.load_model(args.get("model_path", ""));  // sets need_viewdir / appearance_embedding as the single-GPU path does
        rend.options.max_guided_samples = (int)args.l("max_guided_samples", 128);
    }

    mnv_comm *comm = join_ranks(rank, world, rv);

    const long frames = args.l("frames", 1);
    const double orbit = args.f("orbit", 0.f) * M_PI / 180.0;
    const std::string out = args.get("out", "");
    const bool raw = args.has("raw");
    std::vector<mnv_camera> cams;  // the whole camera path, as the single-GPU loop would walk it
    for (long f = 0; f < frames; ++f) {
        rend.camera._update();
        cams.push_back(rend.camera.c_abi());
        if (orbit != 0.0) orbit_step(rend.camera, orbit);
    }

    const mnv_rect full = {0, 0, width, height};
    const int32_t root_period = world > 1 ? (int32_t)std::max<long>(2, args.l("root_period", std::lround(64.0 / world))) : 0;
    const mnv_partition part = {rank, world, kMacroW, kMacroH, root_period};
    int32_t j_max = 0;
    for (int r = 0; r < world; ++r) {
        const mnv_partition pr = {r, world, kMacroW, kMacroH, root_period};
        j_max = std::max(j_max, mnv_partition_local_tiles(full, pr));
    }
    const int batch = (int)std::min<long>(MNV_MAX_BATCH, std::max<long>(frames, 1));
    const size_t tile_px = (size_t)j_max * kMacroW * kMacroH, local_px = tile_px * batch, frame_px = (size_t)width * height;

    // the march runs on a stream that leaves `reserve` compute units to the RCCL kernels of the gather
    const int reserve = (int)args.l("reserve_cus", world > 1 ? 32 : 0);
    void *march_stream = nullptr;
    int32_t enabled = 0;
    if (mnv_stream_create_reserved(reserve, &march_stream, &enabled) != MNV_OK) {
        // no CU masking on this system: run unmasked (the gather of a batch then waits for the next batch's march to drain)
        std::fprintf(stderr, "mnv_render[rank %d]: %s; continuing without reserved compute units\n", rank, mnv_last_error());
        mnv_ok(mnv_stream_create_reserved(0, &march_stream, &enabled), "mnv_stream_create_reserved");
    }
    mnv_ok(mnv_accel_set_cu_budget(tree.device.accel, enabled), "mnv_accel_set_cu_budget");
    hipStream_t side = nullptr;
    hip_ok(hipStreamCreateWithFlags(&side, hipStreamNonBlocking), "hipStreamCreate");

    constexpr int RING = 2;
    struct Slot {
        uint8_t *local8 = nullptr, *gathered8 = nullptr, *frames8 = nullptr;
        float *local = nullptr, *gathered = nullptr, *framesf = nullptr;
        hipEvent_t rendered = nullptr, sent = nullptr, assembled = nullptr;
        long first = -1;
        int count = 0;
    } slots[RING];
    for (Slot &s : slots) {
        hip_ok(hipMalloc((void **)&s.local8, local_px * 4), "hipMalloc(local tiles)");
        if (raw) hip_ok(hipMalloc((void **)&s.local, local_px * 16), "hipMalloc(local tiles f32)");
        if (rank == 0) {
            hip_ok(hipMalloc((void **)&s.gathered8, local_px * 4 * world), "hipMalloc(gather table)");
            hip_ok(hipMalloc((void **)&s.frames8, frame_px * 4 * batch), "hipMalloc(frames)");
            if (raw) {
                hip_ok(hipMalloc((void **)&s.gathered, local_px * 16 * world), "hipMalloc(gather table f32)");
                hip_ok(hipMalloc((void **)&s.framesf, frame_px * 16 * batch), "hipMalloc(frames f32)");
            }
        }
        hip_ok(hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&s.sent, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&s.assembled, hipEventDisableTiming), "hipEventCreate");
    }
    std::vector<uint8_t> host8;
    std::vector<float> hostf;
    auto flush = [&](Slot &s) {  // rank 0: wait for the slot's assembled frames and write them
        if (s.first < 0) return;
        hip_ok(hipEventSynchronize(s.assembled), "assemble");
        if (rank == 0 && !out.empty()) {
            host8.resize(frame_px * 4 * s.count);
            hip_ok(hipMemcpy(host8.data(), s.frames8, host8.size(), hipMemcpyDeviceToHost), "download frames");
            if (raw) {
                hostf.resize(frame_px * 4 * s.count);
                hip_ok(hipMemcpy(hostf.data(), s.framesf, hostf.size() * 4, hipMemcpyDeviceToHost), "download frames f32");
            }
            for (int i = 0; i < s.count; ++i)
                write_frame(out, s.first + i, width, height, host8.data() + frame_px * 4 * i, raw ? hostf.data() + frame_px * 4 * i : nullptr);
        }
        s.first = -1;
    };

    const auto wall0 = std::chrono::steady_clock::now();
    int b = 0;
    for (long f0 = 0; f0 < frames; f0 += batch, ++b) {
        Slot &s = slots[b % RING];
        flush(s);  // frames of the batch that used this slot two batches ago
        const int n = (int)std::min<long>(batch, frames - f0);
        s.first = f0;
        s.count = n;
        // one launch per rank and batch; the buffer is reused only after the gather that last read it
        hip_ok(hipStreamWaitEvent((hipStream_t)march_stream, s.sent, 0), "wait(sent)");
        if (!guided) {
            mnv_ok(mnv_render_voxels_accel_batch(tree.device.accel, cams.data() + f0, n, rend.options.c_abi(), full, part, s.local, s.local8, march_stream),
                   "mnv_render_voxels_accel_batch");
        } else {
            // guided sampling: one fused launch (march + networks + composite) per frame, into the frame's place in the batch buffer
            for (int i = 0; i < n; ++i)
                mnv_ok(mnv_render_guided_fused_part(tree.device.accel, &cams[f0 + i], rend.options.c_abi(), full, part, rend.model(), &rend.cluster_grid(),
                                                    s.local ? s.local + (size_t)i * tile_px * 4 : nullptr, s.local8 + (size_t)i * tile_px * 4, nullptr,
                                                    march_stream),
                       "mnv_render_guided_fused_part");
        }
        hip_ok(hipEventRecord(s.rendered, (hipStream_t)march_stream), "record(rendered)");
        // gather + un-permute on the side stream, overlapping the next batch's march
        hip_ok(hipStreamWaitEvent(side, s.rendered, 0), "wait(rendered)");
        mnv_ok(mnv_gather_tiles(comm, s.local8, s.gathered8, tile_px * 4 * n, 0, side), "mnv_gather_tiles");
        if (raw) mnv_ok(mnv_gather_tiles(comm, s.local, s.gathered, tile_px * 16 * n, 0, side), "mnv_gather_tiles(f32)");
        hip_ok(hipEventRecord(s.sent, side), "record(sent)");
        if (rank == 0) {
            mnv_ok(mnv_assemble_tiles(s.gathered8, s.frames8, width, height, part, n, 4, side), "mnv_assemble_tiles");
            if (raw) mnv_ok(mnv_assemble_tiles(s.gathered, s.framesf, width, height, part, n, 16, side), "mnv_assemble_tiles(f32)");
        }
        hip_ok(hipEventRecord(s.assembled, side), "record(assembled)");
    }
    for (int k = 0; k < RING; ++k) flush(slots[(b + k) % RING]);
    hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (rank == 0)
        std::printf("%s x %d (RCCL %d): %ld frame(s) %dx%d, interleaved %dx%d macro tiles, %.3f ms/frame wall%s, %.1f Mrays/s\n", rend.get_backend(), world,
                    (int)mnv_comm_rccl_version(), frames, width, height, kMacroW, kMacroH, wall_ms / std::max<long>(frames, 1),
                    out.empty() ? "" : " incl. download + file output", wall_ms > 0 ? (double)width * height * frames / wall_ms / 1e3 : 0.0);
    for (Slot &s : slots) {
        for (void *p : {(void *)s.local8, (void *)s.gathered8, (void *)s.frames8, (void *)s.local, (void *)s.gathered, (void *)s.framesf})
            if (p) (void)hipFree(p);
        (void)hipEventDestroy(s.rendered);
        (void)hipEventDestroy(s.sent);
        (void)hipEventDestroy(s.assembled);
    }
    (void)hipStreamDestroy(side);
    mnv_comm_destroy(comm);
    (void)mnv_stream_destroy(march_stream);
    return 0;
}

// Fork one child per rank BEFORE anything in this process touches HIP, wait for them, and take a failing rank's peers down
// with it (they would otherwise wait in RCCL forever).
int run_distributed(const Args &args, int world) {
    if (world < 1 || world > 64) throw std::runtime_error("--gpus must be 1 .. 64");
    if (args.has("use_splitting") && !args.has("model_path")) throw std::runtime_error("--use_splitting needs --model_path");
    if (args.has("use_guided_sampling") && !args.has("model_path")) throw std::runtime_error("--use_guided_sampling needs --model_path");
    void *page = mmap(nullptr, sizeof(Rendezvous), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (page == MAP_FAILED) throw std::runtime_error("mmap failed");
    Rendezvous *rv = new (page) Rendezvous();
    rv->id_ready.store(0);
    rv->failed.store(0);
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        std::fflush(nullptr);
        const pid_t pid = fork();
        if (pid < 0) throw std::runtime_error("fork failed");
        if (pid == 0) {
            int rc = 1;
            try {
                rc = args.has("use_splitting") ? run_rank_refine(args, r, world, rv) : run_rank(args, r, world, rv);
            } catch (const std::exception &e) {
                std::fprintf(stderr, "mnv_render[rank %d]: %s\n", r, e.what());
                rv->failed.store(1);
            }
            std::fflush(nullptr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    int worst = 0, left = world;
    while (left > 0) {
        int status = 0;
        const pid_t pid = wait(&status);
        if (pid < 0) break;
        --left;
        const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (rc != 0 && worst == 0) {
            worst = rc;
            for (pid_t k : kids)
                if (k != pid) (void)kill(k, SIGTERM);  // exactly the processes started above
        }
    }
    munmap(page, sizeof(Rendezvous));
    return worst;
}

}  // namespace

int main(int argc, char **argv) {
    try {
        const Args args = parse(argc, argv);
        if (args.has("help") || args.file.empty()) {
            usage();
            return args.has("help") ? 0 : 2;
        }
        if (args.has("gpus")) return run_distributed(args, (int)args.l("gpus", 1));  // before any HIP call in this process
        if (hipSetDevice((int)args.l("gpu", 0)) != hipSuccess) throw std::runtime_error("no usable HIP device");

        viewer::N3Tree tree(args.file);  // main.cpp:528
        if (args.has("bounds_only") && tree.N > 0) {  // main.cpp:529-538: keep one empty root chunk
            tree.capacity = 1;
            tree.data.assign((size_t)8 * tree.data_dim, 0);
            tree.child.assign(8, 0);
            tree.parent.assign(1, -1);
            tree.sample_counts.assign(8, 8);
        }
        const int width = (int)args.l("width", 800), height = (int)args.l("height", 800);  // main.cpp:491-492
        viewer::VolumeRenderer rend;
        configure(args, rend, width, height);
        const bool refine = args.has("model_path") && (args.has("use_splitting") || args.has("use_guided_sampling"));
        // the reference reserves max_tree_capacity (default 20M chunks) up front; without refinement the tree cannot grow
        const long max_capacity = refine ? std::max<long>(tree.capacity, args.l("max_tree_capacity", 20000000)) : tree.capacity;
        if (tree.N > 0) rend.set(tree, max_capacity);
        rend.resize(width, height);
        if (args.has("model_path")) {  // main.cpp:585-589
            rend.load_model(args.get("model_path", ""));
            rend.options.use_splitting = args.has("use_splitting");
            rend.options.use_guided_sampling = args.has("use_guided_sampling");
            rend.options.max_depth = (int)args.l("max_depth", rend.options.max_depth);
            rend.options.max_sample_count = (int)args.l("max_sample_count", rend.options.max_sample_count);
            rend.seed = (uint64_t)args.l("seed", 0);
        }

        const long frames = args.l("frames", 1);
        const double orbit = args.f("orbit", 0.f) * M_PI / 180.0;
        const std::string out = args.get("out", "");
        std::vector<float> rgba;
        std::vector<uint8_t> rgba8;
        rend.frames_in_flight = (int)std::max<long>(1, args.l("in_flight", rend.frames_in_flight));
        rend.guided_in_flight = args.has("guided_in_flight");
        std::deque<std::pair<long, int>> pending;  // (frame, slot) rendered but not yet written
        auto write_oldest = [&]() {
            const long f = pending.front().first;
            const int slot = pending.front().second;
            pending.pop_front();
            rend.download_slot(slot, args.has("raw") ? &rgba : nullptr, &rgba8);
            if (refine && rend.guided_in_flight) std::printf("frame %ld: guided samples %ld\n", f, rend.slot_guided_samples(slot));
            write_frame(out, f, width, height, rgba8.data(), args.has("raw") ? rgba.data() : nullptr);
        };
        const auto wall0 = std::chrono::steady_clock::now();
        for (long f = 0; f < frames; ++f) {
            // the slot frame f is about to take must have been written: the renderer says which one that is (frames that cannot
            // overlap -- refinement, a tree without a packed accel -- all take slot 0, whatever --in_flight says)
            const int next = rend.next_slot();
            size_t keep = pending.size();
            for (size_t i = 0; i < pending.size(); ++i)
                if (pending[i].second == next) keep = pending.size() - 1 - i;
            while (pending.size() > keep) write_oldest();
            rend.render();
            if (refine) print_refine_stats(f, rend.stats);
            if (!out.empty()) pending.emplace_back(f, rend.last_slot());
            if (orbit != 0.0) orbit_step(rend.camera, orbit);
        }
        while (!pending.empty()) write_oldest();
        rend.sync_tree_streams();
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        if (args.has("save_tree") && tree.N > 0) {
            rend.sync_tree();
            tree.save_npz(args.get("save_tree", ""));
        }
        std::printf("%s: %ld frame(s) %dx%d, %.3f ms/frame wall (%d in flight%s), %.1f Mrays/s\n", rend.get_backend(), frames,
                    width, height, wall_ms / frames, refine && !rend.overlaps_next() ? 1 : rend.frames_in_flight, out.empty() ? "" : ", incl. download + file output",
                    wall_ms > 0 ? (double)width * height * frames / wall_ms / 1e3 : 0.0);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "mnv_render: %s\n", e.what());
        return 1;
    }
}
