// volume_renderer.hpp -- viewer::VolumeRenderer for offline batch rendering.
//
// Same role and member names as the reference facade (reference include/renderer/renderer.hpp:9-39,
// src/renderer/cuda_renderer.cpp:68-163,383-516): it owns a Camera and RenderOptions, `set()` moves a
// tree to the device, `resize()` sizes the frame, `render()` launches the march for the current
// camera.  The GL framebuffers / CUDA-GL interop / blit of the reference are replaced by a linear
// device frame (float RGBA + RGBA8) that `download()` copies to the host.
//
// With a model loaded (load_model / set_model) render() also runs the reference's refinement loop
// (cuda_renderer.cpp:98-156,205-381): options.use_guided_sampling composites per-sample network outputs
// instead of tree colours, options.use_splitting grows the tree from the per-ray trackers
// (expand_voxels / get_more_samples) and prunes unvisited chunks when it is nearly full (prune_tree).
// Every step is a device-resident libmnv entry point; the host only sequences them.  The networks are the
// build's own small MLPs (include/mnv.h, mnv_mlp_desc) -- the reference's TorchScript containers are not
// part of its repository.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "camera.hpp"
#include "n3tree.hpp"
#include "render_options.hpp"

namespace viewer {

struct VolumeRenderer {
    explicit VolumeRenderer();
    ~VolumeRenderer();

    // Render the currently set tree with `camera` and `options` (asynchronous on the internal stream).
    void render();
    // Set volumetric data to render: tree.move_to_device(max_tree_capacity, true, true) and
    // basis_minmax = {0, max(basis_dim - 1, 0)} as cuda_renderer.cpp:498-516.
    void set(N3Tree &tree, long max_tree_capacity);
    // Clear the volumetric data
    void clear();
    // Resize the frame.  The first call only sets the size; later calls rescale the intrinsics
    // (fx, fy, cx, cy each scaled once -- the reference scales cy twice, cuda_renderer.cpp:398,404-406).
    void resize(int width, int height);
    // Name identifying the renderer backend
    const char *get_backend();

    // Wait for the last render() and copy the frame to the host ([height][width][4]).
    void download(std::vector<float> *rgba, std::vector<uint8_t> *rgba8);
    // Frames in flight: plain frames (no refinement, accel current) rotate over this many slots, each with its own HIP stream
    // and frame buffers, so that the tail of one launch overlaps the next launches (1 = the reference's one-stream behaviour).
    // render() returns at once; last_slot() names the slot it used and download_slot() waits for that slot's frame only --
    // a caller that wants overlap downloads frame k after it has issued frames k+1 .. k+frames_in_flight-1.
    int frames_in_flight = 3;
    // Guided-sampling frames that change nothing (use_guided_sampling without use_splitting, the fused kernel, a tree below 3/4 of its
    // capacity) may rotate over the slots as well (default off: such a frame's sample count is then not known when render() returns --
    // stats.guided_samples is -1 and slot_guided_samples() waits for it).  1.20 -> 1.09 ms per 1080p frame on cfg2 with three in flight.
    bool guided_in_flight = false;
    long slot_guided_samples(int slot);
    int last_slot() const;
    int next_slot() const;       // the slot the next render() will take (with the options, tree and camera as they are now)
    bool overlaps_next() const;  // ... and whether that frame runs beside the previous ones (a plain frame on the packed accel)
    void download_slot(int slot, std::vector<float> *rgba, std::vector<uint8_t> *rgba8);
    // Wait for every frame in flight.
    void sync_tree_streams();
    // Device pointers of the current frame (valid until the next resize()).
    const float *device_rgba() const;
    const uint8_t *device_rgba8() const;
    // Average device time per render() since the last call, in ms (HIP events).

    // Role of load_model (cuda_renderer.cpp:518-539): read a model container.  Here an .npz with
    //   mlp_desc int32[9] (n_clusters, pos_octaves, dir_octaves, need_viewdir, n_embeddings, embedding_dim,
    //   hidden_width, hidden_layers, out_dim), mlp_center f32[3], mlp_inv_extent f32[3], mlp_params f16/u16 [..],
    //   grid_dim int32[2], min_position f32[3], max_position f32[3]
    // Sets options.need_viewdir / appearance_embedding from the description as the reference does.
    void load_model(const std::string &npz_path);
    void set_model(const mnv_mlp_desc &desc, const uint16_t *params, size_t n_halfs, const mnv_cluster_grid &grid);
    bool has_model() const;
    // The loaded model as the C ABI sees it (for callers that drive libmnv entry points themselves, e.g. the multi-GPU mode of mnv_render).
    const mnv_mlp *model() const;
    const mnv_cluster_grid &cluster_grid() const;
    // Copy the (refined) device tree back into the N3Tree's host arrays, e.g. before N3Tree::save_npz.
    void sync_tree();

    // Several ranks, one process per GPU, render and refine ONE scene in lock step (BASELINE.json configs[4] on several GPUs; SURVEY.md
    // 8(e): "identical deterministic refinement on every rank").  Every rank holds the whole tree and the networks.  render() then
    //   * marches only the macro tiles of this rank (pixels, tracker rows and visit marks: mnv_render_voxels_accel_visit_part /
    //     mnv_render_guided_fused_track_part),
    //   * all-gathers the tracker rows (mnv_allgather) -- the votes of cuda_renderer.cpp:205-227 count candidates over the whole frame and
    //     do not depend on the order of the rows -- and, on a visit-mark frame, the marks (mnv_merge_visit_marks),
    //   * runs the same split / resample / prune on every rank (same candidates, same jitter seed, same network): the replicas stay
    //     identical chunk for chunk, and equal to what one GPU makes of the same camera path,
    //   * gathers the tiles to rank 0 (mnv_gather_tiles) and un-permutes them there: rank 0's frame is the whole picture.
    // Needs the packed accel; guided sampling needs a network the fused kernel covers.  `comm` stays the caller's.
    void set_ranks(mnv_comm *comm, int tile_w = 64, int tile_h = 24);
    // The reference's render loop calls its three launchers with offscreen == false (cuda_renderer.cpp:111-113,135-136,141-142): rays stop at the
    // depth attachment of the GL pass, the frame is composited over its image.  Here the two attachments are device arrays of the caller
    // ([height][width] float, [height][width][4] uint8; either may be null) that stay valid while frames are rendered; both null (the default) is
    // the offline renderer.  One rank only.
    void set_frame_inputs(const float *tmax_px_device, const uint8_t *rgba8_init_device);

    // What the last render() did (the reference prints these to stdout).
    struct FrameStats {
        bool track_visit = false, used_accel = false, full = false;
        bool fused = false;                         // the guided-sampling frame ran as one kernel (mnv_render_guided_fused)
        int split_candidates = 0, added = 0;        // expand_voxels
        int sample_candidates = 0, resampled = 0;   // get_more_samples
        int pruned = 0;                             // prune_tree (-1: ran, nothing to prune)
        long guided_samples = 0;                    // rows sent to the networks by guided sampling
        long capacity = 0;
    } stats;
    // Seed of the sample-position jitter (torch::rand in the reference); frame f uses (seed, f).
    uint64_t seed = 0;
    // Guided-sampling frames run as one fused kernel when nothing but the picture is wanted from them (false: always the
    // four-step path of the reference, cuda_renderer.cpp:107-139).
    bool use_fused_guided = true;
    // Frames without a tree change after which the packed accel is rebuilt and used again.
    int accel_rebuild_after = 4;

    // Camera instance
    Camera camera;
    // Rendering options
    RenderOptions options;

private:
    void render_ranks();
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace viewer
