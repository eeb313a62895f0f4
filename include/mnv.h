/*
 * mnv.h -- C ABI of the MI355X-native N3Tree (PlenOctree) ray-march path.
 *
 * This is the drop-in boundary for ONE path of cmusatyalab/mega-nerf-viewer:
 * the per-ray octree traversal + spherical-harmonic evaluation + front-to-back
 * alpha compositing that the reference launches through
 *
 *     viewer::render_voxels(N3Tree&, const Camera&, const RenderOptions&, ...)
 *         reference: include/cuda/renderer_kernel.hpp:23-34
 *                    src/cuda/renderer_kernel.cu:396-437 (launcher), :243-292 (kernel)
 *
 * Conventions kept from the reference: non-owning views, caller-owned output
 * buffers, POD structs by value/pointer, asynchronous on a caller stream.
 * Conventions changed on purpose: every entry point returns
 * an int status (0 = ok, otherwise a hipError_t or MNV_E_* code) instead of
 * cuda_assert's exit(); the c2w matrix travels by value inside mnv_camera
 * (no hidden default-stream memcpy, camera.cpp:113-123); an explicit tile
 * rectangle selects the pixels to render (multi-GPU partition); float RGBA is
 * the primary output, RGBA8 (renderer_kernel.cu:237) an optional second one.
 *
 * All pointers named "device" must be HIP device pointers on the current
 * device.  No torch types cross this boundary.
 *
 * HOW THIS HEADER IS ORGANISED.  The DROP-IN CORE is the 30 entry points below, one per function / method of the reference's interface for
 * this path -- a maintainer who replaces src/cuda + the loaders binds exactly these (include/mnv_reference_binding.hpp does, inside a build of
 * the reference).  Everything else in this file is an EXTENSION the reference has no counterpart for, grouped by purpose further down: the
 * packed layout ("accel": mnv_accel_*, mnv_render_voxels_accel*, mnv_render_guided_fused*), multi-GPU (mnv_partition*, mnv_comm_*,
 * mnv_gather_tiles, mnv_assemble_tiles, ...), the device-side forms of the host logic between frames (mnv_select_*, mnv_apply_*,
 * mnv_prune_tree*, mnv_query_submodules, mnv_mlp_*), synthetic trees (mnv_synth_*).  No call's result depends on process-wide state, with
 * one opt-in exception that is documented where it is declared (mnv_set_tree_cache).
 *
 *   reference                                                   drop-in core (tests/test_capi_symbols.py: CORE)
 *   include/cuda/renderer_kernel.hpp:23-34  render_voxels        mnv_render_voxels, mnv_render_voxels_ex (offscreen == false)
 *   :36-52  get_samples_from_voxels                              mnv_get_samples_from_voxels, mnv_get_samples_from_voxels_ex
 *   :12-21  render_nerf_results                                  mnv_render_nerf_results
 *   :54-63  add_children_and_generate_samples                    mnv_add_children_and_generate_samples
 *   :65-73  generate_samples                                     mnv_generate_samples
 *   :75-79  adjust_parents_and_children                          mnv_adjust_parents_and_children
 *   include/renderer/renderer.hpp:9-39  VolumeRenderer           mnv_renderer_create, _destroy, _set, _load_model, _resize, _options,
 *                                                                _set_camera, _render, _download
 *   include/n3tree/n3tree.hpp:17-69  N3Tree                      mnv_n3tree_open, _free, _move_to_device, _host_view, _device_view
 *   include/data_format.hpp:7-22  DataFormat                     mnv_data_format_parse, mnv_data_format_to_string
 *   include/camera.hpp:12-87  Camera                             mnv_camera_init, mnv_camera_set_pose, mnv_camera_drag
 *   include/render_options.hpp:9-56, src/opts.cpp:17-32          mnv_default_render_options, mnv_cli_render_options
 *   src/cuda/common.cu:8-20  cuda_assert (exit)                  status codes + mnv_last_error
 */
#ifndef MNV_H
#define MNV_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNV_VERSION 1
#define MNV_BASIS_MAX 25 /* reference include/render_options.hpp:4 */

/* status codes (positive values are hipError_t) */
#define MNV_OK 0
#define MNV_E_INVALID (-1)     /* bad argument */
#define MNV_E_UNSUPPORTED (-2) /* e.g. N != 2, basis_dim not in {-1,0,1,4,9,16,25} */
#define MNV_E_NO_DEVICE (-3)   /* no HIP device / extension unusable */
#define MNV_E_IO (-4)          /* file / npz errors */
#define MNV_E_NO_RCCL (-5)     /* mnv_comm_*: librccl.so.1 cannot be loaded */
#define MNV_E_RCCL (-6)        /* mnv_comm_*: an RCCL call failed (text in mnv_last_error) */
#define MNV_E_FAULT (-7)       /* an EARLIER asynchronous frame on this object is known to be wrong (mnv_render_guided_fused*: watchdog) */

#define MNV_FORMAT_RGBA 0 /* reference include/data_format.hpp:8-12 */
#define MNV_FORMAT_SH 1

/*
 * Non-owning view of an N3Tree in the reference's own array layout
 * (replaces viewer::internal::TreeSpec, include/data_spec.hpp:25-50).
 */
typedef struct mnv_tree_view {
    const uint16_t *data;         /* [capacity][N^3][data_dim] binary16, n3tree.cpp:183-188 */
    const int32_t *child;         /* [capacity][N^3] relative chunk offsets, 0 = leaf, n3tree.cpp:92-97 */
    const int32_t *parent;        /* [capacity], may be NULL, n3tree.cpp:99-107 */
    const int16_t *sample_counts; /* [capacity][N^3], may be NULL, n3tree.cpp:191-193 */
    float offset[3];
    float scale[3];
    int32_t N;         /* spatial branching factor per axis.  N = 2 (every PlenOctree file) everywhere; mnv_render_voxels also takes
                          3 <= N <= 16 (arrays [capacity][N^3]...) through a general walk, as the reference's descent multiplies by tree.N
                          (rt_core.cuh:137-143; its loader warns about N != 2, n3tree.cpp:85-87).  The packed accel, the sample march and the
                          refinement entry points answer MNV_E_UNSUPPORTED for N != 2 (generate_samples is N == 2 in the reference too) */
    int32_t data_dim;  /* halfs per voxel row; sigma is column data_dim-1 */
    int32_t format;    /* MNV_FORMAT_* */
    int32_t basis_dim; /* SH basis functions per channel, -1 if none */
    int32_t capacity;  /* chunks in use */
} mnv_tree_view;

/* Replaces viewer::internal::CameraSpec (include/data_spec.hpp:9-23). */
typedef struct mnv_camera {
    int32_t width, height;
    float fx, fy, cx, cy;
    float c2w[12]; /* column-major [right | up | back | center], camera.cpp:55-82 */
} mnv_camera;

/* viewer::RenderOptions, field for field (include/render_options.hpp:9-56). */
typedef struct mnv_render_options {
    float step_size;
    float sigma_thresh;
    float stop_thresh;
    float background_brightness;
    float render_bbox[6];
    int32_t basis_minmax[2];
    float rot_dirs[3];
    bool show_grid;
    int32_t grid_max_depth;
    bool render_depth;
    bool use_splitting;
    bool use_guided_sampling;
    int32_t max_depth;
    int32_t samples_per_corner;
    int32_t split_batch_size;
    int32_t nerf_batch_size;
    int32_t max_sample_count;
    bool need_viewdir;
    int32_t appearance_embedding;
    int32_t max_guided_samples;
} mnv_render_options;

/* Tile of the camera image; the full frame is {0, 0, width, height}. */
typedef struct mnv_rect {
    int32_t x0, y0, w, h;
} mnv_rect;

/* ------------------------------------------------------------------ misc */

int mnv_version(void);
/* Thread-local description of the last non-zero status returned on this thread. */
const char *mnv_last_error(void);
/* first 16 hex digits of the SHA-256 over the library's sources (csrc/, host/, include/ in the Makefile's order) at build time */
const char *mnv_source_sha(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int mnv_device_count(void);
/* struct defaults of include/render_options.hpp:9-56 */
void mnv_default_render_options(mnv_render_options *opt);
/* CLI defaults of src/opts.cpp:17-32,49-67 (--bg 0, split/nerf batch 4096) */
void mnv_cli_render_options(mnv_render_options *opt);
/* Camera::Camera + Camera::_update pose math (src/camera.cpp:29-82):
 * fy<0 -> fx, cx<0 -> width/2, cy<0 -> height/2 (integer division as in the reference). */
void mnv_camera_init(mnv_camera *cam, int32_t width, int32_t height, float fx, float fy, float cx, float cy);
void mnv_camera_set_pose(mnv_camera *cam, const float center[3], const float v_back[3],
                         const float v_world_up[3]);
/* Camera's drag helpers (include/camera.hpp:22-25, src/camera.cpp:132-187) as one stateless call for hosts without the C++ struct: the
 * pose (center, v_back, origin; updated in place) after begin_drag(x0, y0, is_pan, about_origin); drag_update(x1, y1); end_drag(); and the
 * _update() of the next frame; cam->c2w receives the new matrix.  The C++ host model (host/camera.hpp) has the four members themselves. */
void mnv_camera_drag(mnv_camera *cam, float center[3], float v_back[3], const float v_world_up[3], float origin[3], float movement_speed,
                     int is_pan, int about_origin, float x0, float y0, float x1, float y1);

/* ------------------------------------------------- the hot path (device) */

/*
 * Direct replacement of viewer::render_voxels (offscreen branch,
 * renderer_kernel.cu:225-229,260,277-280) on the reference array layout.
 *   tree        device view (all array pointers are device pointers)
 *   tile        pixels [x0,x0+w) x [y0,y0+h) of cam's image
 *   rgba_out    device float [tile.h][tile.w][4]: rgb after background
 *               composite, a = accumulated opacity (the four floats the
 *               reference holds just before its u8 cast); may be NULL
 *   rgba8_out   device uint8 [tile.h][tile.w][4], renderer_kernel.cu:237; may be NULL
 *   split_track device float [tile.h][tile.w][3] (priority, chunk, child) or NULL
 *   sample_track same for the sample tracker or NULL (rt_core.cuh:237-252,308-321);
 *               the caller pre-fills both with -1 as cuda_renderer.cpp:97-98 does
 *   visited     device int32 [capacity] or NULL; marked when track_visit != 0
 *   hip_stream  hipStream_t; the call is asynchronous on it
 */
int mnv_render_voxels(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt,
                      mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                      float *sample_track, int32_t *visited, int track_visit, void *hip_stream);

/*
 * The reference's LIVE call shape.  VolumeRenderer::Impl::render passes offscreen == false to all three launchers
 * (src/renderer/cuda_renderer.cpp:111-113,135-136,141-142): the kernels then read, per pixel, a ray limit from the depth attachment
 * (float t_max = surf2Dread(surf_obj_depth), renderer_kernel.cu:277-280,354-357) and composite over the pixel that is already in
 * the image instead of over background_brightness (renderer_kernel.cu:230-234,260-264) -- how the volume goes behind / in front of
 * the wireframe mesh the GL pass drew.  The two cudaArray_t attachments become two linear device arrays, indexed like the outputs:
 */
typedef struct mnv_frame_inputs {
    const float *tmax_px;       /* device float [tile.h][tile.w]: t_max of every pixel; NULL = 1e9f everywhere (offscreen == true) */
    const uint8_t *rgba8_init;  /* device uint8 [tile.h][tile.w][4]: the image under the volume; NULL = background_brightness
                                   (offscreen == true).  May be the same buffer as rgba8_out (the reference reads and writes one surface). */
} mnv_frame_inputs;
/* mnv_render_voxels with the inputs of offscreen == false; inputs == NULL (or both members NULL) is mnv_render_voxels exactly. */
int mnv_render_voxels_ex(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                         const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                         int32_t *visited, int track_visit, void *hip_stream);

/*
 * Packed device re-layout of a tree ("accel"): one 32-bit word per voxel
 * (child link or leaf sigma), colour rows padded to 64 B, plus a dense
 * top-of-tree lookup grid.  Built once per tree upload -- the counterpart of
 * N3Tree::move_to_device (n3tree.cpp:207-246); results are bit-identical to
 * mnv_render_voxels.
 */
typedef struct mnv_accel mnv_accel;
int mnv_accel_create(const mnv_tree_view *device_tree, void *hip_stream, mnv_accel **out);
/* The same with room for max_capacity chunks, so that mnv_accel_refresh can follow a tree that grows by refinement. */
int mnv_accel_create_reserved(const mnv_tree_view *device_tree, int64_t max_capacity, void *hip_stream, mnv_accel **out);
/*
 * Bring the accel up to date after a refinement step on the same device arrays, without rebuilding it:
 *   tree           the view with its NEW capacity; chunks [old_capacity, tree->capacity) were appended under existing
 *                  leaves (mnv_add_children_and_generate_samples + mnv_apply_split_results); needs tree->parent then
 *   changed_nodes  device int32 [n_changed][2] (chunk, child) of existing leaves whose data row was rewritten
 *                  (mnv_apply_sample_results), or NULL
 * Node words, colour rows and chunk depths of the affected voxels are patched, and of the lookup grids exactly the cells
 * those voxels cover.  After mnv_prune_tree, which renumbers chunks, destroy and create instead.  Synchronises hip_stream.
 */
int mnv_accel_refresh(mnv_accel *accel, const mnv_tree_view *tree, int32_t old_capacity, const int32_t *changed_nodes,
                      int32_t n_changed, void *hip_stream);
/* Rebuild every derived array from the tree in place (e.g. after mnv_prune_tree renumbered the chunks): the reserved arrays are
 * reused, nothing is reallocated unless the lookup-grid levels change.  Synchronises hip_stream. */
int mnv_accel_rebuild(mnv_accel *accel, const mnv_tree_view *tree, void *hip_stream);
void mnv_accel_destroy(mnv_accel *accel);
size_t mnv_accel_device_bytes(const mnv_accel *accel);
/* level of the brick-ordered second lookup grid (0: none): what a report names beside the bytes above */
int32_t mnv_accel_grid2_level(const mnv_accel *accel);
/* levels below that grid which every frame kind resolves WITHOUT walking node words: 1 = the last level is folded into the grid's cell words
 * (a depth-10 tree under a level-9 grid reads no node word at all in plain frames), 2 = the next two levels also come from 64-byte brick
 * records (trees with leaves two or more levels below the grid), 0 = neither (shallow trees).  mnv_accel_refresh patches both with the tree edit,
 * a prune derives them again: the value does not drop between frames of the refinement loop.  Frames are bit-identical either way. */
int32_t mnv_accel_brick_levels(const mnv_accel *accel);
/* How much of the tree those words cover -- a gauge for the one limit they have: an inline cell word names its chunk in 22 bits, relative to
 * the smallest chunk number of that depth (4.19 M numbers from there on); a chunk outside that span is not inline (records / node words
 * answer, bit-identical, one more dependent load).  out4: [0] non-leaf cells of the second grid (= chunks one level below it), [1] of them
 * inline, [2] of them NOT inline although all eight children are leaves (lost to the chunk field; 0 on every tree of the test suite up to
 * 12.7 M chunks), [3] chunks with a brick record.  Waits for the device. */
int mnv_accel_lookup_coverage(const mnv_accel *accel, int64_t out4[4]);
/* Compute units the tuned kernel may fill with its persistent workgroups (8 per unit).  Default (and num_cus <= 0): every unit
 * of the device.  A caller that launches on a stream created with hipExtStreamCreateWithCUMask -- to leave units free for
 * the RCCL kernels of the tile gather, which cannot become resident next to a full set of persistent workgroups -- passes
 * the number of units its mask enables.  New; the reference has no counterpart (auto_cuda_threads, renderer_kernel.cu:14-28,
 * sizes a non-persistent launch). */
int mnv_accel_set_cu_budget(mnv_accel *accel, int32_t num_cus);
int mnv_render_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                            mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, void *hip_stream);
/* the same with the per-pixel inputs of the reference's offscreen == false call shape (mnv_frame_inputs above) */
int mnv_render_voxels_accel_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                               const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, void *hip_stream);

/*
 * Interleaved macro-tile partition of `tile` for multi-GPU rendering: the
 * rectangle is cut into macro tiles of tile_w x tile_h pixels (multiples of 8), numbered
 * row-major; this call renders the macro tiles m with m % world == rank.  Outputs are compact
 * and local-tile-major: local tile j = m / world is stored at
 * rgba_out[j][tile_h][tile_w][4] (pixels outside `tile` are left untouched), which is the
 * contiguous buffer each rank hands to the RCCL gather.  world <= 1 with tile_w == 0 is the plain call; world == 1 with a
 * tile size gives the same macro-tile-major layout from a single rank.
 */
typedef struct mnv_partition {
    int32_t rank, world;
    int32_t tile_w, tile_h;
    /* 0: macro tile m belongs to rank m % world (local index m / world).  M >= 2: tiles are still dealt in rounds of `world`, but every
     * M-th round leaves rank 0 out -- rank 0 renders (M - 1) / M of a plain share.  For the rank that also receives the gather and
     * un-permutes the frames (measured on one MI355X for world 8: that work stretches its march by 19 %, tools/root_emulation.py). */
    int32_t root_period;
} mnv_partition;
/* number of local macro tiles of `rank` (the leading dimension of its output buffer) */
int32_t mnv_partition_local_tiles(mnv_rect tile, mnv_partition part);
int mnv_render_voxels_accel_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                 mnv_rect tile, mnv_partition part, float *rgba_out, uint8_t *rgba8_out,
                                 void *hip_stream);

/*
 * Colour arithmetic of the tuned SH march, per accel (default 0).  0: every value is computed exactly as the arithmetic
 * specification says and frames are bit-identical to the oracle.  1: the colour sigmoid
 * weight / (1 + exp(-dot)) uses the hardware exp2 and reciprocal (about 1 ulp each).  Opacity, transmittance, step sizes and
 * every branch stay exact -- SURVEY.md section 7: colour-only arithmetic is continuous and feeds no control flow -- so alpha and
 * the step sequence are unchanged and colours move by ~1e-7 (bound asserted in the tests: 2e-6; contract 1e-4).
 * Trackers, sample emission, depth mode and RGBA-format trees are unaffected.
 */
/* mode 0 exact, 1 fast.  Two renderers of one process can differ; launches already queued keep the mode they were launched with.  (There is no
 * process-wide switch: no call's arithmetic depends on state outside its arguments.) */
int mnv_accel_set_colour_math(mnv_accel *accel, int mode);

/*
 * (mnv_render_voxels builds, in front of every launch of at least 65536 rays, a dense level-7 lookup table of the tree on the caller's stream --
 * stream-ordered scratch memory, about 10 us, 16 MB; csrc/mnv_march_ref_layout.hip -- because the caller's arrays may change between calls and
 * nothing is kept.  Smaller launches (tiles) derive a 512-cell table per workgroup instead.  Frames are bit-identical either way; the threshold
 * is not a switch of this library -- the test-hook build alone can move it, so that the tests run either path at any size.)
 */
/*
 * A memory for the stateless entry point.  viewer::render_voxels (renderer_kernel.hpp:23-34) takes the tree's arrays with every call
 * and keeps nothing, because its caller may edit them in between (the refinement of cuda_renderer.cpp:205-381 does); mnv_render_voxels
 * honours that by default and pays for it: it walks the reference's arrays (one stream: 0.62 ms per 1080p frame of cfg2 against 0.46 ms on
 * the packed re-layout).  mnv_set_tree_cache(1) (process-wide, default 0) lets it keep, per tree -- identified by the addresses of
 * `child` and `data`, `capacity` and the row format; up to four trees, least recently used out -- the packed re-layout of
 * mnv_accel_create, built on the caller's stream at the first call (tens of milliseconds, once) and used for every later plain frame
 * -- with the refinement trackers as well (the reference passes both tensors with every call, cuda_renderer.cpp:141-142; sample counts from
 * the call's view); only frames that ask for visit marks walk the arrays.  Frames and tracker rows are bit-identical either way.
 *   THE RULE: after changing the contents of a cached tree's arrays in place, call mnv_tree_invalidate(child) (NULL: every tree) before the
 *   next frame; a tree that moved or grew (other addresses, other capacity) is a new tree by itself.  Both calls wait for the device.
 *   FREEING a cached tree's arrays counts as changing them: an allocator that hands the same addresses to the next tree of the same
 *   capacity and row format (torch's caching allocator does) would otherwise be answered with the old tree's frames -- invalidate when a
 *   tree dies.  offset / scale are not part of a tree's identity: every frame uses the ones of the view it was called with.
 *   A tree whose re-layout cannot be built (deeper than 23 levels, out of memory) is remembered as such and rendered on the stateless
 *   path without another attempt until it is invalidated.
 * Memory: the re-layout is about 3.6 x the tree (cfg2: 2.6 GB beside 0.72 GB).
 */
void mnv_set_tree_cache(int enable);
void mnv_tree_invalidate(const void *child);

/*
 * The un-permute step on the gathering rank: `gathered` is what the RCCL gather of the ranks'
 * compact buffers produces, [world][n_frames][ceil(macro tiles / world)][tile_h][tile_w] pixels (n_frames = 1 for
 * mnv_render_voxels_accel_part), `frames` receives [n_frames][height][width] pixels.  bytes_per_pixel: 4 (RGBA8) or
 * 16 (float RGBA).  part.rank is ignored; part.root_period must be the one the ranks rendered with.
 */
int mnv_assemble_tiles(const void *gathered, void *frames, int32_t width, int32_t height, mnv_partition part, int32_t n_frames,
                       int32_t bytes_per_pixel, void *hip_stream);

/*
 * The gather that precedes it: the only collective of the path, RCCL over xGMI, one process per GPU (the
 * reference renders on one device, src/renderer/cuda_renderer.cpp:68-163, so there is no reference interface to cite).
 * RCCL (librccl.so.1) is bound at the first mnv_comm_* call; MNV_E_NO_RCCL when it cannot be loaded.
 *
 *   mnv_comm_get_unique_id   one rank (the root) draws the 128-byte id; the host program hands it to the other ranks
 *                            (pipe, shared memory, torch.distributed store -- the library does not care)
 *   mnv_comm_init_rank       collective over the `world` ranks; binds the communicator to the calling thread's current device
 *   mnv_gather_tiles         asynchronous on `hip_stream`: rank r's `local` (bytes_per_rank bytes, device) lands at
 *                            gathered + r * bytes_per_rank on the root (`gathered` is ignored elsewhere): grouped ncclRecv on the
 *                            root, one ncclSend on every other rank, a device copy for the root's own share (a send / receive
 *                            to itself when world == 1, so that a one-GPU box runs the RCCL path for real)
 * A communicator is used from one host thread at a time; calls on it are ordered like RCCL calls (same order on every rank).
 */
#define MNV_COMM_ID_BYTES 128
typedef struct mnv_comm mnv_comm;
int mnv_comm_get_unique_id(void *id_out /* MNV_COMM_ID_BYTES */);
int mnv_comm_init_rank(const void *id, int32_t world, int32_t rank, mnv_comm **out);
int32_t mnv_comm_rank(const mnv_comm *comm);
int32_t mnv_comm_world(const mnv_comm *comm);
int32_t mnv_comm_rccl_version(void); /* ncclGetVersion of the library that was bound, 0 if none */
int mnv_gather_tiles(mnv_comm *comm, const void *local, void *gathered, size_t bytes_per_rank, int32_t root, void *hip_stream);
/*
 * Refinement on several ranks (BASELINE.json configs[4]; SURVEY.md 8(e): "run identical deterministic refinement on every rank"): every
 * rank holds the whole tree, renders its macro tiles of the tracker frame and then needs ALL tracker rows to cast the same votes
 * (cuda_renderer.cpp:205-227 counts candidates over the whole frame; the count does not depend on the order of the rows).
 *   mnv_allgather            in place, asynchronous on `hip_stream`: `table` is [world][bytes_per_rank] on every rank, rank r has written
 *                            block r; afterwards every rank holds every block (grouped ncclSend / ncclRecv between all pairs: on xGMI
 *                            each pair has its own link)
 *   mnv_merge_visit_marks    visited[c] = max over r of table[r][c]: the union of the ranks' visit marks (each rank marks the chunks ITS
 *                            rays went through; a prune needs the frame's), after mnv_allgather of the per-rank mark arrays
 */
int mnv_allgather(mnv_comm *comm, void *table, size_t bytes_per_rank, void *hip_stream);
int mnv_merge_visit_marks(const int32_t *table, int32_t world, int32_t capacity, int32_t *visited, void *hip_stream);
void mnv_comm_destroy(mnv_comm *comm);

/*
 * Several frames in ONE launch: cams[0 .. n_cams) (same image size, same options, same tile /
 * partition); frame f is written at rgba_out + f * frame_elems * 4 (frame_elems = tile.w * tile.h, or
 * ceil(macro_tiles / world) * tile_w * tile_h under a partition -- the same on every rank, so that
 * the per-rank buffers have one shape even when a rank owns one tile fewer).  The ray queues span the frames of the
 * batch (frame-major), so wavefronts that finish their share of one frame carry on with the next: the tail of a frame overlaps
 * the start of the next -- for a camera path this is ~2x faster than one launch per frame.  n_cams <= MNV_MAX_BATCH.
 */
#define MNV_MAX_BATCH 64
int mnv_render_voxels_accel_batch(const mnv_accel *accel, const mnv_camera *cams, int32_t n_cams,
                                  const mnv_render_options *opt, mnv_rect tile, mnv_partition part, float *rgba_out,
                                  uint8_t *rgba8_out, void *hip_stream);

/*
 * The tuned march with the refinement trackers of render_voxels_trace_ray
 * (include/cuda/rt_core.cuh:179-180,237-252,308-321; consumed by cuda_renderer.cpp:150-175):
 * split_track / sample_track are [tile.h][tile.w][3] float rows (priority, chunk, child) exactly
 * as mnv_render_voxels writes them; sample_counts is the tree's live [capacity][N^3] int16 array in
 * the reference layout (it changes between frames, so it is not part of the accel) or NULL.
 * opt->max_depth / opt->max_sample_count bound the candidates.  The `visited` marks
 * (track_visit): mnv_render_voxels_accel_visit.
 * Either tracker may be NULL; with both NULL this is mnv_render_voxels_accel.
 */
int mnv_render_voxels_accel_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                  mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                  float *sample_track, const int16_t *sample_counts, void *hip_stream);

/* ------------------------------------------------ guided sampling kernels (BASELINE config 5) */

/* Cluster grid over world y,z used to route samples to sub-module MLPs
 * (model attributes grid_dim / min_position / range, cuda_renderer.cpp:524-539). */
typedef struct mnv_cluster_grid {
    int32_t grid_dim[2];
    float min_position[3];
    float range[3];
} mnv_cluster_grid;

/*
 * viewer::get_samples_from_voxels (include/cuda/renderer_kernel.hpp:36-52,
 * src/cuda/renderer_kernel.cu:329-363,439-485; rt_core.cuh:418-576), offscreen: the same march as
 * render_voxels, but every dense step emits (z, world xyz[, view dir][, embedding]) instead of colour.
 *   num_samples     device int16 [h*w], in/out; the caller zero-fills it (cuda_renderer.cpp:109)
 *   samples         device float [h*w][opt->max_guided_samples][samples_dim]; only emitted rows are
 *                   written (the caller pre-fills column 0 with -1, cuda_renderer.cpp:110)
 *   samples_dim     4 + 3 * need_viewdir + (appearance_embedding != -1)
 *   cluster_indices device int16 [h*w][max_guided_samples]
 *   split_track / sample_track / visited as in mnv_render_voxels
 */
int mnv_get_samples_from_voxels(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt,
                                mnv_rect tile, float *split_track, float *sample_track, int32_t *visited,
                                int track_visit, int16_t *num_samples, float *samples, int32_t samples_dim,
                                int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream);

/* ... with the depth attachment of the reference's offscreen == false call (renderer_kernel.cu:354-357: `float t_max = surf2Dread(surf_obj_depth)`):
 * inputs->tmax_px [tile.h][tile.w] limits every ray; inputs == NULL or tmax_px == NULL is the call above.  rgba8_init is not read (no image is written). */
int mnv_get_samples_from_voxels_ex(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                   const mnv_frame_inputs *inputs, float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                   int16_t *num_samples, float *samples, int32_t samples_dim, int16_t *cluster_indices,
                                   const mnv_cluster_grid *grid, void *hip_stream);

/* The same march on the packed accel (visit marks: mnv_get_samples_from_voxels_accel_visit below); sample_counts is the tree's
 * live [capacity][8] array or NULL.  Bit-identical rows. */
int mnv_get_samples_from_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                      float *split_track, float *sample_track, const int16_t *sample_counts, int16_t *num_samples,
                                      float *samples, int32_t samples_dim, int16_t *cluster_indices, const mnv_cluster_grid *grid,
                                      void *hip_stream);

/*
 * Both tracker marches on the packed accel WITH visit marks (the role of render_voxels / get_samples_from_voxels called with
 * track_visit = true, cuda_renderer.cpp:101-102,112-114,141-142).  The reference marks every chunk of every descent
 * (query_single_from_root, rt_core.cuh:132-134); the packed layout skips most of the descent, so the march marks the chunk of every
 * leaf it steps through and a second small kernel closes the marks under `parent` -- the same array, element for element.
 * `visited` NULL = the plain tracker entry points; `parent` is the tree's device parent array [capacity].
 */
int mnv_render_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, float *rgba_out,
                                  uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                  const int32_t *parent, void *hip_stream);
/* ... in the reference's LIVE call shape (mnv_frame_inputs, offscreen == false: the tracker frame of cuda_renderer.cpp:141-142 with its depth
 * image and the image under the volume); trackers and `visited` may be NULL (then mnv_render_voxels_accel_ex) */
int mnv_render_voxels_accel_visit_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                     const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                                     const int16_t *sample_counts, int32_t *visited, const int32_t *parent, void *hip_stream);
int mnv_get_samples_from_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                            float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                            const int32_t *parent, int16_t *num_samples, float *samples, int32_t samples_dim,
                                            int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream);
/* ... and with the ray limits of offscreen == false (as mnv_get_samples_from_voxels_ex) */
int mnv_get_samples_from_voxels_accel_visit_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                               const mnv_frame_inputs *inputs, float *split_track, float *sample_track, const int16_t *sample_counts,
                                               int32_t *visited, const int32_t *parent, int16_t *num_samples, float *samples, int32_t samples_dim,
                                               int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream);
/* The tracker frame of one rank of a multi-GPU run: pixels AND tracker rows of the macro tiles `part` assigns to the rank, both in the
 * compact tile-major order of mnv_render_voxels_accel_part (row p of a tracker belongs to pixel p of the rank's buffer); rows of tiles a
 * ragged partition leaves out are not written (pre-fill with -1 as for a frame).  `visited` receives the marks of this rank's rays. */
int mnv_render_voxels_accel_visit_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts,
                                       int32_t *visited, const int32_t *parent, void *hip_stream);

/*
 * viewer::render_nerf_results (include/cuda/renderer_kernel.hpp:12-21,
 * src/cuda/renderer_kernel.cu:294-327,365-394; rt_core.cuh:334-416), offscreen: composites
 * per-sample network outputs along every ray.
 *   tree          only offset/scale/format/basis_dim are read (host or device view alike)
 *   sample_values device float [n][value_stride]; sigma is column 3 (rt_core.cuh:365)
 *   z_vals        device float [n]
 *   offsets       device int64 [h*w]: inclusive prefix sums of the per-ray sample counts
 */
int mnv_render_nerf_results(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt,
                            mnv_rect tile, const float *sample_values, int32_t value_stride, const float *z_vals,
                            const int64_t *offsets, float *rgba_out, uint8_t *rgba8_out, void *hip_stream);

/* --------------------------------------------- refinement kernels (BASELINE config 5) */

/* Mutable device view of the tree topology for the refinement kernels (the arrays a TreeSpec exposes
 * non-const, include/data_spec.hpp:26-28). */
typedef struct mnv_tree_edit {
    int32_t *child;   /* [max_capacity][8] */
    int32_t *parent;  /* [max_capacity] */
    float offset[3];
    float scale[3];
    int32_t N;        /* 2 */
    int32_t capacity; /* chunks in use BEFORE the call */
} mnv_tree_edit;

/*
 * viewer::add_children_and_generate_samples (include/cuda/renderer_kernel.hpp:54-63,
 * src/cuda/renderer_kernel.cu:170-198,487-510): appends num_parents chunks at [capacity, capacity +
 * num_parents), links them under parent_nodes[i] = (chunk, child), copies the visit mark, and turns the
 * caller's uniform [0,1) numbers in samples[num_parents*8][samples_per_corner][samples_dim] into
 * world-space sample points inside each new voxel (+ view dir / embedding columns, + cluster ids).
 */
int mnv_add_children_and_generate_samples(const mnv_tree_edit *tree, const mnv_render_options *opt,
                                          const int32_t *parent_nodes, int32_t num_parents, float *samples,
                                          int32_t samples_dim, int16_t *cluster_indices, int32_t *visited,
                                          const mnv_cluster_grid *grid, void *hip_stream);
/* viewer::generate_samples (renderer_kernel.hpp:65-73, renderer_kernel.cu:200-213,512-534): the same
 * sample generation for existing voxels nodes[i] = (chunk, child). */
int mnv_generate_samples(const mnv_tree_edit *tree, const mnv_render_options *opt, const int32_t *nodes,
                         int32_t num_items, float *samples, int32_t samples_dim, int16_t *cluster_indices,
                         const mnv_cluster_grid *grid, void *hip_stream);
/* viewer::adjust_parents_and_children (renderer_kernel.hpp:75-79, renderer_kernel.cu:63-86,536-549):
 * fixes relative child offsets and parent ids of chunks [first_shift_index, capacity) before compaction;
 * to_delete is a bool (1 byte) array, index_shifts an int32 array, both [capacity]. */
int mnv_adjust_parents_and_children(const mnv_tree_edit *tree, int32_t first_shift_index, const uint8_t *to_delete,
                                    const int32_t *index_shifts, void *hip_stream);

/* --------------------------- refinement: what the reference does with the trackers between frames
 * (VolumeRenderer::Impl::expand_voxels / get_more_samples / prune_tree, src/renderer/cuda_renderer.cpp:205-381,
 * there a chain of libtorch tensor ops; here device-resident, one call per step).  These calls
 * synchronise `hip_stream` before returning because they hand counts back to the host, as the
 * reference's `.size(0)` / `.item()` reads do. */

/*
 * expand_voxels' vote (cuda_renderer.cpp:205-227): among the rows (priority, chunk, child) of split_track
 * with chunk >= 0, count identical rows, keep those proposed by >= 2 rays, order by (count descending,
 * priority, chunk, child ascending) -- torch::unique_dim's lexicographic order on (-count, row) -- and write
 * the first min(max_out, n) as int32 (chunk, child) pairs to nodes_out (device, [max_out][2]).
 *   n_out         host: pairs written            n_candidates  host: rows that qualified ("Split candidates: N")
 * Rows hold integer-valued floats (|priority| < 32768, 0 <= chunk < 2^31, 0 <= child < 8), which is what the
 * march writes.  max_out = opt.split_batch_size.
 */
int mnv_select_split_candidates(const float *split_track, int64_t n_rows, int32_t max_out, int32_t *nodes_out,
                                int32_t *n_out, int32_t *n_candidates, void *hip_stream);
/* get_more_samples' selection (cuda_renderer.cpp:281-296): unique rows of sample_track with chunk >= 0 in
 * ascending (priority, chunk, child) order, no vote threshold; first min(max_out, n) as (chunk, child). */
int mnv_select_sample_candidates(const float *sample_track, int64_t n_rows, int32_t max_out, int32_t *nodes_out,
                                 int32_t *n_out, int32_t *n_candidates, void *hip_stream);
/*
 * cuda_renderer.cpp:262-270: rows [capacity*8, (capacity + num_parents)*8) of data (device binary16,
 * [max_capacity][8][data_dim]) become the mean over samples_per_corner of results[num_parents*8]
 * [samples_per_corner][result_stride] (device float, result_stride >= data_dim; the reference's is
 * data_dim + 1); sample_counts (may be NULL) of the new chunks is set to samples_per_corner.  `capacity` is
 * the chunk count BEFORE the split; the caller then adds num_parents to it.  Mean = fp32 sum in sample order /
 * n, rounded once to binary16 (torch's reduction order is unspecified: agreement is to 1 binary16 ulp).
 */
int mnv_apply_split_results(uint16_t *data, int16_t *sample_counts, int32_t capacity, int32_t num_parents,
                            const float *results, int32_t result_stride, int32_t samples_per_corner, int32_t data_dim,
                            void *hip_stream);
/*
 * cuda_renderer.cpp:307-332: running average of existing voxels nodes[i] = (chunk, child) with
 * samples_per_corner new network outputs each: data += (sum_new - half(n_new * data)) / (count + n_new),
 * count += n_new.  The reference expression stops at index_add_ (binary16 self, fp32 source is rejected by
 * libtorch), so the last rounding is this build's choice: one rounding of (old + update) to binary16.
 */
int mnv_apply_sample_results(uint16_t *data, int16_t *sample_counts, const int32_t *nodes, int32_t num_items,
                             const float *results, int32_t result_stride, int32_t samples_per_corner, int32_t data_dim,
                             void *hip_stream);
/*
 * prune_tree (cuda_renderer.cpp:335-381): chunks of [0, tree->capacity) whose visit mark is 0 are deleted,
 * the survivors are compacted in order (child offsets and parent ids fixed up by
 * mnv_adjust_parents_and_children with first_shift_index 0), and all marks but the root's are cleared over
 * [1, max_capacity).  data / tree->child / tree->parent are compacted; sample_counts too when non-NULL (the
 * reference leaves sample_counts uncompacted -- pass NULL for that behaviour).
 *   new_capacity  host: chunks in use after the call     num_deleted  host: 0 = "Nothing can be pruned"
 * Returns MNV_E_INVALID when the root itself is unmarked (no track_visit frame was rendered).
 */
int mnv_prune_tree(const mnv_tree_edit *tree, uint16_t *data, int32_t data_dim, int16_t *sample_counts, int32_t *visited,
                   int32_t max_capacity, int32_t *new_capacity, int32_t *num_deleted, void *hip_stream);
/* The same, and the packed accel of the tree follows in place (NULL: plain mnv_prune_tree): surviving chunks are renumbered in the node
 * words and both lookup grids, voxels whose sub-tree went become leaves, the colour rows are compacted -- about 0.9 ms of traffic for
 * the 1.5 M-chunk tree instead of the 2.9 ms of mnv_accel_rebuild.  The accel must describe the tree as it is before the call. */
int mnv_prune_tree_accel(const mnv_tree_edit *tree, uint16_t *data, int32_t data_dim, int16_t *sample_counts, int32_t *visited,
                         int32_t max_capacity, mnv_accel *accel, int32_t *new_capacity, int32_t *num_deleted, void *hip_stream);

/* torch::rand's role in expand_voxels / get_more_samples (cuda_renderer.cpp:247-250,298-301): n uniform numbers in
 * [0, 1) with 24 random bits each, a pure function of (seed, index) -- reproducible, unlike the reference. */
int mnv_fill_uniform(float *out, int64_t n, uint64_t seed, void *hip_stream);
/*
 * The step between mnv_get_samples_from_voxels and the network (cuda_renderer.cpp:116-121,135):
 * offsets_out[ray] = inclusive prefix sum of num_samples (device int64 [n_rays], what mnv_render_nerf_results
 * takes), and the emitted rows of samples[n_rays][max_guided_samples][samples_dim] packed in ray order:
 * z_vals_out[total] = column 0, rows_out[total][samples_dim - 1] = the remaining columns,
 * clusters_out[total].  total_out (host) = offsets_out[n_rays - 1]; returns MNV_E_INVALID when it exceeds
 * rows_capacity.  With all three outputs NULL only offsets_out and total_out are produced (to size the buffers).
 * Synchronises hip_stream (the total sizes the network launch).
 */
int mnv_compact_guided_samples(const int16_t *num_samples, const float *samples, const int16_t *cluster_indices, int64_t n_rays,
                               int32_t max_guided_samples, int32_t samples_dim, int64_t *offsets_out, float *z_vals_out,
                               float *rows_out, int16_t *clusters_out, int64_t rows_capacity, int64_t *total_out,
                               void *hip_stream);

/* ------------------------------------------------ per-sample sub-module network
 *
 * Stands in for VolumeRenderer::Impl::query_submodules (src/renderer/cuda_renderer.cpp:165-203): every sample
 * row is routed to the network of its cluster and the outputs are scattered back in input order.  The
 * reference loads its networks as TorchScript files that are not part of its repository (model_path,
 * cuda_renderer.cpp:518-539), so PARITY IS UNPINNED for the arithmetic: this is the build's own small MLP
 * (triangle-wave position encoding, `hidden_layers` x `hidden_width` ReLU layers with binary16 activations
 * and fp32 accumulation on the matrix cores, linear fp32 output), checked against its own CPU restatement.
 */
typedef struct mnv_mlp_desc {
    int32_t n_clusters;     /* sub-modules; cluster ids 0 .. n_clusters-1 (cuda_renderer.cpp:529-533) */
    int32_t pos_octaves;    /* encoded position: 3 + 6 * pos_octaves features */
    int32_t dir_octaves;    /* encoded direction: 3 + 6 * dir_octaves features when need_viewdir */
    int32_t need_viewdir;   /* sample rows are xyz, dir[3] (opt.need_viewdir) */
    int32_t n_embeddings;   /* > 0: rows end with an appearance-embedding index (opt.appearance_embedding != -1) */
    int32_t embedding_dim;  /* features per embedding (1 .. 64) */
    int32_t hidden_width;   /* 64 or 128 */
    int32_t hidden_layers;  /* >= 1 */
    int32_t out_dim;        /* tree data_dim + 1 (cuda_renderer.cpp:255-257), <= hidden_width */
    float center[3];        /* position normalisation p = (xyz - center) * inv_extent */
    float inv_extent[3];
} mnv_mlp_desc;
typedef struct mnv_mlp mnv_mlp;
/* binary16 parameters per cluster, in this order, all row-major: W0[hidden][in], b0[hidden], then
 * (hidden_layers - 1) x { W[hidden][hidden], b[hidden] }, Wout[out][hidden], bout[out],
 * embedding[n_embeddings][embedding_dim]; in = 3 + 6 pos_octaves (+ 3 + 6 dir_octaves) (+ embedding_dim). */
size_t mnv_mlp_param_count(const mnv_mlp_desc *desc);
/* params: host, n_clusters * mnv_mlp_param_count() binary16 values, cluster-major. */
int mnv_mlp_create(const mnv_mlp_desc *desc, const uint16_t *params, size_t n_halfs, void *hip_stream, mnv_mlp **out);
void mnv_mlp_destroy(mnv_mlp *mlp);
/*
 * samples: device float [n][samples_stride], columns xyz[, dir][, embedding index] first (the layout
 * mnv_get_samples_from_voxels writes after its z column and the refinement kernels write from column 0);
 * cluster_indices: device int16 [n]; results: device float [n][result_stride], columns [0, out_dim) written.
 * Rows whose cluster id is outside [0, n_clusters) get zeros.  Asynchronous on hip_stream; the handle holds the
 * launch scratch, so one handle serves one stream at a time.
 */
int mnv_query_submodules(mnv_mlp *mlp, const int16_t *cluster_indices, const float *samples, int32_t samples_stride,
                         int64_t n, float *results, int32_t result_stride, void *hip_stream);

/*
 * The guided-sampling frame as ONE kernel (BASELINE.json configs[4]): what the reference does with get_samples_from_voxels,
 * a cumsum / boolean-mask compaction, query_submodules and render_nerf_results (src/renderer/cuda_renderer.cpp:107-139) --
 * and this library's own four entry points above do the same way -- happens inside the march: every lane marches its ray on
 * the packed accel and releases its samples into a per-wavefront pool in LDS; the wavefront evaluates the sub-module network for
 * 64 pooled samples at a time on the matrix cores (weights of the samples' cluster as the MFMA A operand), and the lane that owns a
 * ray composites its samples in ray order.  No sample buffer exists in global memory.  The frame is bit-identical to mnv_get_samples_from_voxels_accel -> mnv_compact_guided_samples ->
 * mnv_query_submodules -> mnv_render_nerf_results (same march, same MFMA sequence, same composite arithmetic).
 * Restrictions (MNV_E_UNSUPPORTED otherwise; use the four-step path): 64-wide networks with at most 64 encoded inputs,
 * RGBA / SH1/4/9/16 trees, no render_depth.  Refinement trackers and visit marks: mnv_render_guided_fused_track.
 *   sample_counter  optional device counter, ONE word: += network evaluations (= what the four-step path reports as guided samples)
 * Two kernels serve these entry points (csrc/mnv_guided_fused2.h, csrc/mnv_guided_fused.h), same frames bit for bit: when the weights
 * of one sub-module fit a workgroup's LDS beside the sample rings (every 64-wide network of up to about 6 layers), a workgroup
 * is three wavefronts that only march and push their samples into rings in LDS plus one wavefront that only evaluates the
 * network -- weights resident in LDS, results handed back through the rings, the owning lanes composite; otherwise every
 * wavefront does both jobs in turn (weights from L2).  mnv_accel_set_fused_kernel selects one explicitly.
 */
int mnv_render_guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                            const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, unsigned long long *sample_counter,
                            void *hip_stream);
/* Which kernel the mnv_render_guided_fused* entry points launch on THIS accel: 0 = choose by the network's size (default),
 * 1 = the one-role kernel (every wavefront marches and evaluates), 2 = producer / consumer wavefronts wherever the rings, the
 * rays' constants and the weights of enough sub-modules fit a workgroup's LDS (the one-role kernel otherwise). */
int mnv_accel_set_fused_kernel(mnv_accel *accel, int version);
/* Diagnostics of the fused kernels on THIS accel: `words32` = NULL (default, none) or a device buffer of 32 64-bit words the
 * kernels add run counts and per-phase times to (tools/guided_bench.py names them).  Costs a few per cent while set.  The library keeps
 * only the address: the buffer must outlive every launch made while it is set (pass NULL before freeing it). */
int mnv_accel_set_fused_diag(mnv_accel *accel, unsigned long long *words32);
/* The producer / consumer kernel waits with s_sleep polls under a watchdog (a bug ends in wrong pixels, never in a hung device).  A
 * wait the watchdog abandons is counted in a device word of the accel -- always, with or without mnv_accel_set_fused_diag.  The frame's own
 * call has long returned (the entry points are asynchronous), so: the NEXT mnv_render_guided_fused* call on this accel prints one line
 * on stderr and returns MNV_E_FAULT once per fault, and mnv_accel_fused_faults reads the count since mnv_accel_create (it waits for
 * the device).  No wait has ever been abandoned in a test or stress run; the count is 0 on a healthy build. */
int mnv_accel_fused_faults(const mnv_accel *accel, uint32_t *count_out);
/* The same frame when refinement is on as well (BASELINE.json configs[4] has both switches on: cuda_renderer.cpp:107-156): the fused
 * kernel also writes the refinement trackers (rows pre-filled with -1 by the caller, as cuda_renderer.cpp:97-98) and, with `visited` +
 * `parent`, the visit marks -- what get_samples_from_voxels produces besides the samples (rt_core.cuh:475-507,561-574).  A ray that
 * has emitted max_guided_samples samples keeps marching here (its trackers follow the remaining steps); the picture is the same. */
int mnv_render_guided_fused_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                                  const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                                  const int16_t *sample_counts, int32_t *visited, const int32_t *parent, unsigned long long *sample_counter,
                                  void *hip_stream);
/* The fused frame in the reference's LIVE call shape (offscreen == false in get_samples_from_voxels and render_nerf_results,
 * cuda_renderer.cpp:111-113,135-136): every ray stops at inputs->tmax_px (renderer_kernel.cu:354-357).  inputs->rgba8_init is not read:
 * render_nerf_results_kernel leaves out[3] at 1 (renderer_kernel.cu:316), so composite_and_write adds the image under the volume with
 * weight 1 - 1 = 0 (:224-234) -- with or without it the frame is the same.  Tracker arguments may be NULL as in mnv_render_guided_fused. */
int mnv_render_guided_fused_track_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                     const mnv_frame_inputs *inputs, const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out,
                                     float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                                     unsigned long long *sample_counter, void *hip_stream);
/* The fused frame of one rank of a multi-GPU run: the macro tiles `part` assigns to the rank, written tile-major like
 * mnv_render_voxels_accel_part (guided sampling reads the tree only, so the ranks need no exchange but the usual tile gather). */
int mnv_render_guided_fused_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                 const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out,
                                 unsigned long long *sample_counter, void *hip_stream);
/* ... with the tracker rows and visit marks of mnv_render_guided_fused_track for the rank's tiles (compact order, as
 * mnv_render_voxels_accel_visit_part): the frame of configs[4] -- refinement and guided sampling both on -- on one rank of several. */
int mnv_render_guided_fused_track_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                       float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                                       unsigned long long *sample_counter, void *hip_stream);

/* A HIP stream whose kernels run on all but `reserve_cus` compute units (hipExtStreamCreateWithCUMask; the units are taken
 * evenly from the XCDs, and from their shader engines when reserve_cus is a multiple of 32).  The tuned kernel is persistent and
 * fills every wave slot of the units it may use, so kernels of other streams -- the RCCL channel workgroups of the multi-GPU
 * tile gather (about 288 VGPRs per lane: they do not fit beside more than three march workgroups) -- would otherwise wait
 * for a whole launch to drain.  Launch the march on this stream and give the accel the matching budget
 * (mnv_accel_set_cu_budget(accel, *enabled_cus)).  reserve_cus == 0: an ordinary non-blocking stream.  New: the reference
 * renders on one GPU and has no gather. */
int mnv_stream_create_reserved(int32_t reserve_cus, void **stream_out, int32_t *enabled_cus);
int mnv_stream_destroy(void *stream);

/* (Device times: every entry point launches on the caller's stream and records nothing itself -- bracket the call with two HIP events on that
 * stream, as bench.py does for roofline.achieved.) */

/* ------------------------------------------ host-side data model (C++ core) */

/* Opaque handle to the C++ viewer::N3Tree (host arrays + optional device copy);
 * mirrors include/n3tree/n3tree.hpp:17-69. */
typedef struct mnv_n3tree mnv_n3tree;

int mnv_n3tree_open(const char *npz_path, mnv_n3tree **out);          /* N3Tree::open, n3tree.cpp:16-205 */
int mnv_n3tree_from_arrays(const mnv_tree_view *host_view, mnv_n3tree **out); /* copies */
void mnv_n3tree_free(mnv_n3tree *t);
int mnv_n3tree_host_view(const mnv_n3tree *t, mnv_tree_view *view);
/* N3Tree::move_to_device (n3tree.cpp:207-246): allocates [max_capacity,...] device
 * arrays, copies the first `capacity` rows, builds the accel. */
int mnv_n3tree_move_to_device(mnv_n3tree *t, int64_t max_capacity, int need_parent,
                              int need_sample_counts, void *hip_stream);
int mnv_n3tree_device_view(const mnv_n3tree *t, mnv_tree_view *view);
const mnv_accel *mnv_n3tree_accel(const mnv_n3tree *t);
int mnv_n3tree_save_npz(const mnv_n3tree *t, const char *npz_path); /* svox layout, stored (no deflate) */
/* DataFormat::parse / to_string (src/data_format.cpp:5-41) */
void mnv_data_format_parse(const char *str, int32_t *format, int32_t *basis_dim);
int mnv_data_format_to_string(int32_t format, int32_t basis_dim, char *buf, size_t buflen);

/* ------------------------------------------------ the VolumeRenderer role (batch / offscreen)
 * viewer::VolumeRenderer (include/renderer/renderer.hpp:9-39, src/renderer/cuda_renderer.cpp) behind the C ABI,
 * for hosts that are not C++: it owns a camera, the options and an internal stream; set() uploads the tree
 * (move_to_device(max_tree_capacity, true, true)), render() draws one frame and -- with a model loaded and
 * options.use_splitting / use_guided_sampling -- runs the reference's refinement loop
 * (cuda_renderer.cpp:98-156: trackers -> expand_voxels / get_more_samples -> prune_tree). */
typedef struct mnv_renderer mnv_renderer;
typedef struct mnv_renderer_stats {  /* what the reference prints per frame */
    int32_t track_visit, used_accel, full;
    int32_t split_candidates, added;       /* expand_voxels: "Split candidates: N", "Added: N" */
    int32_t sample_candidates, resampled;  /* get_more_samples */
    int32_t pruned;                        /* prune_tree: reclaimed chunks, -1 = nothing to prune, 0 = did not run */
    int64_t guided_samples;                /* rows sent to the networks by guided sampling */
    int64_t capacity;                      /* chunks in use after the frame */
    int32_t fused;                         /* the guided-sampling frame ran as one kernel (mnv_render_guided_fused) */
    int32_t reserved;
} mnv_renderer_stats;
int mnv_renderer_create(mnv_renderer **out);
void mnv_renderer_destroy(mnv_renderer *r);
/* VolumeRenderer::set (cuda_renderer.cpp:498-516); the tree must outlive the renderer or the next set() */
int mnv_renderer_set(mnv_renderer *r, mnv_n3tree *tree, int64_t max_tree_capacity);
/* load_model (cuda_renderer.cpp:518-539) from an .npz container: mlp_desc int32[9] (the mnv_mlp_desc integers in
 * order), mlp_center f32[3], mlp_inv_extent f32[3], mlp_params binary16, grid_dim int[2], min_position f32[3],
 * max_position f32[3] */
int mnv_renderer_load_model(mnv_renderer *r, const char *npz_path);
int mnv_renderer_set_model(mnv_renderer *r, const mnv_mlp_desc *desc, const uint16_t *params, size_t n_halfs,
                           const mnv_cluster_grid *grid);
int mnv_renderer_resize(mnv_renderer *r, int32_t width, int32_t height);
/* the renderer's RenderOptions, mutable in place (VolumeRenderer::options) */
mnv_render_options *mnv_renderer_options(mnv_renderer *r);
/* VolumeRenderer::camera pose and focal length (fx <= 0 keeps the current one; fy <= 0 means fy = fx) */
int mnv_renderer_set_camera(mnv_renderer *r, float fx, float fy, const float center[3], const float v_back[3],
                            const float v_world_up[3]);
/* seed of the sample jitter (torch::rand in the reference); accel_rebuild_after < 0 keeps the default */
int mnv_renderer_set_seed(mnv_renderer *r, uint64_t seed, int32_t accel_rebuild_after);
int mnv_renderer_render(mnv_renderer *r, mnv_renderer_stats *stats /* may be NULL */);
/* wait for the frame and copy it to host buffers [height][width][4] (either may be NULL) */
int mnv_renderer_download(mnv_renderer *r, float *rgba_host, uint8_t *rgba8_host);
/*
 * Frames in flight (VolumeRenderer::frames_in_flight, default 3).  The reference issues one render_voxels call per frame on
 * one stream (src/renderer/cuda_renderer.cpp:141-142); here plain frames (no refinement, packed accel current) rotate over
 * `count` slots, each with its own HIP stream and frame buffers, so the tail of one launch overlaps the next launches.
 * mnv_renderer_render returns at once; mnv_renderer_last_slot names the slot it rendered into and
 * mnv_renderer_download_slot waits for that slot's frame only.  count = 1 restores the one-stream behaviour.
 */
int mnv_renderer_set_frames_in_flight(mnv_renderer *r, int32_t count);
/* VolumeRenderer::guided_in_flight (default off): guided-sampling frames that change nothing -- use_guided_sampling without use_splitting, a
 * network the fused kernel covers, a tree below 3/4 of its capacity -- rotate over the slots like plain frames (1.20 -> 1.09 ms per 1080p
 * frame on the cfg2 tree with three in flight).  Such a frame's sample count is not known when mnv_renderer_render returns: the stats carry
 * guided_samples = -1 and mnv_renderer_slot_guided_samples waits for the frame of `slot` and returns its count. */
int mnv_renderer_set_guided_in_flight(mnv_renderer *r, int enable);
int mnv_renderer_slot_guided_samples(mnv_renderer *r, int32_t slot, int64_t *count_out);
/* VolumeRenderer::use_fused_guided (default on): guided-sampling frames that need nothing but the picture run as one kernel;
 * off = always the four steps of cuda_renderer.cpp:107-139 (sample march, compaction, networks, composite) */
int mnv_renderer_set_fused_guided(mnv_renderer *r, int enable);
/* The reference's render loop passes offscreen == false to all three launchers (cuda_renderer.cpp:111-113,135-136,141-142): the frame is limited by
 * the GL pass's depth attachment and composited over its image.  tmax_px [height][width] float / rgba8_init [height][width][4] uint8: device
 * arrays of the caller that stay valid while frames are rendered (either may be NULL; both NULL = the offline renderer again, the default).
 * Single-rank rendering only. */
int mnv_renderer_set_frame_inputs(mnv_renderer *r, const float *tmax_px, const uint8_t *rgba8_init);
/* VolumeRenderer::set_ranks: several ranks (one process per GPU, each with its own renderer over the same scene and networks) render and
 * refine in lock step -- this rank marches its macro tiles, tracker rows and visit marks are all-gathered, the same refinement runs on
 * every replica, rank 0's frame is the whole picture.  comm NULL = back to one rank.  The communicator stays the caller's. */
int mnv_renderer_set_ranks(mnv_renderer *r, mnv_comm *comm, int32_t tile_w, int32_t tile_h);
int32_t mnv_renderer_last_slot(const mnv_renderer *r);
int mnv_renderer_download_slot(mnv_renderer *r, int32_t slot, float *rgba_host, uint8_t *rgba8_host);
/* copy the (refined) device tree back into the mnv_n3tree's host arrays */
int mnv_renderer_sync_tree(mnv_renderer *r);

/* ------------------------------------------------ deterministic synthetic trees */
/* Integer-hash PRNG, IEEE-only arithmetic: bit-identical on every host. */

typedef struct mnv_synth_random_params {
    int32_t depth;          /* chunk levels; finest voxel 2^-depth */
    int32_t format;         /* MNV_FORMAT_* */
    int32_t basis_dim;      /* SH: 1,4,9,16,25; RGBA: -1 */
    float refine_prob;      /* probability a non-finest voxel is refined */
    float empty_prob;       /* probability a leaf has sigma = 0 */
    float sigma_max;        /* dense leaves: sigma ~ U(0, sigma_max) */
    float coef_sd;          /* SH-DC ~ approx N(0, coef_sd^2), higher orders decay by 1/2 per degree */
    float offset[3];
    float scale[3];         /* invradius3 */
    uint64_t seed;
} mnv_synth_random_params;

typedef struct mnv_synth_shell_params {
    int32_t depth;          /* 10 for the headline config */
    int32_t basis_dim;      /* 9 for the headline config (format SH) */
    float radius;           /* 0.35 unit-cube units */
    float half_thickness;   /* 1.5/1024 */
    float sigma_lo, sigma_hi; /* 50, 400 */
    float offset[3];
    float scale[3];
    uint64_t seed;
} mnv_synth_shell_params;

/* "Merged Mega-NeRF" stand-in (BASELINE configs[2]): anisotropic volume with a
 * terrain-like occupied layer x = h(y, z), cut into bricks_y x bricks_z sub-modules with independent seeds. */
typedef struct mnv_synth_terrain_params {
    int32_t depth;
    int32_t basis_dim;
    int32_t bricks_y, bricks_z;  /* 4 x 2 */
    int32_t noise_cells;         /* lattice cells of the value noise per unit length */
    float base, amplitude;       /* height range in unit-cube x: [base, base + amplitude] */
    float thickness;             /* half thickness of the occupied layer, unit-cube units */
    float sigma_lo, sigma_hi;
    float offset[3];
    float scale[3];              /* invradius3, e.g. (0.5, 0.125, 0.125) for a 1 : 4 : 4 world extent */
    uint64_t seed;
} mnv_synth_terrain_params;

int mnv_synth_random_tree(const mnv_synth_random_params *p, mnv_n3tree **out);
int mnv_synth_terrain_tree(const mnv_synth_terrain_params *p, mnv_n3tree **out);
int mnv_synth_shell_tree(const mnv_synth_shell_params *p, mnv_n3tree **out);

#ifdef __cplusplus
}
#endif
#endif /* MNV_H */
