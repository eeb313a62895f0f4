/*
 * mnv_oracle.h -- CPU parity oracle for the N3Tree (PlenOctree) ray-march path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker.  The product (libmnv.so) never links or calls
 * into this file.
 *
 * What it restates (all paths relative to /root/reference):
 *   src/cuda/renderer_kernel.cu:30-38    screen2worlddir
 *   src/cuda/renderer_kernel.cu:40-61    rodrigues
 *   src/cuda/renderer_kernel.cu:215-241  composite_and_write (offscreen branch)
 *   src/cuda/renderer_kernel.cu:243-292  render_voxels_kernel (per-pixel driver)
 *   include/cuda/rt_core.cuh:12-68       maybe_precalc_basis
 *   include/cuda/rt_core.cuh:70-100      _dda_world / _dda_unit
 *   include/cuda/rt_core.cuh:102-115     _get_delta_scale
 *   include/cuda/rt_core.cuh:117-159     query_single_from_root
 *   include/cuda/rt_core.cuh:162-332     render_voxels_trace_ray
 *   src/camera.cpp:54-82                 Camera::_update pose math
 *
 * Arithmetic specification (see DESIGN.md "Arithmetic spec"): the reference
 * source read literally under C++ usual arithmetic conversions, IEEE-754
 * binary32/binary64, round-to-nearest-even, NO fused multiply-add contraction,
 * expf = glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c algorithm (restated in
 * orc_expf, checked bit-for-bit against this container's libm in
 * tests/test_oracle_math.py), powf(2, depth) = exact 2^depth.
 *
 * Pinning status: the reference ships no tests, golden vectors or fixtures for
 * this path.  The oracle is pinned against outputs of the
 * reference's own device code (include/cuda/rt_core.cuh) compiled for gfx950 by
 * the recipe in oracle/Makefile.ref into oracle/_ref/ and run on an MI355X; the
 * resulting vectors are committed under tests/golden/ (see tests/golden/README.md): frames, the refinement trackers and
 * visit marks (ref_trackers_*.npz), the guided-sampling pair and the three refinement kernels.
 */
#ifndef MNV_ORACLE_H
#define MNV_ORACLE_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BASIS_MAX 25 /* render_options.hpp:4 VIEWER_GLOBAL_BASIS_MAX */

/* Host view of an N3Tree in the reference's own layout (n3tree.cpp:92-97,183-188). */
typedef struct {
    const uint16_t *data;          /* [capacity][N^3][data_dim] IEEE binary16 bits */
    const int32_t *child;          /* [capacity][N^3] relative chunk offsets, 0 = leaf */
    const int16_t *sample_counts;  /* [capacity][N^3], may be NULL (trackers then skip it) */
    float offset[3];
    float scale[3];
    int32_t N;
    int32_t data_dim;
    int32_t format;                /* 0 = RGBA, 1 = SH (data_format.hpp:8-12) */
    int32_t basis_dim;             /* data_format.hpp:15 */
    int32_t capacity;
} orc_tree;

/* data_spec.hpp:9-23 CameraSpec, with the 12-float c2w by value. */
typedef struct {
    int32_t width, height;
    float fx, fy, cx, cy;
    float c2w[12];                 /* column-major right|up|back|center (camera.cpp:55-82) */
} orc_camera;

/* render_options.hpp:9-56, field for field. */
typedef struct {
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
} orc_options;

/* Integer work counters (SURVEY.md section 8(d) algorithmic-bytes formula). */
typedef struct {
    uint64_t rays;          /* rays rendered */
    uint64_t rays_in_bbox;  /* rays that passed the render_bbox slab test */
    uint64_t rays_hit;      /* rays with >= 1 dense (sigma > sigma_thresh) sample */
    uint64_t steps;         /* march steps (iterations of rt_core.cuh:220) */
    uint64_t levels;        /* child-word reads (iterations of rt_core.cuh:131) */
    uint64_t hits;          /* dense samples (rt_core.cuh:233 taken) */
    uint64_t early_stops;   /* rays ended by rt_core.cuh:295 */
    uint64_t max_steps;     /* max steps on one ray */
} orc_counters;

void orc_default_options(orc_options *opt);        /* struct defaults, render_options.hpp */
float orc_expf(float x);                           /* glibc 2.35 expf algorithm */
float orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);               /* round-to-nearest-even */
void orc_sh_basis(int basis_dim, const float vdir[3], float out[ORC_BASIS_MAX]);

/* camera.cpp:54-82: pose vectors -> 12-float column-major c2w. */
void orc_camera_pose(const float center[3], const float v_back[3], const float v_world_up[3],
                     float c2w_out[12]);

/* bytes = sum_rays [16 + sum_steps (4 d_s + 2 + hit_s * 6 * basis_dim_eff)]. */
uint64_t orc_algorithmic_bytes(const orc_counters *c, int32_t format, int32_t basis_dim);

/*
 * Render the tile [x0,x0+w) x [y0,y0+h) of the camera's image, offscreen branch
 * (background = opt->background_brightness, t_max = 1e9).
 *   rgba        [h][w][4] float, the four floats *before* the u8 cast (may be NULL)
 *   rgba8       [h][w][4] uint8, renderer_kernel.cu:237 pack (may be NULL)
 *   split_track [h][w][3] float (priority, chunk, child), may be NULL
 *   sample_track[h][w][3] float, may be NULL
 *   visited     [capacity] int32, used when track_visit != 0
 *   steps_out   [h][w] int32 per-ray step count, may be NULL
 *   ctr         accumulated counters, may be NULL
 * Returns 0, or -1 on invalid arguments.  n_threads <= 0 means all cores.
 */
int orc_render_voxels(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                      int32_t x0, int32_t y0, int32_t w, int32_t h,
                      float *rgba, uint8_t *rgba8,
                      float *split_track, float *sample_track,
                      int32_t *visited, int track_visit,
                      int32_t *steps_out, orc_counters *ctr, int n_threads);
/* The same with the two per-pixel inputs of the reference's offscreen == false call shape (renderer_kernel.cu:260-264,277-280,
 * 230-234), indexed like the outputs: tmax_px [h][w] float (NULL: 1e9f), rgba8_init [h][w][4] uint8 (NULL: background_brightness). */
int orc_render_voxels_ex(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                         int32_t x0, int32_t y0, int32_t w, int32_t h,
                         const float *tmax_px, const uint8_t *rgba8_init,
                         float *rgba, uint8_t *rgba8, float *split_track, float *sample_track,
                         int32_t *visited, int track_visit, int32_t *steps_out, orc_counters *ctr,
                         int n_threads);

/* Cluster grid of the guided-sampling path (cuda_renderer.cpp:524-539 model attributes). */
typedef struct {
    int32_t grid_dim[2];
    float min_position[3];
    float range[3];
} orc_cluster_grid;

/*
 * get_samples_from_voxels_kernel + get_samples_trace_ray (renderer_kernel.cu:329-363,
 * rt_core.cuh:418-576), offscreen (t_max = 1e9).  Full frame.
 *   num_samples     [h*w] int16, in/out (the caller zero-fills, cuda_renderer.cpp:109)
 *   samples         [h*w][max_guided_samples][samples_dim] float; only emitted rows are written
 *   cluster_indices [h*w][max_guided_samples] int16
 *   trackers / visited as in orc_render_voxels (full frame, may be NULL)
 */
int orc_get_samples_from_voxels(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                                float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                int16_t *num_samples, float *samples, int32_t samples_dim,
                                int16_t *cluster_indices, const orc_cluster_grid *grid, int n_threads);
/* the same with the reference's offscreen == false input: tmax_px [h][w], the ray limit read from the depth attachment
 * (renderer_kernel.cu:354-357); NULL = 1e9f everywhere = the function above */
int orc_get_samples_from_voxels_ex(const orc_tree *tree, const orc_camera *cam, const orc_options *opt, const float *tmax_px,
                                   float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                   int16_t *num_samples, float *samples, int32_t samples_dim,
                                   int16_t *cluster_indices, const orc_cluster_grid *grid, int n_threads);

/*
 * render_nerf_results_kernel + composite_nerf_results (renderer_kernel.cu:294-327, rt_core.cuh:334-416),
 * offscreen composite.  sample_values [n][value_stride] float, z_vals [n], offsets [h*w] inclusive
 * prefix sums of the per-ray sample counts (torch::cumsum, cuda_renderer.cpp:116).
 */
int orc_render_nerf_results(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                            const float *sample_values, int32_t value_stride, const float *z_vals,
                            const int64_t *offsets, float *rgba, uint8_t *rgba8, int n_threads);

/*
 * Refinement kernels (src/cuda/renderer_kernel.cu:63-213), serial restatements.  NOTE: these three live
 * in renderer_kernel.cu, which as a whole cannot be built for gfx950 (surf2Dread / surf2Dwrite); oracle/Makefile.ref
 * cuts the block with these three kernels (renderer_kernel.cu:63-213, no surface calls) out of the reference file verbatim
 * and builds it into oracle/_ref, and tests/golden/ref_refine_kernels.npz holds what they produced on gfx950: these
 * restatements reproduce it bit for bit (tests/test_goldens.py).
 * child/parent are edited in place; samples holds the caller's uniform [0,1) numbers on entry.
 */
int orc_add_children_and_generate_samples(int32_t *child, int32_t *parent, const float offset[3], const float scale[3],
                                          int32_t capacity, const orc_options *opt, const int32_t *parent_nodes,
                                          int32_t num_parents, float *samples, int32_t samples_dim,
                                          int16_t *cluster_indices, int32_t *visited, const orc_cluster_grid *grid);
int orc_generate_samples(const int32_t *parent, const float offset[3], const float scale[3], const orc_options *opt,
                         const int32_t *nodes, int32_t num_items, float *samples, int32_t samples_dim,
                         int16_t *cluster_indices, const orc_cluster_grid *grid);
int orc_adjust_parents_and_children(int32_t *child, int32_t *parent, int32_t capacity, int32_t first_shift_index,
                                    const uint8_t *to_delete, const int32_t *index_shifts);

/*
 * The build's own per-sample MLP (mega-nerf-viewer_amd/csrc/mnv_mlp.hip; stands in for query_submodules,
 * cuda_renderer.cpp:165-203).  PARITY UNPINNED: the reference's networks are TorchScript artefacts outside its
 * repository, so this restates the build's definition, not the reference: triangle-wave encoding (exact IEEE
 * arithmetic), binary16 weights and activations, fp32 accumulation in ascending input order, ReLU hidden layers,
 * linear fp32 output.  Same descriptor and parameter order as mnv_mlp_desc / mnv_mlp_create (include/mnv.h).
 */
typedef struct {
    int32_t n_clusters, pos_octaves, dir_octaves, need_viewdir, n_embeddings, embedding_dim;
    int32_t hidden_width, hidden_layers, out_dim;
    float center[3], inv_extent[3];
} orc_mlp_desc;
int orc_mlp_forward(const orc_mlp_desc *desc, const uint16_t *params, const int16_t *cluster_indices, const float *samples,
                    int32_t samples_stride, int64_t n, float *results, int32_t result_stride);

int orc_num_threads(void);

/* CPU-baseline fairness: a copy of data / child whose pages are first touched in parallel (spread over the NUMA nodes of a many-socket
 * host); the copy renders exactly like the original.  orc_tree_free_copy releases the two arrays of such a copy. */
int orc_tree_copy_first_touch(const orc_tree *src, orc_tree *dst, int n_threads);
void orc_tree_free_copy(orc_tree *t);

/* Analysis hook (no parity test depends on it): while `hist_2x32` is non-NULL every march step of orc_render_voxels adds 1 to
 * hist[dense][leaf depth] (dense = sigma > sigma_thresh).  tools/step_depths.py uses it to say where a workload's steps land. */
void orc_set_depth_histogram(uint64_t *hist_2x32);

#ifdef __cplusplus
}
#endif
#endif
