// mnv_guided.hip -- the two pure kernels of the guided-sampling path (BASELINE config 5), on the
// reference array layout:
//   mnv_get_samples_from_voxels  <- get_samples_from_voxels_kernel  src/cuda/renderer_kernel.cu:329-363
//                                   + device::get_samples_trace_ray  include/cuda/rt_core.cuh:418-576
//   mnv_render_nerf_results      <- render_nerf_results_kernel      src/cuda/renderer_kernel.cu:294-327
//                                   + device::composite_nerf_results include/cuda/rt_core.cuh:334-416
// The per-sample MLP that sits between them in the reference (cuda_renderer.cpp:165-203) is an
// external TorchScript artefact and is not part of this repository (DESIGN.md "Out of scope").
// One lane per ray, 8x8-pixel tile per wavefront; same arithmetic contract as the march kernels.
#include <cstring>

#include "mnv_device.h"
#include "mnv_internal.h"

#pragma clang fp contract(off)

namespace mnv {

struct SampleParams {
    MarchParams M;
    int32_t max_guided_samples, samples_dim, need_viewdir, appearance_embedding;
    int16_t *num_samples;
    float *samples;
    int16_t *cluster_indices;
    int32_t grid_dim[2];
    float min_position[3], range[3];
};

__global__ __launch_bounds__(256) void get_samples_kernel(const SampleParams S) {
    __shared__ uint64_t s_exp[32];
    load_exp_table(s_exp);
    const MarchParams &P = S.M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int by = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (bx >= P.tw || by >= P.th) return;
    const int64_t p = (int64_t)by * P.tw + bx;

    RaySetup<1> r;  // no SH basis needed here
    setup_ray<0>(P, P.cam, P.x0 + bx, P.y0 + by, r, frame_tmax(P, p));
    // world-space ray for the emitted sample positions: true_dir / true_cen / vdir (renderer_kernel.cu:348-351)
    const float *m = P.cam.c2w;
    float true_dir[3], vdir[3];
    world_ray_dirs(P, P.cam, P.x0 + bx, P.y0 + by, true_dir, vdir);

    float sp_prio = (float)(P.max_depth + 1), sp_chunk = -1.f, sp_child = -1.f;
    float sa_prio = (float)(P.max_sample_count + 1), sa_chunk = -1.f, sa_child = -1.f;
    int ns = S.num_samples[p];
    if (r.in_bbox) {
        float T = 1.f, t = r.tmin, max_weight = -1.f, max_sample_weight = -1.f;
        while (t < r.tmax) {
            float pos[3];
            for (int i = 0; i < 3; ++i) {
                pos[i] = P.cam.cen[i] + t * r.dir[i];
                pos[i] = fmaxf(fminf(pos[i], 1.f - 1e-6f), 0.f);
            }
            int32_t chunk = 0, cidx;
            int depth = 1;
            for (;;) {
                // rt_core.cuh:132-134 marks with atomicCAS(&visited[chunk], 0, 1); the mark only ever goes 0 -> 1, so a
                // load and a conditional plain store leave the same array -- without every ray serialising on the
                // root's word (measured: 974 ms -> about the unmarked frame time on the cfg2 frame)
                if (P.track_visit && __hip_atomic_load(&P.visited[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) P.visited[chunk] = 1;
                cidx = 0;
                for (int i = 0; i < 3; ++i) {
                    pos[i] *= 2.f;
                    const float f = floorf(pos[i]);
                    cidx = cidx * 2 + (int)f;
                    pos[i] -= f;
                }
                const int32_t skip = P.child[(int64_t)chunk * 8 + cidx];
                if (skip == 0) break;
                ++depth;
                chunk += skip;
            }
            float tu = 1e4f;
            for (int i = 0; i < 3; ++i) {
                const float t1 = -pos[i] * r.invdir[i];
                const float t2 = t1 + r.invdir[i];
                tu = fminf(tu, fmaxf(t1, t2));
            }
            const float cube = __uint_as_float((uint32_t)(127 + depth) << 23);
            const float delta_t = tu / cube + P.step_size;
            const float sigma = half_bits_to_float(P.data[((int64_t)chunk * 8 + cidx) * P.data_dim + P.data_dim - 1]);
            const int16_t sc = P.sample_counts ? P.sample_counts[(int64_t)chunk * 8 + cidx] : (int16_t)0;
            if (sigma > P.sigma_thresh) {
                const float att = exact_expf(-delta_t * r.delta_scale * sigma, s_exp);
                const float weight = T * (1.f - att);
                if (weight > max_weight && depth < P.max_depth) {
                    sp_chunk = (float)chunk; sp_child = (float)cidx; sp_prio = (float)depth;
                    max_weight = weight;
                }
                if (P.sample_counts && weight > max_sample_weight && sc < P.max_sample_count) {
                    sa_chunk = (float)chunk; sa_child = (float)cidx; sa_prio = (float)sc;
                    max_sample_weight = weight;
                }
                if (ns < S.max_guided_samples) {  // rt_core.cuh:508-549
                    float *row = S.samples + ((int64_t)p * S.max_guided_samples + ns) * S.samples_dim;
                    float tz[3];
                    for (int i = 0; i < 3; ++i) tz[i] = t * r.dir[i] / P.scale[i];
                    const float z = sqrtf(tz[0] * tz[0] + tz[1] * tz[1] + tz[2] * tz[2]);
                    const float wx = m[9] + true_dir[0] * z, wy = m[10] + true_dir[1] * z, wz = m[11] + true_dir[2] * z;
                    row[0] = z;
                    row[1] = wx;
                    row[2] = wy;
                    row[3] = wz;
                    if (S.need_viewdir) {
                        row[4] = vdir[0];
                        row[5] = vdir[1];
                        row[6] = vdir[2];
                        if (S.appearance_embedding != -1) row[7] = (float)S.appearance_embedding;
                    } else if (S.appearance_embedding != -1) {
                        row[4] = (float)S.appearance_embedding;
                    }
                    const int g1 = (int)fmaxf(fminf((wy - S.min_position[1]) / S.range[1] * (float)S.grid_dim[0], (float)S.grid_dim[0] - 1.0f), 0.0f);
                    const int g2 = (int)fmaxf(fminf((wz - S.min_position[2]) / S.range[2] * (float)S.grid_dim[1], (float)S.grid_dim[1] - 1.0f), 0.0f);
                    S.cluster_indices[(int64_t)p * S.max_guided_samples + ns] = (int16_t)(g1 * S.grid_dim[1] + g2);
                    ns += 1;
                }
                T *= att;
                if (T < P.stop_thresh) break;
            } else {
                if (max_weight == -1.f && depth < P.max_depth) {
                    sp_chunk = (float)chunk; sp_child = (float)cidx; sp_prio = (float)depth;
                }
                if (P.sample_counts && max_sample_weight == -1.f && sc < P.max_sample_count) {
                    sa_chunk = (float)chunk; sa_child = (float)cidx; sa_prio = (float)sc;
                }
            }
            t += delta_t;
        }
    }
    S.num_samples[p] = (int16_t)ns;
    if (P.split_track) {
        P.split_track[p * 3 + 0] = sp_prio;
        P.split_track[p * 3 + 1] = sp_chunk;
        P.split_track[p * 3 + 2] = sp_child;
    }
    if (P.sample_track) {
        P.sample_track[p * 3 + 0] = sa_prio;
        P.sample_track[p * 3 + 1] = sa_chunk;
        P.sample_track[p * 3 + 2] = sa_child;
    }
}

struct CompositeParams {
    FrameParams P;
    const float *sample_values;
    const float *z_vals;
    const int64_t *offsets;
    int32_t value_stride, basis_dim, format;
};

template <int BASIS>
__global__ __launch_bounds__(256) void composite_nerf_kernel(const CompositeParams Cp) {
    __shared__ uint64_t s_exp[32];
    load_exp_table(s_exp);
    const FrameParams &P = Cp.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int by = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (bx >= P.tw || by >= P.th) return;
    const int64_t p = (int64_t)by * P.tw + bx;
    constexpr int NB = BASIS > 0 ? BASIS : 1;
    RaySetup<NB> r;  // only r.basis (SH basis of the rotated view direction) is used
    setup_ray<(BASIS > 0 ? BASIS : 0)>(P, P.cam, P.x0 + bx, P.y0 + by, r);
    if constexpr (BASIS == 0) r.basis[0] = (0 < P.basis_min || 0 > P.basis_max) ? 0.f : (float)0.28209479177387814;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    const int64_t start = p == 0 ? 0 : Cp.offsets[p - 1], end = Cp.offsets[p];
    if (start != end) {
        float ti = 1.f, wc = 0.f, weight;
        for (int64_t i = start; i < end; ++i) {
            const float *sv = Cp.sample_values + i * Cp.value_stride;
            if (i < end - 1) {
                const float delta_i = Cp.z_vals[i + 1] - Cp.z_vals[i];
                wc = exact_expf(-sv[3] * delta_i, s_exp);
                weight = ti * (1.0f - wc);
            } else {
                weight = ti;
            }
            if (P.render_depth) {
                o0 += weight * ti;
            } else if constexpr (BASIS >= 0) {
                auto coef = [&](int k) { return sv[k]; };
                const int stride = BASIS > 0 ? BASIS : Cp.basis_dim;
                const float c0 = sh_channel<BASIS>(r.basis, coef, 0);
                const float c1 = sh_channel<BASIS>(r.basis, coef, stride);
                const float c2 = sh_channel<BASIS>(r.basis, coef, 2 * stride);
                o0 += weight / (1.f + exact_expf(-c0, s_exp));
                o1 += weight / (1.f + exact_expf(-c1, s_exp));
                o2 += weight / (1.f + exact_expf(-c2, s_exp));
            } else {
                o0 += weight * sv[0];
                o1 += weight * sv[1];
                o2 += weight * sv[2];
            }
            ti *= wc;
        }
        if (P.render_depth) o0 = o1 = o2 = fminf(o0 * 0.3f, 1.0f);
    }
    composite_and_write(P, p, o0, o1, o2, 1.0f);  // out[3] = 1 (renderer_kernel.cu:316): no background shows through
}

// ---- refinement kernels (renderer_kernel.cu:63-213) ---------------------------------------------

struct RefineParams {
    int32_t *child;
    int32_t *parent;
    float offset[3], scale[3];
    int32_t capacity;
    int32_t samples_per_corner, samples_dim, need_viewdir, appearance_embedding;
    float *samples;
    int16_t *cluster_indices;
    int32_t grid_dim[2];
    float min_position[3], range[3];
};

// generate_samples_inner, renderer_kernel.cu:88-168.  `top_parent` is the parent entry of
// `abs_chunk` when that chunk is being created by this very launch (the reference reads it back from
// tree.parent while another thread of the launch writes it); -1 otherwise.
__device__ __forceinline__ void generate_samples_inner(const RefineParams &R, int64_t idx, int32_t abs_chunk, int32_t child_idx,
                                                       int32_t top_parent) {
    int32_t cur = abs_chunk * 8 + child_idx;
    int depth = 0;
    float corners[3] = {0.f, 0.f, 0.f};
    bool first = true;
    for (;;) {
        const int32_t cz = cur % 2;
        cur /= 2;
        const int32_t cy = cur % 2;
        cur /= 2;
        const int32_t cx = cur % 2;
        cur /= 2;
        corners[0] = (corners[0] + (float)cx) / 2.f;
        corners[1] = (corners[1] + (float)cy) / 2.f;
        corners[2] = (corners[2] + (float)cz) / 2.f;
        if (cur == 0) break;
        cur = (first && top_parent >= 0) ? top_parent : R.parent[cur];
        first = false;
        depth += 1;
    }
    const float length_local = __uint_as_float((uint32_t)(127 - depth - 1) << 23);  // pow(N, -depth - 1)
    const int spc = R.samples_per_corner, dim = R.samples_dim;
    float *row0 = R.samples + idx * spc * dim;
    for (int i = 0; i < 3; ++i) {
        corners[i] -= R.offset[i];
        corners[i] /= R.scale[i];
        const float mul = length_local / R.scale[i];
        for (int j = 0; j < spc; ++j) {
            float v = row0[j * dim + i];
            v *= mul;
            v += corners[i];
            row0[j * dim + i] = v;
        }
    }
    if (R.need_viewdir) {
        for (int j = 0; j < spc; ++j) {
            row0[j * dim + 3] = 1.f;
            row0[j * dim + 4] = 0.f;
            row0[j * dim + 5] = 0.f;
            if (R.appearance_embedding != -1) row0[j * dim + 6] = (float)R.appearance_embedding;
        }
    } else if (R.appearance_embedding != -1) {
        for (int j = 0; j < spc; ++j) row0[j * dim + 3] = (float)R.appearance_embedding;
    }
    for (int j = 0; j < spc; ++j) {
        const int g1 = (int)fmaxf(fminf((row0[j * dim + 1] - R.min_position[1]) / R.range[1] * (float)R.grid_dim[0], (float)R.grid_dim[0] - 1.0f), 0.0f);
        const int g2 = (int)fmaxf(fminf((row0[j * dim + 2] - R.min_position[2]) / R.range[2] * (float)R.grid_dim[1], (float)R.grid_dim[1] - 1.0f), 0.0f);
        R.cluster_indices[idx * spc + j] = (int16_t)(g1 * R.grid_dim[1] + g2);
    }
}

__global__ void add_children_kernel(const RefineParams R, const int32_t *parent_nodes, int32_t *visited, int32_t num_parents) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (int64_t)num_parents * 8) return;
    const int32_t rel = (int32_t)(tid / 8), child_idx = (int32_t)(tid % 8);
    const int32_t abs_chunk = R.capacity + rel;
    const int32_t pc = parent_nodes[rel * 2], pj = parent_nodes[rel * 2 + 1];
    if (child_idx == 0) {
        R.child[(int64_t)pc * 8 + pj] = abs_chunk - pc;
        R.parent[abs_chunk] = pc * 8 + pj;
        visited[abs_chunk] = visited[pc];
    }
    R.child[(int64_t)abs_chunk * 8 + child_idx] = 0;
    generate_samples_inner(R, tid, abs_chunk, child_idx, pc * 8 + pj);
}

__global__ void generate_samples_kernel(const RefineParams R, const int32_t *nodes, int32_t num_items) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= num_items) return;
    generate_samples_inner(R, tid, nodes[tid * 2], nodes[tid * 2 + 1], -1);
}

__global__ void adjust_parents_kernel(int32_t *child, int32_t *parent, int32_t capacity, int32_t first_shift_index,
                                      const uint8_t *to_delete, const int32_t *index_shifts) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= capacity - first_shift_index) return;
    const int32_t chunk = (int32_t)tid + first_shift_index;
    // The root has no parent slot to fix.  The reference's host code does pass first_shift_index 0
    // (cuda_renderer.cpp:350: argmin of a cumsum); its thread for chunk 0 then adds 0 to child[0][0]
    // (svox stores parent 0 for the root), i.e. changes nothing -- skipped here, which also removes the
    // unsynchronised read-modify-write on that word.
    if (chunk == 0) return;
    const int32_t par = parent[chunk];
    const int32_t pc = par / 8, pj = par % 8;
    if (to_delete[chunk]) {
        child[(int64_t)pc * 8 + pj] = 0;
    } else {
        const int32_t parent_shift = index_shifts[pc], child_shift = index_shifts[chunk];
        child[(int64_t)pc * 8 + pj] += (parent_shift - child_shift);
        parent[chunk] = par - index_shifts[pc] * 8;
    }
}

static int fill_refine(RefineParams &R, const mnv_tree_edit *t, const mnv_render_options *opt, float *samples, int32_t samples_dim,
                       int16_t *cluster_indices, const mnv_cluster_grid *grid) {
    if (!t || !opt || !samples || !cluster_indices || !grid || !t->child || !t->parent) return set_error(MNV_E_INVALID, "null argument");
    if (t->N != 2) return set_error(MNV_E_UNSUPPORTED, "only N == 2 octrees are supported");
    const int need = 3 + (opt->need_viewdir ? 3 : 0) + (opt->appearance_embedding != -1 ? 1 : 0);
    if (samples_dim < need) return set_error(MNV_E_INVALID, "samples_dim too small for the requested columns");
    R.child = t->child;
    R.parent = t->parent;
    std::memcpy(R.offset, t->offset, sizeof(R.offset));
    std::memcpy(R.scale, t->scale, sizeof(R.scale));
    R.capacity = t->capacity;
    R.samples_per_corner = opt->samples_per_corner;
    R.samples_dim = samples_dim;
    R.need_viewdir = opt->need_viewdir ? 1 : 0;
    R.appearance_embedding = opt->appearance_embedding;
    R.samples = samples;
    R.cluster_indices = cluster_indices;
    std::memcpy(R.grid_dim, grid->grid_dim, sizeof(R.grid_dim));
    std::memcpy(R.min_position, grid->min_position, sizeof(R.min_position));
    std::memcpy(R.range, grid->range, sizeof(R.range));
    return MNV_OK;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

int mnv_get_samples_from_voxels(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                int16_t *num_samples, float *samples, int32_t samples_dim, int16_t *cluster_indices,
                                const mnv_cluster_grid *grid, void *hip_stream) {
    return mnv_get_samples_from_voxels_ex(tree, cam, opt, tile, nullptr, split_track, sample_track, visited, track_visit, num_samples, samples, samples_dim,
                                          cluster_indices, grid, hip_stream);
}

int mnv_get_samples_from_voxels_ex(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                   const mnv_frame_inputs *inputs, float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                   int16_t *num_samples, float *samples, int32_t samples_dim, int16_t *cluster_indices,
                                   const mnv_cluster_grid *grid, void *hip_stream) {
    if (!num_samples || !samples || !cluster_indices || !grid || !opt) return set_error(MNV_E_INVALID, "null argument");
    if (!tree || tree->N <= 0) return set_error(MNV_E_INVALID, "get_samples needs a non-empty tree");
    const int need = 4 + (opt->need_viewdir ? 3 : 0) + (opt->appearance_embedding != -1 ? 1 : 0);
    if (samples_dim < need) return set_error(MNV_E_INVALID, "samples_dim too small for the requested columns");
    if (track_visit && !visited) return set_error(MNV_E_INVALID, "track_visit set but visited is null");
    SampleParams S;
    std::memset(static_cast<void *>(&S), 0, sizeof(S));
    int rc = fill_params(S.M, cam, opt, tile);
    if (rc) return rc;
    rc = fill_tree_params(S.M, tree);
    if (rc) return rc;
    if (tree->N > 0 && tree->N != 2)
        return set_error(MNV_E_UNSUPPORTED, "the sample march supports N == 2 trees (generate_samples is written for N == 2 in the reference as well, renderer_kernel.cu:88-168)");
    S.M.max_depth = opt->max_depth;
    S.M.max_sample_count = opt->max_sample_count;
    if (inputs) S.M.tmax_px = inputs->tmax_px;  // offscreen == false: the ray limit of every pixel (renderer_kernel.cu:354-357)
    S.M.split_track = split_track;
    S.M.sample_track = sample_track;
    S.M.visited = visited;
    S.M.track_visit = track_visit ? 1 : 0;
    S.max_guided_samples = opt->max_guided_samples;
    S.samples_dim = samples_dim;
    S.need_viewdir = opt->need_viewdir ? 1 : 0;
    S.appearance_embedding = opt->appearance_embedding;
    S.num_samples = num_samples;
    S.samples = samples;
    S.cluster_indices = cluster_indices;
    std::memcpy(S.grid_dim, grid->grid_dim, sizeof(S.grid_dim));
    std::memcpy(S.min_position, grid->min_position, sizeof(S.min_position));
    std::memcpy(S.range, grid->range, sizeof(S.range));
    if (tile.w <= 0 || tile.h <= 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(get_samples_kernel, dim3((tile.w + 15) / 16, (tile.h + 15) / 16), dim3(256), 0, stream, S);
    return check_hip(hipGetLastError(), "get_samples_kernel");
}

int mnv_render_nerf_results(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                            const float *sample_values, int32_t value_stride, const float *z_vals, const int64_t *offsets,
                            float *rgba_out, uint8_t *rgba8_out, void *hip_stream) {
    if (!tree || !offsets || !opt) return set_error(MNV_E_INVALID, "null argument");
    if (value_stride < 4) return set_error(MNV_E_INVALID, "value_stride < 4");
    CompositeParams Cp;
    std::memset(static_cast<void *>(&Cp), 0, sizeof(Cp));
    int rc = fill_params(Cp.P, cam, opt, tile);
    if (rc) return rc;
    std::memcpy(Cp.P.offset, tree->offset, sizeof(Cp.P.offset));
    std::memcpy(Cp.P.scale, tree->scale, sizeof(Cp.P.scale));
    fill_origin(Cp.P.cam, Cp.P.offset, Cp.P.scale);
    Cp.P.rgba = rgba_out;
    Cp.P.rgba8 = rgba8_out;
    Cp.sample_values = sample_values;
    Cp.z_vals = z_vals;
    Cp.offsets = offsets;
    Cp.value_stride = value_stride;
    Cp.basis_dim = tree->basis_dim;
    Cp.format = tree->format;
    if (tile.w <= 0 || tile.h <= 0) return MNV_OK;
    const int b = tree->basis_dim;  // the colour branch keys on basis_dim >= 0 (rt_core.cuh:374)
    if (b >= 0 && value_stride < 3 * b) return set_error(MNV_E_INVALID, "value_stride too small for 3 * basis_dim");
    hipStream_t stream = (hipStream_t)hip_stream;
    dim3 grid((tile.w + 15) / 16, (tile.h + 15) / 16), block(256);
    const bool sh = tree->format == MNV_FORMAT_SH;
#define MNV_LAUNCH(B) hipLaunchKernelGGL(composite_nerf_kernel<B>, grid, block, 0, stream, Cp)
    if (b < 0) MNV_LAUNCH(-1);
    else if (sh && b == 4) MNV_LAUNCH(4);
    else if (sh && b == 9) MNV_LAUNCH(9);
    else if (sh && b == 16) MNV_LAUNCH(16);
    else if (sh && b == 25) MNV_LAUNCH(25);
    else MNV_LAUNCH(0);
#undef MNV_LAUNCH
    return check_hip(hipGetLastError(), "composite_nerf_kernel");
}

int mnv_add_children_and_generate_samples(const mnv_tree_edit *tree, const mnv_render_options *opt, const int32_t *parent_nodes,
                                          int32_t num_parents, float *samples, int32_t samples_dim, int16_t *cluster_indices,
                                          int32_t *visited, const mnv_cluster_grid *grid, void *hip_stream) {
    RefineParams R;
    std::memset(&R, 0, sizeof(R));
    const int rc = fill_refine(R, tree, opt, samples, samples_dim, cluster_indices, grid);
    if (rc) return rc;
    if (!parent_nodes || !visited) return set_error(MNV_E_INVALID, "null argument");
    if (num_parents <= 0) return MNV_OK;
    const int64_t n = (int64_t)num_parents * 8;
    hipLaunchKernelGGL(add_children_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, R, parent_nodes, visited, num_parents);
    return check_hip(hipGetLastError(), "add_children_kernel");
}

int mnv_generate_samples(const mnv_tree_edit *tree, const mnv_render_options *opt, const int32_t *nodes, int32_t num_items,
                         float *samples, int32_t samples_dim, int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream) {
    RefineParams R;
    std::memset(&R, 0, sizeof(R));
    const int rc = fill_refine(R, tree, opt, samples, samples_dim, cluster_indices, grid);
    if (rc) return rc;
    if (!nodes) return set_error(MNV_E_INVALID, "null argument");
    if (num_items <= 0) return MNV_OK;
    hipLaunchKernelGGL(generate_samples_kernel, dim3((unsigned)((num_items + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, R, nodes, num_items);
    return check_hip(hipGetLastError(), "generate_samples_kernel");
}

int mnv_adjust_parents_and_children(const mnv_tree_edit *tree, int32_t first_shift_index, const uint8_t *to_delete,
                                    const int32_t *index_shifts, void *hip_stream) {
    if (!tree || !tree->child || !tree->parent || !to_delete || !index_shifts) return set_error(MNV_E_INVALID, "null argument");
    if (tree->N != 2) return set_error(MNV_E_UNSUPPORTED, "only N == 2 octrees are supported");
    if (first_shift_index < 0) return set_error(MNV_E_INVALID, "first_shift_index must be >= 0");
    const int64_t n = (int64_t)tree->capacity - first_shift_index;
    if (n <= 0) return MNV_OK;
    hipLaunchKernelGGL(adjust_parents_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, tree->child, tree->parent,
                       tree->capacity, first_shift_index, to_delete, index_shifts);
    return check_hip(hipGetLastError(), "adjust_parents_kernel");
}

}  // extern "C"
