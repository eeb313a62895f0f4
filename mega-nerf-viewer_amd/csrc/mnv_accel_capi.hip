// mnv_accel_capi.hip -- launch planning for the kernels of the packed layout (ray queues, partition, launch slots), the tile assembly of
// rank 0, and the C-ABI entry points of include/mnv.h that render on an accel.  The kernels themselves: mnv_march_accel_kernel.h
// (instantiated by mnv_accel_march.hip), mnv_guided_fused*.h (mnv_accel_fused.hip).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "mnv_accel_launch.h"
#include "mnv_knobs.h"

namespace mnv {

// Per-launch parameters that live in device memory: zeroes the ray-queue heads of every frame and
// stores the camera blocks (handed over by value, so no host staging buffer or copy engine is involved).
constexpr int kStageCams = 32;
struct StageCams {
    CamBlock c[kStageCams];
};
__global__ void stage_launch_kernel(uint32_t *heads, int32_t head_words, CamBlock *dst, const StageCams cams, int32_t count) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < head_words) heads[i] = 0u;
    if (i < count * (int32_t)(sizeof(CamBlock) / 4))
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(cams.c)[i];
}

// Visit marks on the packed layout: the march marks the chunk that holds each leaf it steps through; the reference marks every
// chunk of every descent (query_single_from_root, rt_core.cuh:132-134), i.e. those chunks and all their ancestors.  One thread
// per marked chunk walks up the parent words until it meets a chunk that is marked already.
__global__ void close_visit_marks(int32_t *visited, const int32_t *parent, int32_t capacity) {
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= capacity || c == 0 || visited[c] == 0) return;
    int32_t p = parent[c] >> 3;
    while (p >= 0 && p < capacity) {
        if (__hip_atomic_load(&visited[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        visited[p] = 1;
        if (p == 0) break;
        p = parent[p] >> 3;
    }
}

// Rank 0 after the gather: macro tile m of frame f sits at gathered[m % world][f][m / world];
// one thread per pixel, the destination is written row-major (coalesced), the source is read in tile rows.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16 bytes: one float RGBA pixel or four RGBA8 pixels
template <typename PIXEL>
__global__ void assemble_tiles_kernel(const PIXEL *__restrict__ gathered, PIXEL *__restrict__ frames, int32_t width, int32_t height,
                                      int32_t tile_w, int32_t tile_h, int32_t macros_x, int32_t j_max, int32_t world, int32_t period, int32_t n_frames) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_frame = (int64_t)width * height;
    if (idx >= per_frame * n_frames) return;
    const int32_t f = (int32_t)(idx / per_frame);
    const int32_t p = (int32_t)(idx - (int64_t)f * per_frame);
    const int32_t y = p / width, x = p - y * width;
    const int32_t mx = x / tile_w, my = y / tile_h;
    uint32_t r, j;
    part_owner_of((uint32_t)(my * macros_x + mx), world, period, r, j);
    const int64_t src = ((((int64_t)r * n_frames + f) * j_max + j) * tile_h + (y - my * tile_h)) * tile_w + (x - mx * tile_w);
    // streamed once: keep these lines from displacing the march's lookup structures in L2
    __builtin_nontemporal_store(__builtin_nontemporal_load(&gathered[src]), &frames[idx]);
}

int32_t partition_local_tiles(mnv_rect tile, mnv_partition part) {
    if (tile.w <= 0 || tile.h <= 0 || !is_partitioned(part)) return is_partitioned(part) ? 0 : 1;
    const int64_t mx = (tile.w + part.tile_w - 1) / part.tile_w, my = (tile.h + part.tile_h - 1) / part.tile_h;
    return (int32_t)part_local_count(mx * my, part.rank, part.world, root_period_of(part));
}

int launch_accel(const mnv_accel *accel, const FrameParams &P, const CamBlock *cams, int n_frames, mnv_partition part,
                 const AccelTrack *track, hipStream_t stream) {
    if (P.tw <= 0 || P.th <= 0 || n_frames <= 0) return 0;
    AccelLaunch K;
    std::memset(static_cast<void *>(&K), 0, sizeof(K));
    K.P = P;
    if (track) {
        K.split_track = track->split_track;
        K.sample_track = track->sample_track;
        K.sample_counts = track->sample_counts;
        K.max_depth = track->max_depth;
        K.max_sample_count = track->max_sample_count;
        K.visited = track->visited;
        if (track->samples) {
            K.num_samples = track->num_samples;
            K.samples = track->samples;
            K.cluster_indices = track->cluster_indices;
            K.max_guided_samples = track->max_guided_samples;
            K.samples_dim = track->samples_dim;
            K.need_viewdir = track->need_viewdir;
            K.appearance_embedding = track->appearance_embedding;
            for (int i = 0; i < 2; ++i) K.grid_dim[i] = track->grid->grid_dim[i];
            for (int i = 0; i < 3; ++i) {
                K.min_position[i] = track->grid->min_position[i];
                K.range[i] = track->grid->range[i];
            }
        }
    }
    K.fast_colour = accel->colour_math.load(std::memory_order_relaxed) > 0 ? 1 : 0;  // mnv_accel_set_colour_math: the accel's own setting (two renderers of one process may differ)
    K.part_rank = part.rank;
    K.part_world = is_partitioned(part) ? part.world : 0;
    K.part_period = root_period_of(part);
    static const int env_wlog = knob_int(KNOB_TILE_WLOG, 3);
    K.tile_wlog = (env_wlog >= 0 && env_wlog <= 6) ? (uint32_t)env_wlog : 3u;
    if (!is_partitioned(part)) {
        const uint32_t tile_w = 1u << K.tile_wlog, tile_h = 64u >> K.tile_wlog;
        K.tiles_x = (uint32_t)((P.tw + tile_w - 1) / tile_w);
        const uint32_t tiles_y = (uint32_t)((P.th + tile_h - 1) / tile_h);
        K.n_tiles = K.tiles_x * tiles_y;
        // contiguous bands of tile rows per queue
        for (int q = 0; q <= kNumQueues; ++q) K.band_begin[q] = (uint32_t)(((uint64_t)tiles_y * q) / kNumQueues) * K.tiles_x;
    } else {
        const mnv_rect rect = {P.x0, P.y0, P.tw, P.th};
        const uint32_t local = (uint32_t)partition_local_tiles(rect, part);
        if (local == 0) return 0;
        K.macro_w = (uint32_t)part.tile_w;
        K.macro_h = (uint32_t)part.tile_h;
        K.macros_x = (uint32_t)((P.tw + part.tile_w - 1) / part.tile_w);
        K.micro_x = K.macro_w / 8;
        K.micro_per_macro = K.micro_x * (K.macro_h / 8);
        K.tiles_x = K.micro_x;
        K.n_tiles = local * K.micro_per_macro;
        // contiguous runs of micro tiles (in local macro-tile order) per queue
        for (int q = 0; q <= kNumQueues; ++q) K.band_begin[q] = (uint32_t)(((uint64_t)K.n_tiles * q) / kNumQueues);
    }
    static const int env_queues = knob_int(KNOB_QUEUES, kNumQueues);
    if (env_queues >= 1 && env_queues < kNumQueues) {
        // diagnostics: fewer, larger queues (queue q of the first env_queues covers 1/env_queues of the tiles)
        const uint32_t total = K.band_begin[kNumQueues];
        for (int q = 0; q <= kNumQueues; ++q)
            K.band_begin[q] = q >= env_queues ? total : (uint32_t)(((uint64_t)total * q) / env_queues);
    }
    // per-launch slot: zeroed queue heads + the camera blocks, written by stage_launch_kernel on the launch stream
    K.n_frames = (uint32_t)n_frames;
    // frames of a batch are j_max = ceil(macro tiles / world) local tiles apart on EVERY rank, so that the
    // per-rank buffers have one shape (what the gather needs) even when the tile count is ragged
    if (!is_partitioned(part)) {
        K.frame_stride_px = (uint32_t)P.tw * (uint32_t)P.th;
    } else {
        const uint32_t n_macro = K.macros_x * (uint32_t)((P.th + part.tile_h - 1) / part.tile_h);
        K.frame_stride_px = (uint32_t)part_j_max(n_macro, part.world, root_period_of(part)) * K.macro_w * K.macro_h;
    }
    // The per-launch slot bookkeeping is the handle's only mutable state on this path; launches from several host threads
    // (or one thread feeding several streams) serialise here from the slot's acquisition to the record of its event.
    mnv_accel *mut = const_cast<mnv_accel *>(accel);
    std::lock_guard<std::mutex> launch_lock(mut->launch_mutex);
    K.A = accel->view;  // (under the lock: a launching thread never sees a half-written view)
    const int b = (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim >= 0) ? accel->view.basis_dim : -1;
    // An inline cell word / a brick record calls a leaf whose sigma bits are 0 "not dense" without reading it: true for every
    // sigma_thresh >= 0.  A frame with a negative threshold walks the node words (every other lookup array says the same).
    if (P.sigma_thresh < 0.f) {
        K.A.grid2i = nullptr;
        K.A.recs = nullptr;
    }
    const uint32_t slot = mut->slot_counter.fetch_add(1) % kSlots;
    // a caller that runs more than kSlots launches ahead of the device waits here for the launch that last used the slot
    if (mut->slot_used[slot]) {
        hipError_t es = hipEventSynchronize(mut->slot_done[slot]);
        if (es != hipSuccess) return (int)es;
    }
    uint8_t *ds = accel->slots_dev + (size_t)slot * kSlotBytes;
    const size_t heads_bytes = (size_t)kNumQueues * 64;  // one head per queue; a queue spans the frames of the batch
    K.queue = reinterpret_cast<uint32_t *>(ds);
    CamBlock *dcams = reinterpret_cast<CamBlock *>(ds + heads_bytes);
    K.cams = dcams;
    for (int first = 0; first < n_frames; first += kStageCams) {
        StageCams sc;
        const int count = n_frames - first < kStageCams ? n_frames - first : kStageCams;
        std::memcpy(sc.c, cams + first, (size_t)count * sizeof(CamBlock));
        const int head_words = first == 0 ? (int)(heads_bytes / 4) : 0;
        const int n_threads = std::max(head_words, count * (int)(sizeof(CamBlock) / 4));
        hipLaunchKernelGGL(stage_launch_kernel, dim3((n_threads + 255) / 256), dim3(256), 0, stream, K.queue, head_words, dcams + first, sc, count);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;

    // diagnostics knobs; read once
    static const int env_level = knob_int(KNOB_LDS_LEVEL, -1);
    int lds_level = accel->view.grid_level < 3 ? accel->view.grid_level : 3;  // 2 KB; level 4 (16 KB) measured equal and costs occupancy
    if (env_level >= 1 && env_level <= accel->view.grid_level) lds_level = env_level;
    K.lds_level = lds_level;
    // sample emission and the depth image read no colour rows: one instantiation (BASIS 9) serves every row format
    const bool colourless = K.samples != nullptr || (P.render_depth && !K.split_track && !K.sample_track && !K.visited);
    const int nb_lds = colourless ? 9 : (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim > 0) ? accel->view.basis_dim : 1;
    const bool tracked = K.split_track || K.sample_track || K.samples || K.visited;  // (those frames keep one more value per ray in LDS: t_min)
    const size_t lds_bytes = 256 + (nb_lds >= 16 ? 1024 : 0) + (size_t)(nb_lds + 2 + (tracked ? 1 : 0)) * 256 * 4 + ((size_t)4 << (3 * lds_level));
    static const int env_bpc = knob_int(KNOB_BLOCKS_PER_CU, 0);
    static const int env_refill = knob_int(KNOB_REFILL_MIN, 0);
    static const int env_ablate = knob_int(KNOB_ABLATE, 0);
    K.ablate = env_ablate;
    static const bool env_stats = knob_set(KNOB_STATS);
    static const char *env_timeline = knob_str(KNOB_TIMELINE);
    static const char *env_footprint = knob_str(KNOB_FOOTPRINT);
    K.stats = (env_stats || env_ablate || env_timeline || env_footprint) ? accel->stats : nullptr;  // all four run on the diagnostics instantiation
    if (env_footprint && !(track && track->fused)) {
        // one bit per 128-byte line of every array the march loads from; the bits of all launches accumulate until mnv_accel_destroy counts them
        if (!mut->line_bits) {
            const uint64_t cells2 = accel->view.grid2_level > 0 ? (uint64_t)1 << (3 * accel->view.grid2_level) : 0, cells1 = (uint64_t)1 << (3 * accel->view.grid_level);
            const uint64_t nvox = (uint64_t)accel->reserved * 8;
            const uint64_t bytes[6] = {cells2 * 4, (uint64_t)accel->reserved * kRecWords * 4, nvox * 4, nvox * (uint64_t)accel->view.row_bytes, cells2 * 4, cells1 * 4};
            uint64_t at = 0;
            for (int i = 0; i < 6; ++i) {
                mut->line_base[i] = (uint32_t)at;
                at += (bytes[i] + 127) / 128 + 1;
            }
            mut->line_base[6] = (uint32_t)at;
            const size_t words = (size_t)(at + 31) / 32;
            if (hipMalloc((void **)&mut->line_bits, words * 4) != hipSuccess) return (int)hipErrorOutOfMemory;
            (void)hipMemsetAsync(mut->line_bits, 0, words * 4, stream);
        }
        K.line_bits = mut->line_bits;
        for (int i = 0; i < 6; ++i) K.line_base[i] = mut->line_base[i];
    }
    static const int env_shadow = knob_int(KNOB_SHADOW, 0);
    if (env_shadow & (16 | 32 | 64)) {
        // shadow loads (test-hook build + a -DMNV_SHADOW_MASK variant of the march): copies of the arrays at other addresses, made once
        const size_t nvox = (size_t)accel->reserved * 8;
        auto shadow = [&](void **dst, const void *src, size_t bytes) -> int {
            if (*dst || !src) return 0;
            hipError_t es = hipMalloc(dst, bytes);
            if (es == hipSuccess) es = hipMemcpyAsync(*dst, src, bytes, hipMemcpyDeviceToDevice, stream);
            return (int)es;
        };
        int es = 0;
        if (env_shadow & 16) es = shadow((void **)&mut->shadow_nodes, accel->nodes, nvox * 4);
        if (!es && (env_shadow & 32)) es = shadow((void **)&mut->shadow_rows, accel->rows, nvox * (size_t)accel->view.row_bytes);
        if (!es && (env_shadow & 64) && accel->recs) es = shadow((void **)&mut->shadow_nodes, accel->recs, (size_t)accel->reserved * kRecWords * 4);  // (bits 16 and 64 are not combined)
        if (es) return es;
        K.shadow_nodes = accel->shadow_nodes;
        K.shadow_rows = accel->shadow_rows;
        K.shadow_recs = reinterpret_cast<const uint2 *>(accel->shadow_nodes);
    }
    static const int env_stats_level = env_stats ? std::max(1, knob_int(KNOB_STATS, 1)) : 0;
    K.count_stats = env_stats ? env_stats_level : 0;
    K.refill_min = (env_refill > 0 && n_frames == 1) ? env_refill : 64;  // batches refill whole tiles (a grab must not straddle frames);  // sweep in DESIGN.md: 16 -> 0.606 ms, 32 -> 0.535, 48 -> 0.507, 56 -> 0.504, 64 -> 0.506
    int blocks_per_cu = lds_level >= 5 ? 1 : (lds_level == 4 ? 6 : 8);
    if ((K.split_track || K.sample_track || K.samples || K.visited) && blocks_per_cu > MNV_TRACK_WAVES) blocks_per_cu = MNV_TRACK_WAVES;
    if (env_bpc > 0) blocks_per_cu = env_bpc;
    int n_blocks = accel->num_cus * blocks_per_cu;
    const uint64_t n_waves_needed = (uint64_t)K.n_tiles * (uint64_t)n_frames;  // one initial 8x8 tile per wave
    if ((uint64_t)n_blocks * 4u > n_waves_needed) n_blocks = (int)((n_waves_needed + 3) / 4);
    if (n_blocks < 1) n_blocks = 1;

    if (env_timeline && K.stats) {
        // diagnostics: (re)allocate the record buffer of this launch; mnv_accel_destroy writes the last launch's records to the file
        const size_t tiles = (size_t)K.n_tiles * (size_t)n_frames, waves = (size_t)n_blocks * 4;
        const size_t bytes = (tiles * 4 + waves * 2) * 8;
        if (mut->timeline_bytes < bytes) {
            if (mut->timeline) (void)hipFree(mut->timeline);
            mut->timeline = nullptr;
            if (hipMalloc((void **)&mut->timeline, bytes) != hipSuccess) return (int)hipErrorOutOfMemory;
            mut->timeline_bytes = bytes;
        }
        (void)hipMemsetAsync(mut->timeline, 0, bytes, stream);
        mut->timeline_tiles = tiles;
        mut->timeline_waves = waves;
        mut->timeline_tiles_per_frame = K.n_tiles;
        K.timeline = mut->timeline;
        K.timeline_tiles = (uint32_t)tiles;
    }
    int rc = kUnsupportedBasis;
    if (track && track->fused) {
        rc = launch_fused(accel, K, *track->fused, b, lds_level, n_waves_needed, stream);
    } else if (K.A.grid2i && (b < 16 || colourless)) {
        // (SH16 / SH25 rows are evaluated by the cooperative pass, which has no brick variant)
        rc = launch_march_brick(K, b, colourless, n_blocks, lds_bytes, stream);
    } else {
        K.A.grid2i = nullptr;
        K.A.recs = nullptr;
        rc = launch_march(K, b, colourless, n_blocks, lds_bytes, stream);
    }
    if (rc == 0 && K.visited && track->parent) {
        hipLaunchKernelGGL(close_visit_marks, dim3((unsigned)((accel->view.capacity + 255) / 256)), dim3(256), 0, stream, K.visited, track->parent,
                           accel->view.capacity);
        rc = (int)hipGetLastError();
    }
    if (rc == 0 && track && track->fused)  // the fault word's pinned mirror follows the frame on its stream (guided_fused, mnv_accel_fused_faults)
        rc = (int)hipMemcpyAsync(mut->fault_host, mut->fault_dev, 4, hipMemcpyDeviceToHost, stream);
    if (rc == 0) {
        rc = (int)hipEventRecord(mut->slot_done[slot], stream);
        mut->slot_used[slot] = true;
    }
    return rc;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

int32_t mnv_partition_local_tiles(mnv_rect tile, mnv_partition part) { return partition_local_tiles(tile, part); }

int mnv_accel_set_colour_math(mnv_accel *accel, int mode) {
    if (!accel) return set_error(MNV_E_INVALID, "accel is null");
    accel->colour_math.store(mode > 0 ? 1 : 0, std::memory_order_relaxed);
    return MNV_OK;
}

int mnv_accel_set_fused_kernel(mnv_accel *accel, int version) {
    if (!accel) return set_error(MNV_E_INVALID, "accel is null");
    accel->fused_kernel.store(version == 1 || version == 2 ? version : 0, std::memory_order_relaxed);
    return MNV_OK;
}

int mnv_accel_set_fused_diag(mnv_accel *accel, unsigned long long *words32) {
    if (!accel) return set_error(MNV_E_INVALID, "accel is null");
    accel->fused_diag.store(words32, std::memory_order_relaxed);
    return MNV_OK;
}

int mnv_accel_fused_faults(const mnv_accel *accel, uint32_t *count_out) {
    if (!accel || !count_out) return set_error(MNV_E_INVALID, "null argument");
    uint32_t v = 0;
    const int rc = check_hip(hipMemcpy(&v, accel->fault_dev, 4, hipMemcpyDeviceToHost), "read fault word");  // waits for the device
    if (rc) return rc;
    *count_out = v;
    return MNV_OK;
}

int mnv_assemble_tiles(const void *gathered, void *frames, int32_t width, int32_t height, mnv_partition part, int32_t n_frames,
                       int32_t bytes_per_pixel, void *hip_stream) {
    if (!gathered || !frames || width < 1 || height < 1 || n_frames < 1 || part.world < 1 || part.tile_w < 8 || part.tile_h < 8 ||
        part.tile_w % 8 || part.tile_h % 8)
        return set_error(MNV_E_INVALID, "invalid tile-assembly arguments");
    if (bytes_per_pixel != 4 && bytes_per_pixel != 16) return set_error(MNV_E_UNSUPPORTED, "pixels are RGBA8 (4 bytes) or float RGBA (16 bytes)");
    const int32_t macros_x = (width + part.tile_w - 1) / part.tile_w, macros_y = (height + part.tile_h - 1) / part.tile_h;
    if (part.root_period < 0) return set_error(MNV_E_INVALID, "root_period must be 0 or >= 2");
    const int32_t period = root_period_of(part);
    const int32_t j_max = (int32_t)part_j_max((int64_t)macros_x * macros_y, part.world, period);
    hipStream_t stream = (hipStream_t)hip_stream;
    static const bool env_narrow = knob_set(KNOB_ASSEMBLE_NARROW);  // diagnostics: one RGBA8 pixel per thread
    if (bytes_per_pixel == 4 && width % 4 == 0 && !env_narrow) {
        // RGBA8: tile rows and frame rows are contiguous runs of pixels and tile_w is a multiple of 8, so the same index arithmetic
        // holds in units of four pixels: 16 bytes per thread instead of 4 (rank 0 runs this beside its march on the few compute
        // units the march leaves free)
        const int64_t n4 = (int64_t)(width / 4) * height * n_frames;
        hipLaunchKernelGGL(assemble_tiles_kernel<u32x4>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, static_cast<const u32x4 *>(gathered),
                           static_cast<u32x4 *>(frames), width / 4, height, part.tile_w / 4, part.tile_h, macros_x, j_max, part.world, period, n_frames);
        return check_hip(hipGetLastError(), "assemble_tiles_kernel");
    }
    const int64_t n = (int64_t)width * height * n_frames;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (bytes_per_pixel == 4)
        hipLaunchKernelGGL(assemble_tiles_kernel<uint32_t>, grid, block, 0, stream, static_cast<const uint32_t *>(gathered), static_cast<uint32_t *>(frames),
                           width, height, part.tile_w, part.tile_h, macros_x, j_max, part.world, period, n_frames);
    else
        hipLaunchKernelGGL(assemble_tiles_kernel<u32x4>, grid, block, 0, stream, static_cast<const u32x4 *>(gathered), static_cast<u32x4 *>(frames), width,
                           height, part.tile_w, part.tile_h, macros_x, j_max, part.world, period, n_frames);
    return check_hip(hipGetLastError(), "assemble_tiles_kernel");
}

int mnv_render_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                            mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return mnv_render_voxels_accel_batch(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, hip_stream);
}

int mnv_render_voxels_accel_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                 mnv_rect tile, mnv_partition part, float *rgba_out, uint8_t *rgba8_out,
                                 void *hip_stream) {
    return mnv_render_voxels_accel_batch(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, hip_stream);
}

// inputs: the per-pixel arrays of the reference's offscreen == false call shape (NULL: offscreen).  xform: offset[3] + scale[3] of the
// caller's tree view when they may differ from the ones the accel was built with (the tree cache of mnv_render_voxels), else NULL.
static int render_accel(const mnv_accel *accel, const mnv_camera *cams, int32_t n_cams, const mnv_render_options *opt,
                        mnv_rect tile, mnv_partition part, float *rgba_out, uint8_t *rgba8_out, const AccelTrack *track,
                        void *hip_stream, const mnv_frame_inputs *inputs = nullptr, const float *xform = nullptr) {
    if (!accel) return set_error(MNV_E_INVALID, "accel is null");
    if (!cams || n_cams < 1 || n_cams > MNV_MAX_BATCH) return set_error(MNV_E_INVALID, "need 1 .. MNV_MAX_BATCH cameras");
    if (is_partitioned(part) && (part.rank < 0 || part.rank >= part.world || part.tile_w < 8 || part.tile_h < 8 ||
                           part.tile_w % 8 || part.tile_h % 8 || part.root_period < 0 || part.root_period == 1))
        return set_error(MNV_E_INVALID, "partition needs 0 <= rank < world, macro tiles that are multiples of 8 pixels and root_period 0 or >= 2");
    for (int i = 1; i < n_cams; ++i)
        if (cams[i].width != cams[0].width || cams[i].height != cams[0].height)
            return set_error(MNV_E_INVALID, "all cameras of a batch must have the same image size");
    FrameParams P;
    std::memset(&P, 0, sizeof(P));
    int rc = fill_params(P, &cams[0], opt, tile);
    if (rc) return rc;
    // pixel indices of a launch are 32 bits wide (frame f starts at f * pixels per frame)
    if ((uint64_t)(tile.w > 0 ? tile.w : 0) * (uint64_t)(tile.h > 0 ? tile.h : 0) * (uint64_t)n_cams > 0xffffffffull)
        return set_error(MNV_E_UNSUPPORTED, "more than 2^32 pixels in one launch: render fewer frames per call");
    std::memcpy(P.offset, xform ? xform : accel->view.offset, sizeof(P.offset));
    std::memcpy(P.scale, xform ? xform + 3 : accel->view.scale, sizeof(P.scale));
    P.rgba = rgba_out;
    P.rgba8 = rgba8_out;
    if (inputs) {
        P.tmax_px = inputs->tmax_px;
        P.rgba8_init = inputs->rgba8_init;
    }
    CamBlock blocks[MNV_MAX_BATCH];
    for (int i = 0; i < n_cams; ++i) {
        fill_camera(blocks[i], &cams[i]);
        fill_origin(blocks[i], P.offset, P.scale);
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    rc = launch_accel(accel, P, blocks, n_cams, part, track, stream);
    if (rc == kUnsupportedBasis) return set_error(MNV_E_UNSUPPORTED, "unsupported basis_dim for the accel path");
    return check_hip((hipError_t)rc, "march_accel_kernel");
}

int mnv_render_voxels_accel_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                               const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, nullptr, hip_stream, inputs);
}

int mnv_render_voxels_accel_batch(const mnv_accel *accel, const mnv_camera *cams, int32_t n_cams,
                                  const mnv_render_options *opt, mnv_rect tile, mnv_partition part, float *rgba_out,
                                  uint8_t *rgba8_out, void *hip_stream) {
    return render_accel(accel, cams, n_cams, opt, tile, part, rgba_out, rgba8_out, nullptr, hip_stream);
}

int mnv_render_voxels_accel_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                  mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                  float *sample_track, const int16_t *sample_counts, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (!split_track && !sample_track)
        return mnv_render_voxels_accel(accel, cam, opt, tile, rgba_out, rgba8_out, hip_stream);
    const mnv_partition whole = {0, 1, 0, 0, 0};
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_render_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, float *rgba_out,
                                  uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                  const int32_t *parent, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (!visited) return mnv_render_voxels_accel_track(accel, cam, opt, tile, rgba_out, rgba8_out, split_track, sample_track, sample_counts, hip_stream);
    if (!parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    const mnv_partition whole = {0, 1, 0, 0, 0};
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_render_voxels_accel_visit_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                     const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                                     const int16_t *sample_counts, int32_t *visited, const int32_t *parent, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (!split_track && !sample_track && !visited) return mnv_render_voxels_accel_ex(accel, cam, opt, tile, inputs, rgba_out, rgba8_out, hip_stream);
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    const mnv_partition whole = {0, 1, 0, 0, 0};
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, hip_stream, inputs);
}

int mnv_render_voxels_accel_visit_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts,
                                       int32_t *visited, const int32_t *parent, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (!split_track && !sample_track && !visited) return mnv_render_voxels_accel_part(accel, cam, opt, tile, part, rgba_out, rgba8_out, hip_stream);
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_get_samples_from_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                      float *split_track, float *sample_track, const int16_t *sample_counts, int16_t *num_samples,
                                      float *samples, int32_t samples_dim, int16_t *cluster_indices, const mnv_cluster_grid *grid,
                                      void *hip_stream) {
    return mnv_get_samples_from_voxels_accel_visit(accel, cam, opt, tile, split_track, sample_track, sample_counts, nullptr, nullptr, num_samples,
                                                   samples, samples_dim, cluster_indices, grid, hip_stream);
}

int mnv_get_samples_from_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                            float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                            const int32_t *parent, int16_t *num_samples, float *samples, int32_t samples_dim,
                                            int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream) {
    return mnv_get_samples_from_voxels_accel_visit_ex(accel, cam, opt, tile, nullptr, split_track, sample_track, sample_counts, visited, parent, num_samples,
                                                      samples, samples_dim, cluster_indices, grid, hip_stream);
}

int mnv_get_samples_from_voxels_accel_visit_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                               const mnv_frame_inputs *inputs, float *split_track, float *sample_track, const int16_t *sample_counts,
                                               int32_t *visited, const int32_t *parent, int16_t *num_samples, float *samples, int32_t samples_dim,
                                               int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream) {
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (!opt || !num_samples || !samples || !cluster_indices || !grid) return set_error(MNV_E_INVALID, "null argument");
    const int need = 4 + (opt->need_viewdir ? 3 : 0) + (opt->appearance_embedding != -1 ? 1 : 0);
    if (samples_dim != need) return set_error(MNV_E_INVALID, "samples_dim must be 4 + 3 * need_viewdir + (appearance_embedding != -1)");
    if (opt->max_guided_samples < 1) return set_error(MNV_E_INVALID, "max_guided_samples must be positive");
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.num_samples = num_samples;
    track.samples = samples;
    track.cluster_indices = cluster_indices;
    track.max_guided_samples = opt->max_guided_samples;
    track.samples_dim = samples_dim;
    track.need_viewdir = opt->need_viewdir ? 1 : 0;
    track.appearance_embedding = opt->appearance_embedding;
    track.grid = grid;
    track.visited = visited;
    track.parent = parent;
    const mnv_partition whole = {0, 1, 0, 0, 0};
    const mnv_frame_inputs limit_only = {inputs ? inputs->tmax_px : nullptr, nullptr};  // (this call writes no image: rgba8_init has no role)
    return render_accel(accel, cam, 1, opt, tile, whole, nullptr, nullptr, &track, hip_stream, &limit_only);
}

int mnv_render_guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                            const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, unsigned long long *sample_counter,
                            void *hip_stream) {
    return mnv_render_guided_fused_track(accel, cam, opt, tile, mlp, grid, rgba_out, rgba8_out, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         sample_counter, hip_stream);
}

static int guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                        const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                        float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                        unsigned long long *sample_counter, void *hip_stream, const mnv_frame_inputs *inputs = nullptr);

int mnv_render_guided_fused_track_ex(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                     const mnv_frame_inputs *inputs, const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out,
                                     float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                                     unsigned long long *sample_counter, void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return guided_fused(accel, cam, opt, tile, whole, mlp, grid, rgba_out, rgba8_out, split_track, sample_track, sample_counts, visited, parent,
                        sample_counter, hip_stream, inputs);
}

int mnv_render_guided_fused_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                                  const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                                  const int16_t *sample_counts, int32_t *visited, const int32_t *parent, unsigned long long *sample_counter,
                                  void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return guided_fused(accel, cam, opt, tile, whole, mlp, grid, rgba_out, rgba8_out, split_track, sample_track, sample_counts, visited, parent,
                        sample_counter, hip_stream);
}

int mnv_render_guided_fused_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                 const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out,
                                 unsigned long long *sample_counter, void *hip_stream) {
    return guided_fused(accel, cam, opt, tile, part, mlp, grid, rgba_out, rgba8_out, nullptr, nullptr, nullptr, nullptr, nullptr, sample_counter,
                        hip_stream);
}

int mnv_render_guided_fused_track_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                       float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                                       unsigned long long *sample_counter, void *hip_stream) {
    return guided_fused(accel, cam, opt, tile, part, mlp, grid, rgba_out, rgba8_out, split_track, sample_track, sample_counts, visited, parent,
                        sample_counter, hip_stream);
}

static int guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                        const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                        float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                        unsigned long long *sample_counter, void *hip_stream, const mnv_frame_inputs *inputs) {
    if (!accel || !cam || !opt || !mlp || !grid) return set_error(MNV_E_INVALID, "null argument");
    {
        // A spin-wait that the watchdog of guided_fused2_kernel abandoned leaves wrong pixels behind.  The launches are asynchronous, so the
        // frame itself cannot answer for it: the NEXT call on this accel does, once per fault, and mnv_accel_fused_faults reads the count.
        mnv_accel *mut = const_cast<mnv_accel *>(accel);
        std::lock_guard<std::mutex> g(mut->launch_mutex);
        const uint32_t seen = *static_cast<volatile uint32_t *>(mut->fault_host);
        if (seen != mut->fault_reported) {
            const uint32_t n = seen - mut->fault_reported;
            mut->fault_reported = seen;
            fprintf(stderr, "libmnv: %u wavefront(s) of an earlier fused guided-sampling frame on this accel abandoned a spin-wait (watchdog): that frame is wrong\n", n);
            return set_error(MNV_E_FAULT, "an earlier fused guided-sampling frame on this accel ran into the kernel's watchdog and is wrong; mnv_accel_set_fused_kernel(accel, 1) selects the kernel without spin-waits");
        }
    }
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (opt->render_depth) return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame has no depth mode; use the four-step path");
    if (opt->max_guided_samples < 1) return set_error(MNV_E_INVALID, "max_guided_samples must be positive");
    const MlpShape &S = mlp->shape;
    if (S.hidden_width != 64 || S.nkk0 > 2)
        return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame runs 64-wide networks with at most 64 encoded inputs; use the four-step path");
    if (S.out_dim != accel->view.data_dim + 1) return set_error(MNV_E_INVALID, "the model's out_dim must be the tree's data_dim + 1 (cuda_renderer.cpp:255-257)");
    if ((S.need_viewdir != 0) != (opt->need_viewdir != 0)) return set_error(MNV_E_INVALID, "options.need_viewdir does not match the model");
    if (S.n_embeddings > 0 && opt->appearance_embedding == -1) return set_error(MNV_E_INVALID, "the model needs an appearance embedding");
    const int b = (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim >= 0) ? accel->view.basis_dim : -1;
    if (!(b == -1 || b == 1 || b == 4 || b == 9 || b == 16))
        return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame supports RGBA and SH1/4/9/16 trees; use the four-step path");
    FusedGuided F;
    std::memset(static_cast<void *>(&F), 0, sizeof(F));
    F.S = S;
    F.frags = mlp->frags;
    F.biases = mlp->biases;
    F.embeddings = mlp->embeddings;
    for (int i = 0; i < 2; ++i) F.grid_dim[i] = grid->grid_dim[i];
    for (int i = 0; i < 3; ++i) {
        F.min_position[i] = grid->min_position[i];
        F.range[i] = grid->range[i];
    }
    F.max_guided_samples = opt->max_guided_samples;
    F.appearance_embedding = opt->appearance_embedding;
    static const int env_batch = knob_int(KNOB_FUSED_BATCH_MIN, kFW);
    F.batch_min = env_batch < 1 ? 1 : (env_batch > kFW ? kFW : env_batch);
    F.sample_counter = sample_counter;
    F.diag = accel->fused_diag.load(std::memory_order_relaxed);
    static const int env_switch = knob_int(KNOB_F2_SWITCH_MIN, 32);
    F.switch_min = env_switch;
    AccelTrack track = {};
    track.fused = &F;
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    // offscreen == false: the march stops at the pixel's t_max (get_samples_from_voxels_kernel, renderer_kernel.cu:354-357); the composite of the
    // network's results adds the image under the volume with weight 1 - out[3] = 0 (render_nerf_results_kernel leaves out[3] at 1, :316,
    // composite_and_write :224-234): rgba8_init changes no pixel and is not read
    const mnv_frame_inputs limit_only = {inputs ? inputs->tmax_px : nullptr, nullptr};
    return render_accel(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, &track, hip_stream, &limit_only);
}

}  // extern "C"

namespace mnv {
int render_accel_for_tree(const mnv_accel *accel, const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                          const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                          hipStream_t stream) {
    float xform[6];
    std::memcpy(xform, tree->offset, sizeof(tree->offset));
    std::memcpy(xform + 3, tree->scale, sizeof(tree->scale));
    const mnv_partition whole = {0, 1, 0, 0, 0};
    if (!split_track && !sample_track) return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, nullptr, (void *)stream, inputs, xform);
    // with the refinement trackers (the reference passes them with every call, cuda_renderer.cpp:141-142): the tracker instantiation of the
    // tuned kernel, the voxels' sample counts from the CALL's view
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = tree->sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, (void *)stream, inputs, xform);
}
}  // namespace mnv
