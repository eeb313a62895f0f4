// mnv_accel_launch.h -- what the translation units of the packed-layout ("accel") path share: the launch block of the march
// kernels, the row / lookup-grid index helpers, the tracker block of a launch and the functions that cross the unit boundaries.
//   mnv_accel_build.hip          build kernels, mnv_accel_create / _rebuild / _destroy
//   mnv_accel_refresh_prune.hip  mnv_accel_refresh (patch after a refinement step), accel_apply_prune (the layout follows a prune)
//   mnv_march_accel_kernel.h     march_accel_kernel (the tuned march), instantiated by mnv_accel_march.hip
//   mnv_guided_fused*.h          the guided-sampling frame as one kernel, instantiated by mnv_accel_fused.hip
//   mnv_accel_capi.hip           launch planning (launch_accel), tile assembly, the C-ABI entry points
#pragma once

#include "mnv_accel.h"
#include "mnv_internal.h"
#include "mnv_mlp.h"

namespace mnv {

// Brick-ordered index of cell (cx,cy,cz) of the level-L2 grid: 4x4x4-cell bricks (256 B) in
// row-major brick order, so that the cells neighbouring rays touch share cache lines.
__host__ __device__ __forceinline__ uint32_t grid2_index(uint32_t cx, uint32_t cy, uint32_t cz, int L2) {
    const int LB = L2 - 2;
    const uint32_t brick = (((((cx >> 2) << LB) + (cy >> 2))) << LB) + (cz >> 2);
    return (brick << 6) | ((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u);
}

// grid2i word of a level-L2 cell whose grid2 word is `word` (mnv_accel.h): a non-leaf cell whose chunk holds eight leaves carries their
// sigma != 0 mask inline
__device__ __forceinline__ uint32_t inline_cell_word(const uint32_t *__restrict__ nodes, uint32_t word, uint32_t inline_base) {
    if (!(word & kLeafBit) && word >= inline_base && word - inline_base < (1u << kInlineMaskShift)) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(nodes + (int64_t)word * 8), hi = *reinterpret_cast<const uint4 *>(nodes + (int64_t)word * 8 + 4);
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        bool leaves = true;
        uint32_t mask = 0u;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            leaves = leaves && (w[s] & kLeafBit) != 0u;
            mask |= ((w[s] & 0xffffu) != 0u ? 1u : 0u) << s;
        }
        if (leaves) word = (word - inline_base) | kInlineBit | (mask << kInlineMaskShift);
    }
    return word;
}

// entry `vox` (= chunk * 8 + s1, the chunk of depth L2 + 1) of the brick records (mnv_accel.h) from the node words
__device__ __forceinline__ uint2 brick_record_entry(const uint32_t *__restrict__ nodes, int64_t vox) {
    const uint32_t w1 = nodes[vox];
    uint32_t child = 0u, codes = 0u;
    if (w1 & kLeafBit) {
        if ((w1 & 0xffffu) == 0u) codes = 0x5555u;  // an empty leaf of depth L2 + 1: code 1 for its eight sub-cells
    } else {
        child = w1;
        for (uint32_t s2 = 0; s2 < 8; ++s2) {
            const uint32_t w2 = nodes[(int64_t)w1 * 8 + s2];
            if (w2 & kLeafBit) codes |= ((w2 & 0xffffu) == 0u ? 2u : 3u) << (2 * s2);
        }
    }
    return make_uint2(child, codes);
}

// The levels below the LDS grid for the kernels that need a leaf's sigma in the step itself and read no colour row (the sample march of the
// fused guided kernels): the second lookup grid -- with inline cell words when A.grid2i is there -- then a brick record, then node words.  A
// non-empty leaf found through an inline word or a record fetches its node word (depth + sigma); an empty one costs nothing more.
// word: the LDS grid's (non-leaf) word in, the leaf word out.  src / vox as in march_accel_kernel: src 1 -- vox is the grid2 cell number (the
// voxel is grid2_vox[vox]); src 2 -- vox is the voxel.  q: integer cell coordinates at level Lq; sh1 = Lq - LL, sh2 = Lq - L2.
__device__ __forceinline__ uint32_t descend_to_leaf(const AccelView &A, const uint32_t (&q)[3], uint32_t word, int sh1, int sh2, int L2, int LL, int &src,
                                                    uint32_t &vox) {
    int sh = sh1;
    if (L2 > LL) {
        const int LB = L2 - 2;
        uint32_t gi = q[0] >> (sh2 + 2);
        gi = (gi << LB) | (q[1] >> (sh2 + 2));
        gi = (gi << LB) | (q[2] >> (sh2 + 2));
        gi = (gi << 2) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh2, 2u);
        gi = (gi << 2) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh2, 2u);
        gi = (gi << 2) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh2, 2u);
        const uint32_t *__restrict__ g2 = A.grid2i ? A.grid2i : A.grid2;
        word = g2[gi];
        sh = sh2;
        src = 1;
        vox = gi;
        if (!(word & kLeafBit) && A.grid2i) {
            uint32_t s1 = __builtin_amdgcn_ubfe(q[0], (uint32_t)(sh2 - 1), 1u);
            s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)(sh2 - 1), 1u);
            s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)(sh2 - 1), 1u);
            if (word & kInlineBit) {
                vox = (((word & ((1u << kInlineMaskShift) - 1u)) + A.inline_base) << 3) | s1;
                src = 2;
                const bool filled = ((word >> (kInlineMaskShift + s1)) & 1u) != 0u;
                return filled ? A.nodes[vox] : (kLeafBit | ((uint32_t)(L2 + 1) << 16));
            }
            if (A.recs != nullptr) {
                uint32_t s2 = __builtin_amdgcn_ubfe(q[0], (uint32_t)(sh2 - 2), 1u);
                s2 = (s2 << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)(sh2 - 2), 1u);
                s2 = (s2 << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)(sh2 - 2), 1u);
                const uint2 e = A.recs[(int64_t)word * 8 + s1];
                const uint32_t code = __builtin_amdgcn_ubfe(e.y, s2 << 1, 2u);
                if (code != 0u) {  // 1 / 2: an empty leaf of depth L2 + 1 / L2 + 2; 3: a leaf of depth L2 + 2 with sigma != 0
                    vox = code == 1u ? ((word << 3) | s1) : ((e.x << 3) | s2);
                    src = 2;
                    return code == 3u ? A.nodes[vox] : (kLeafBit | ((uint32_t)(L2 + 2 - (code == 1u ? 1 : 0)) << 16));
                }
                if (e.x != 0u) {  // voxel s1 is an inner voxel and so is its child s2: the walk goes on below them
                    word = e.x;
                    sh = sh2 - 1;
                }
            }
        }
    }
    while (!(word & kLeafBit)) {
        --sh;
        uint32_t v = (word << 1) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh, 1u);
        v = (v << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh, 1u);
        v = (v << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh, 1u);
        word = A.nodes[v];
        src = 2;
        vox = v;
    }
    return word;
}

struct AccelLaunch {
    FrameParams P;   // tile, options, outputs (P.cam is unused: cameras come from `cams`)
    AccelView A;
    const CamBlock *__restrict__ cams;  // [n_frames] device array (written by stage_launch_kernel)
    uint32_t n_frames;
    uint32_t frame_stride_px;  // pixels between consecutive frames in the output buffers
    uint32_t *queue;           // [kNumQueues] heads, 64 B apart; queue q = band q of every frame, frame-major
    uint32_t tiles_x, n_tiles;
    uint32_t tile_wlog;                   // log2 of the ray-tile width (tile = 2^wlog x 2^(6-wlog) pixels, 8x8 by default)
    uint32_t band_begin[kNumQueues + 1];  // tile ranges per queue
    int32_t lds_level;                    // levels staged in LDS (<= A.grid_level)
    int32_t refill_min;                   // refill a wavefront once this many lanes are idle
    // interleaved macro-tile partition (part_world = 0: plain tile)
    int32_t part_rank, part_world, part_period;  // part_period: mnv_partition.root_period
    uint32_t macro_w, macro_h;            // macro tile size in pixels
    uint32_t macros_x;                    // macro tiles per row of the rectangle
    uint32_t micro_x, micro_per_macro;    // 8x8 micro tiles per macro-tile row / per macro tile
    unsigned long long *stats;            // MODE 1 only: 16 counters
    int32_t count_stats;                  // MODE 1 only: 0 = ablation run without the counters' atomics
    unsigned long long *timeline;         // MODE 1 only (MNV_TIMELINE): per tile {t_grab, t_done, wave, iterations}, then per wave {t_entry, t_exit} (100 MHz ticks)
    uint32_t timeline_tiles;              // tile records (n_tiles * n_frames)
    uint32_t *line_bits;                  // MODE 1 only (MNV_FOOTPRINT): bitmap of the 128-byte lines the loads of this launch touch
    uint32_t line_base[6];                // regions: 0 grid2i / grid2, 1 brick records, 2 node words, 3 colour rows, 4 grid2_vox, 5 grid_vox
    // MODE 2 only: refinement trackers (rt_core.cuh:179-180,237-252,308-321), indexed like the pixels
    float *split_track, *sample_track;
    const int16_t *sample_counts;         // reference layout [capacity][8], may be NULL
    int32_t max_depth, max_sample_count;
    int32_t *visited;                     // MODE 2 / 3 only: visit marks [capacity]; the march marks the chunk of every leaf it steps through,
                                          // close_visit_marks adds the ancestors (= every chunk of every descent, rt_core.cuh:132-134)
    int32_t fast_colour;                  // host side only: the colour-math mode this launch resolved to (the accel's own, else the process-wide one)
    int32_t ablate;                       // diagnostics only (breaks results): 1 no colour, 2 no dense samples, 4 cached rows
    const uint32_t *shadow_nodes;         // -DMNV_SHADOW_MASK variants only: copies of nodes / rows / brick records for the shadow loads
    const uint8_t *shadow_rows;
    const uint2 *shadow_recs;
    // MODE 3 only: the sample-emitting march of guided sampling (rt_core.cuh:418-576) -- no colour, rows of
    // (z, world xyz[, view dir][, embedding]) per dense step and the trackers of MODE 2
    int32_t max_guided_samples, samples_dim, need_viewdir, appearance_embedding;
    int16_t *num_samples;
    float *samples;
    int16_t *cluster_indices;
    int32_t grid_dim[2];
    float min_position[3], range[3];
};

// ray id -> pixel of the rectangle (bx, by) and index of the pixel in the output buffer
__device__ __forceinline__ bool ray_pixel(const AccelLaunch &K, uint32_t id, int &bx, int &by, uint32_t &pix) {
    const uint32_t tile = id >> 6, w = id & 63u;
    if (K.part_world < 1) {
        const uint32_t tx = tile % K.tiles_x, ty = tile / K.tiles_x;
        bx = (int)((tx << K.tile_wlog) + (w & ((1u << K.tile_wlog) - 1u)));
        by = (int)((ty << (6 - K.tile_wlog)) + (w >> K.tile_wlog));
        pix = (uint32_t)by * (uint32_t)K.P.tw + (uint32_t)bx;
    } else {
        const uint32_t j = tile / K.micro_per_macro, u = tile % K.micro_per_macro;
        const uint32_t mx = u % K.micro_x, my = u / K.micro_x;
        const uint32_t m = part_tile_of(j, K.part_rank, K.part_world, K.part_period);
        const uint32_t MX = m % K.macros_x, MY = m / K.macros_x;
        const uint32_t lx = mx * 8 + (w & 7u), ly = my * 8 + (w >> 3);
        bx = (int)(MX * K.macro_w + lx);
        by = (int)(MY * K.macro_h + ly);
        pix = (j * K.macro_h + ly) * K.macro_w + lx;
    }
    return bx < K.P.tw && by < K.P.th;
}

__device__ __forceinline__ float lane_read(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// A packed colour row.  Formats whose rows a lane evaluates by itself (RGBA, SH1 / 4 / 9): the reference's row as it is -- the three channels'
// coefficients back to back, then the voxel's sigma -- padded to a power of two (SH9: 28 halfs = 56 -> 64 B), read with 16-byte loads.
// SH16 / SH25 (the cooperative colour pass, one lane per (sample, channel)): every channel block padded to whole dwords, sigma behind them.
// bytes of one colour channel block:
__host__ __device__ constexpr int chan_bytes_for(int basis) { return basis <= 0 ? 2 : basis < 16 ? 2 * basis : ((2 * basis + 3) / 4) * 4; }
// bytes of a packed row: three channel blocks + the sigma half, rounded up to a power of two (8 ... 256), so that a row never straddles
// a 128-B cache line
__host__ __device__ constexpr int row_bytes_pow2(int basis) {
    int r = 8;
    while (r < 3 * chan_bytes_for(basis) + 2) r *= 2;
    return r;
}
// channel block as dwords with 4-byte alignment (the compiler picks the widest legal loads)
template <int N>
struct __attribute__((packed, aligned(4))) ChanWords {
    uint32_t w[N];
};

// Columns per network run of guided_fused_kernel: W = 16 * MNV_FUSED_NT samples; 64 / W lanes share a column in the per-column phases
// (encode, evaluation).  NT = 4: 64 accumulator + 32 activation registers -> 254 VGPRs, 2 wavefronts per SIMD.  NT = 2: half of that ->
// 3 wavefronts per SIMD (168 VGPRs), which is what the march part of the kernel wants; the weights are then fetched twice per 64 samples.
#ifndef MNV_FUSED_NT
#define MNV_FUSED_NT 4
#endif
constexpr int kFNT = MNV_FUSED_NT, kFW = 16 * kFNT, kFParts = 64 / kFW;  // column tiles, columns and lanes per column of a run

struct FusedGuided {
    MlpShape S;
    const uint16_t *frags;       // [n_clusters][frag_halfs]
    const float *biases;         // [n_clusters][bias_floats]
    const uint16_t *embeddings;  // [n_clusters][n_embeddings][embedding_dim]
    int32_t grid_dim[2];
    float min_position[3], range[3];
    int32_t max_guided_samples, appearance_embedding;
    int32_t batch_min;           // run the network once this many samples wait in a wavefront's pool (1 .. 64)
    unsigned long long *sample_counter;  // += samples evaluated (one atomic per wavefront); ONE word
    unsigned long long *diag;            // diagnostics (mnv_set_fused_diag, 32 words of the caller's): NULL = none
    uint32_t *fault;                     // the accel's fault word: += 1 per wavefront that abandons a spin-wait (never NULL)
    int32_t switch_min;                  // guided_fused2_kernel: a consumer stays with its last sub-module while this many of its samples wait
    int32_t weight_slots;                // guided_fused2_kernel: sub-modules whose weights a workgroup keeps in LDS (<= kF2NS, what fits)
};

// wavefronts per SIMD the march kernel is built for (= workgroups of 256 threads per CU)
#ifndef MNV_TRACK_WAVES
#define MNV_TRACK_WAVES 6  // tracker / sample modes carry six more live values per ray
#endif
#ifndef MNV_MIN_WAVES
#define MNV_MIN_WAVES 8  // register budget for 8 waves per SIMD: the few spills land in the ray set-up (A/B in DESIGN.md)
#endif

// world > 1, or a single rank that asks for the macro-tile-major layout by naming a tile size
inline bool is_partitioned(mnv_partition part) { return part.world > 1 || (part.world == 1 && part.tile_w > 0); }
inline int32_t root_period_of(mnv_partition part) { return part.world > 1 && part.root_period >= 2 ? part.root_period : 0; }
inline int row_bytes_for(int basis) { return row_bytes_pow2(basis); }

// Refinement trackers of one launch (all device pointers; rows indexed like the pixels).
struct AccelTrack {
    float *split_track, *sample_track;
    const int16_t *sample_counts;
    int32_t max_depth, max_sample_count;
    // sample emission (MODE 3) when samples != NULL
    int16_t *num_samples;
    float *samples;
    int16_t *cluster_indices;
    int32_t max_guided_samples, samples_dim, need_viewdir, appearance_embedding;
    const mnv_cluster_grid *grid;
    const FusedGuided *fused;  // non-NULL: guided_fused_kernel instead of the march (no trackers, one frame)
    int32_t *visited;          // visit marks (tracker / sample modes) ...
    const int32_t *parent;     // ... closed under the parent words after the march
};

constexpr int kUnsupportedBasis = -1000;  // not a hipError_t

// (Re)build every derived array of `a` from the tree (mnv_accel_build.hip)
int accel_build(mnv_accel *a, const mnv_tree_view *t, hipStream_t stream);
// the two lookup grids, whole, from the node words (mnv_accel_build.hip; the refresh rebuilds the small one, or both without a parent array)
void launch_pack_rows(const uint16_t *data, uint16_t *rows, int64_t nvox, int32_t data_dim, int32_t per_chan, int32_t chan_halfs, int32_t row_halfs,
                      hipStream_t stream);
// brick records of the chunks of depth L2 + 1 (mnv_accel.h); one thread per chunk
void launch_build_recs(const uint32_t *nodes, const int32_t *depth, uint2 *recs, int32_t capacity, int32_t L2, hipStream_t stream);
void launch_build_grid2i(const uint32_t *nodes, const uint32_t *grid2, uint32_t *grid2i, int32_t L2, uint32_t inline_base, hipStream_t stream);
// smallest chunk number of depth `d` (capacity if there is none); synchronises the stream
int min_chunk_of_depth(const int32_t *depth, int32_t capacity, int32_t d, int32_t *scratch, hipStream_t stream, uint32_t *out);
void launch_build_grid(const uint32_t *nodes, uint32_t *grid, uint32_t *grid_vox, int32_t L, hipStream_t stream);
void launch_build_grid2(const uint32_t *nodes, uint32_t *grid2, uint32_t *grid2_vox, int32_t L2, hipStream_t stream);
// march_accel_kernel for the row format `basis` (-1 RGBA, 1 / 4 / 9 / 16 / 25 SH) in the mode the launch block asks for
// (mnv_accel_march.hip); kUnsupportedBasis or a hipError_t
int launch_march(const AccelLaunch &K, int basis, bool colourless, int n_blocks, size_t lds_bytes, hipStream_t stream);
// the same on inline cell words / brick records (K.A.grid2i != NULL), every frame kind of the per-lane row formats (mnv_accel_march_brick.hip)
int launch_march_brick(const AccelLaunch &K, int basis, bool colourless, int n_blocks, size_t lds_bytes, hipStream_t stream);
// guided_fused2_kernel / guided_fused_kernel (mnv_accel_fused.hip); kUnsupportedBasis or a hipError_t
int launch_fused(const mnv_accel *accel, const AccelLaunch &K, const FusedGuided &fused, int basis, int lds_level, uint64_t n_waves_needed,
                 hipStream_t stream);

}  // namespace mnv
