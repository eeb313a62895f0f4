// mnv_mlp.hip -- per-sample sub-module network, fused on the matrix cores.
//
// Replaces VolumeRenderer::Impl::query_submodules (reference src/renderer/cuda_renderer.cpp:165-203): the
// reference sorts the samples by cluster id and runs one TorchScript module per cluster in batches under fp16
// autocast, scattering the outputs back.  The TorchScript artefacts are not part of the reference repository
// (parity unpinned), so the network here is this build's own small MLP:
//
//   encode(x)  = [p, tri(2^k p), tri(2^k p + 1/4)]_k<pos_octaves  (p = (xyz - center) * inv_extent)
//              ++ the same on the view direction (dir_octaves) when need_viewdir
//              ++ embedding_table[index] when n_embeddings > 0
//   hidden_i   = relu(W_i h + b_i) rounded to binary16, i < hidden_layers
//   out        = W_o h + b_o (fp32)
//   tri(t)     = 4 |t - floor(t + 1/2)| - 1: a triangle wave -- exact IEEE arithmetic, so the encoded inputs are
//                bit-identical on host and device; only the accumulation order inside the MFMA differs from
//                the sequential CPU restatement (tests bound that).
//
// Execution: counting sort of the rows by cluster (histogram + scan + scatter) into tiles of 256 or 512 rows of one cluster, then as many
// persistent workgroups as the device holds, each with a contiguous range of the tiles (weights staged when the cluster changes, the next
// tile's rows and samples in flight across tile boundaries).  The whole network runs in registers: with the weights as the MFMA A operand and the samples
// as the B operand (D^T = W X^T), the C layout of one layer (lane: sample l & 15, features 4 (l >> 4) + r) is
// already a valid B layout for the next layer once the K index is permuted -- and the permutation is baked into
// the weight fragments at upload time, so nothing moves between layers but a float -> half convert.
// v_mfma_f32_16x16x32_f16, fragments staged in LDS once per workgroup.  A pass: the first layer's B fragments are encoded (every
// input block starts at a multiple of 16 K slots, mnv_mlp.h: all features at compile-time places), then every layer runs one M tile
// at a time, a finished pair of M tiles packed into the next layer's B fragment at once; the results leave through a lane
// permutation in LDS as 64-byte contiguous stores; the next pass's rows and samples are fetched meanwhile.

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "mnv_internal.h"
#include "mnv_mlp.h"

namespace mnv {

struct MlpLaunch {
    MlpShape S;
    const uint16_t *frags;
    const float *biases;
    const uint16_t *embeddings;
    const float *samples;
    int32_t samples_stride;
    float *results;
    int32_t result_stride;
    const int32_t *order;       // rows sorted by cluster
    const int32_t *seg_start;   // [n_clusters + 1]: first position of each cluster in `order`
    const int32_t *tile_start;  // [n_clusters + 1]: first tile of each cluster (a tile: up to one pass's rows -- the kernel's ROWS -- of one cluster); [n_clusters] = number of tiles
#ifdef MNV_MLP_CLOCKS
    unsigned long long *clocks;  // measurement variant (tools/mlp_clocks.sh): summed s_memtime per phase of wavefront 0 of every workgroup
#endif
};

// Phase clocks of the measurement variant; nothing in the shipped build.
#ifdef MNV_MLP_CLOCKS
#define MLP_CLOCK(i)                                                    \
    do {                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        clk_[i] += now_ - clk_t_;                                       \
        clk_t_ = now_;                                                  \
    } while (0)
#else
#define MLP_CLOCK(i) \
    do {             \
    } while (0)
#endif

// ---------------------------------------------------------------- counting sort by cluster

// Rows of one chunk of the two sort passes (each thread handles kSortItems rows of a chunk); a block takes `chunks` consecutive chunks and
// talks to the global counters ONCE per bin -- with one chunk per block the 8 M rows of the profile were 3906 blocks x 8 bins = 31 k atomics on
// eight addresses per pass, which the L2 serialises: 50 us for a pass that reads 16 MB.
constexpr int kSortItems = 8;
constexpr int kSortRows = 256 * kSortItems;

// (Ranks within a bin: one LDS atomic per row.  Until round 5 a wavefront-aggregated form -- one atomic per distinct bin of the wavefront, found
// by a ballot / shuffle loop: cheaper when neighbouring rows share a bin, but the loop runs once per distinct bin, and on incoherent rows --
// the profile's 8 random clusters -- that was eight trips of ten instructions per row against one ds_add_rtn whose same-address lanes the
// LDS serialises by itself.  The order of rows within a bin is whatever the LDS makes it: rows are independent columns of the matrix
// products, their results do not depend on their place.)
// The thread's eight CONSECUTIVE rows of a chunk (rows base + 8 t .. + 7): one 16-byte load when the chunk is whole and the array 16-byte
// aligned, eight 2-byte loads otherwise; -1 past the end.
__device__ inline void load_chunk_rows(const int16_t *__restrict__ cluster, int64_t n, int64_t base, bool aligned, int (&c)[kSortItems]) {
    static_assert(kSortItems == 8, "eight 16-bit rows per 16-byte load");
    const int64_t i0 = base + (int64_t)threadIdx.x * kSortItems;
    if (aligned && base + kSortRows <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(cluster + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) c[k] = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
    } else {
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) c[k] = i0 + k < n ? cluster[i0 + k] : -1;
    }
}

__global__ __launch_bounds__(256) void mlp_histogram(const int16_t *__restrict__ cluster, int64_t n, int32_t n_clusters, int32_t chunks,
                                                     int32_t *__restrict__ counts, float *__restrict__ results, int32_t result_stride,
                                                     int32_t out_dim) {
    __shared__ int32_t s_bins[kMaxClusters];
    for (int c = threadIdx.x; c < n_clusters; c += 256) s_bins[c] = 0;
    __syncthreads();
    const bool aligned = (reinterpret_cast<uintptr_t>(cluster) & 15) == 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const int64_t base = ((int64_t)blockIdx.x * chunks + ch) * kSortRows;
        if (base >= n) break;
        int c[kSortItems];
        load_chunk_rows(cluster, n, base, aligned, c);
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const int64_t i = base + (int64_t)threadIdx.x * kSortItems + k;
            const bool valid = c[k] >= 0 && c[k] < n_clusters;
            if (valid) atomicAdd(&s_bins[c[k]], 1);
            if (i < n && !valid)
                for (int o = 0; o < out_dim; ++o) results[i * result_stride + o] = 0.f;  // no sub-module: zeros
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_clusters; c += 256)
        if (s_bins[c]) atomicAdd(&counts[c], s_bins[c]);
}

// one block: exclusive scans of the counts -> segment starts and first tile of every cluster
__global__ void mlp_plan(const int32_t *__restrict__ counts, int32_t n_clusters, int32_t *__restrict__ seg_start,
                         int32_t *__restrict__ cursor, int32_t *__restrict__ tile_start, int32_t rows_per_tile) {
    __shared__ int32_t s_start[kMaxClusters + 1], s_tile[kMaxClusters + 1];
    if (threadIdx.x == 0) {
        int32_t a = 0, t = 0;
        for (int c = 0; c < n_clusters; ++c) {
            s_start[c] = a;
            s_tile[c] = t;
            a += counts[c];
            t += (counts[c] + rows_per_tile - 1) / rows_per_tile;
        }
        s_start[n_clusters] = a;
        s_tile[n_clusters] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= n_clusters; c += blockDim.x) {
        seg_start[c] = s_start[c];
        tile_start[c] = s_tile[c];
        if (c < n_clusters) cursor[c] = s_start[c];
    }
}

__global__ __launch_bounds__(256) void mlp_scatter(const int16_t *__restrict__ cluster, int64_t n, int32_t n_clusters, int32_t chunks,
                                                   int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    __shared__ int32_t s_bins[kMaxClusters], s_base[kMaxClusters];
    for (int c = threadIdx.x; c < n_clusters; c += 256) s_bins[c] = 0;
    __syncthreads();
    const bool aligned = (reinterpret_cast<uintptr_t>(cluster) & 15) == 0;
    // the block's rows per bin (its chunks are read twice: 2 bytes a row, the second time from the L2) ...
    for (int ch = 0; ch < chunks; ++ch) {
        const int64_t base = ((int64_t)blockIdx.x * chunks + ch) * kSortRows;
        if (base >= n) break;
        int c[kSortItems];
        load_chunk_rows(cluster, n, base, aligned, c);
#pragma unroll
        for (int k = 0; k < kSortItems; ++k)
            if (c[k] >= 0 && c[k] < n_clusters) atomicAdd(&s_bins[c[k]], 1);
    }
    __syncthreads();
    // ... one global reservation per bin and block ...
    for (int c = threadIdx.x; c < n_clusters; c += 256) {
        s_base[c] = s_bins[c] ? atomicAdd(&cursor[c], s_bins[c]) : 0;
        s_bins[c] = 0;
    }
    __syncthreads();
    // ... and every row to its place
    for (int ch = 0; ch < chunks; ++ch) {
        const int64_t base = ((int64_t)blockIdx.x * chunks + ch) * kSortRows;
        if (base >= n) break;
        int c[kSortItems];
        load_chunk_rows(cluster, n, base, aligned, c);
#pragma unroll
        for (int k = 0; k < kSortItems; ++k)
            if (c[k] >= 0 && c[k] < n_clusters) order[s_base[c[k]] + atomicAdd(&s_bins[c[k]], 1)] = (int32_t)(base + (int64_t)threadIdx.x * kSortItems + k);
    }
}

// ---------------------------------------------------------------- the network

// One workgroup = WAVES wavefronts x (16 * NT) rows per pass.  <4, 4, 4, .>: 64-wide networks, 256 rows per pass.  <8, 4, 8, .>: 128-wide networks --
// their weights fill most of a CU's LDS (131 KB for 4 hidden layers), so one workgroup per CU is all that fits: eight wavefronts (two per SIMD, the
// 256-register budget) of 64 columns each share the weights, 512 rows per pass; every weight fragment read from LDS (1 KB, 8 cycles of the CU's LDS
// port) feeds four 16-cycle MFMAs.  (History: <8, 2, 8> with 32 columns per wavefront until round 4; 128 columns at one wavefront per SIMD,
// <8, 8, 4>, measured 2.3 - 3.7 ms against 1.4 in round 5 and was dropped.)  The encode tiles hold HALF a K tile (16 features: 2 KB per
// wavefront, written and read twice per K tile) -- with whole tiles the 131 KB of weights + 32 KB would not fit a CU.

// tri_wave(x * scale + phase) (mnv_mlp.h) in five instructions instead of seven, bit for bit: scale is a power of two, so x * scale is exact and
// one fused multiply-add rounds exactly where the sum rounds; 4 |r| is exact, so the closing fused multiply-add rounds where the subtraction
// rounds; and the "+ 0.f" of the phase-0 features changes no result (t = -0 gives r = -0, |r| = 0, as t = +0 does).
__device__ __forceinline__ float tri_of(float x, float scale, bool quarter_phase) {
    const float t = quarter_phase ? __builtin_fmaf(x, scale, 0.25f) : x * scale;
    const float r = t - floorf(t + 0.5f);
    return __builtin_fmaf(4.f, fabsf(r), -1.f);
}

// An entry of a table that was written before this kernel started (seg_start, tile_start): read through the constant address space it is a
// scalar load wherever it stands -- behind the kernel's own stores a plain load is a vector load with a wait for everything in flight.
__device__ __forceinline__ int32_t load_uniform(const int32_t *p) { return *(const __attribute__((address_space(4))) int32_t *)(p); }

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int MT, int NT, int WAVES, int NKK0>  // hidden width = 16 * MT; NKK0: K tiles of the first layer, 1 .. 4, or 0 for "S.nkk0, any"
__global__ __launch_bounds__(64 * WAVES, (MT * NT >= 32 ? 1 : 2)) void mlp_forward_kernel(const MlpLaunch L) {
    constexpr int ROWS = WAVES * 16 * NT;  // rows of one pass of the workgroup
    constexpr int OWN = (16 * NT + 63) / 64;  // samples a lane encodes: lane + 64 o
    static_assert(ROWS % kRowsPerPass == 0 && (16 * NT) % 64 == 0, "a pass is a multiple of 256 rows, a wavefront's columns a multiple of its lanes");
    constexpr int COLS = 16 * NT;  // samples (MFMA columns) of one wavefront
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const MlpShape &S = L.S;
    // Persistent workgroups (round 6): the launch holds as many workgroups as the device runs at once, each takes a CONTIGUOUS range of the
    // tiles (a tile = one pass = up to ROWS rows of one cluster) and stages a cluster's weights only when the cluster changes -- with 8 clusters
    // over 8 M rows once or twice per workgroup instead of once per 4096 rows, no workgroup start / drain between tiles, the next tile's rows
    // and samples in flight across the tile boundary, and ranges that differ by one tile at most (1953 workgroups of 4096 rows on 256 compute
    // units were 8 rounds for some and 7 for others).
    const int n_tiles = load_uniform(L.tile_start + S.n_clusters);
    const int tile_first = (int)(((int64_t)blockIdx.x * n_tiles) / gridDim.x), tile_end = (int)(((int64_t)(blockIdx.x + 1) * n_tiles) / gridDim.x);
    if (tile_first >= tile_end) return;
    // the cluster whose tile range holds the first tile: last c with tile_start[c] <= tile_first (uniform: scalar loads)
    int cluster_next;
    {
        int lo = 0, hi = S.n_clusters;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (load_uniform(L.tile_start + mid) <= tile_first) lo = mid;
            else hi = mid;
        }
        cluster_next = lo;
    }
    // tiles and rows of cluster_next: [cl_tile0, cl_tile1) and [cl_seg0, cl_seg1) -- scalar registers, read again only where the cluster changes
    int cl_tile0 = load_uniform(L.tile_start + cluster_next), cl_tile1 = load_uniform(L.tile_start + cluster_next + 1);
    int cl_seg0 = load_uniform(L.seg_start + cluster_next), cl_seg1 = load_uniform(L.seg_start + cluster_next + 1);
#ifdef MNV_MLP_CLOCKS
    unsigned long long clk_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clk_t_ = __builtin_readcyclecounter();
    const unsigned long long clk_begin_ = clk_t_, wall_begin_ = wall_clock64();
#endif

    const half8 *s_frag = reinterpret_cast<const half8 *>(lds);
    const float *s_bias = reinterpret_cast<const float *>(lds + (size_t)S.frag_halfs * 2);
    // weights and biases of a cluster -> LDS
    auto stage_weights = [&](int cluster) __attribute__((always_inline)) {
        // (eight 16-byte loads in flight per thread: one load per trip of a plain copy loop costs a trip to the L2 each -- 18 trips for 147 KB)
        const uint4 *src = reinterpret_cast<const uint4 *>(L.frags + (size_t)cluster * S.frag_halfs);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        const int n16 = S.frag_halfs / 8;
        int i0 = threadIdx.x;
        for (; i0 + 7 * 64 * WAVES < n16; i0 += 8 * 64 * WAVES) {
            uint4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = src[i0 + k * 64 * WAVES];
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[i0 + k * 64 * WAVES] = t[k];
        }
        for (; i0 < n16; i0 += 64 * WAVES) dst[i0] = src[i0];
        const float *bsrc = L.biases + (size_t)cluster * S.bias_floats;
        float *bdst = reinterpret_cast<float *>(lds + (size_t)S.frag_halfs * 2);
        for (int i = threadIdx.x; i < S.bias_floats; i += blockDim.x) bdst[i] = bsrc[i];
    };

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    // the wavefront's encode tile: HALF a K tile (16 features) of its COLS samples, a sample's 32 bytes = four lane groups x 8 bytes (features
    // 4 gg .. 4 gg + 3 of the half): written as four 8-byte stores by the lane that owns the sample, read as one 8-byte load per column tile
    // and half by the lane that feeds the matrix pipe (whole K tiles of the 512-row kernels would not fit a CU beside 131 KB of weights)
    uint2 *s_enc2 = reinterpret_cast<uint2 *>(lds + (size_t)S.frag_halfs * 2 + (size_t)S.bias_floats * 4) + wave * (COLS * 4);
    // (a sample's four 8-byte groups are stored rotated by two bits of its index -- samples 8 apart would meet on the same banks, on the
    // writing side, lane = sample, and on the reading side, lane (g, col) reads group g of sample col + 16 nt: four-way conflicts both)
    const int own_swz = ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1);  // of sample `lane` (and of lane + 64: bits 3 and 4 are the same)
    auto read_swz = [&](int nt) { return ((col >> 3) & 1) * 2 + (nt & 1); };  // of sample col + 16 nt

    // Rows and samples of a pass are fetched DURING the pass before it (a row index from `order`, then that row's three to seven floats: two trips
    // to memory in a row, 17 % of a wavefront's time when they opened each pass): the indices while the pass encodes, the floats while its
    // layers run.  Two roles per lane.  Encoding: lane j owns sample j of the wavefront.  Matrix operand / output: lane (g, col) serves the
    // samples col + 16 nt.  The store: lane l writes 16 bytes (features 4 (l & 3) .. + 3 of an M tile) of sample (l >> 2) + 16 nt -- four
    // neighbouring lanes cover 64 contiguous bytes of one result row (the accumulators' own layout would send 64 separate 4-byte writes
    // per instruction).
    int32_t dst_row_next[NT], own_next[OWN];
    float x_next[OWN][7];
    // tile -> its cluster (tiles of a range come in cluster order: the cluster only moves forward, over clusters without rows), first row, rows
    auto fetch_rows = [&](int tile) __attribute__((always_inline)) {
        while (cl_tile1 <= tile) {
            ++cluster_next;
            cl_tile0 = cl_tile1, cl_seg0 = cl_seg1;
            cl_tile1 = load_uniform(L.tile_start + cluster_next + 1), cl_seg1 = load_uniform(L.seg_start + cluster_next + 1);
        }
        const int first = cl_seg0 + (tile - cl_tile0) * ROWS;
        const int rows = min(ROWS, cl_seg1 - first);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int local = wave * COLS + nt * 16 + (lane >> 2);
            dst_row_next[nt] = local < rows ? L.order[first + local] : -1;
        }
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            const int local = wave * COLS + 64 * o + lane;
            own_next[o] = L.order[first + (local < rows ? local : 0)];
        }
    };
    auto fetch_samples = [&]() __attribute__((always_inline)) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            const float *x = L.samples + (int64_t)own_next[o] * L.samples_stride;
            if (S.need_viewdir) {  // six floats of one row: a 16-byte and an 8-byte load (rows are 4-byte aligned)
                const f4u a = *reinterpret_cast<const f4u *>(x);
                const f2u c = *reinterpret_cast<const f2u *>(x + 4);
                x_next[o][0] = a[0], x_next[o][1] = a[1], x_next[o][2] = a[2], x_next[o][3] = a[3], x_next[o][4] = c[0], x_next[o][5] = c[1];
            } else {
                const f2u a = *reinterpret_cast<const f2u *>(x);
                x_next[o][0] = a[0], x_next[o][1] = a[1], x_next[o][2] = x[2];
                x_next[o][3] = x_next[o][4] = x_next[o][5] = 0.f;
            }
            x_next[o][6] = S.n_embeddings > 0 ? x[S.need_viewdir ? 6 : 3] : 0.f;
        }
    };
    // Loads and stores count down the same counter and may pass each other, so a wait for a load that stands behind stores is a wait for
    // the stores as well: what the loads of tile t + 1 brought is taken over (position, direction, embedding row; the rows of the results)
    // BEFORE tile t's output layer sends its stores, and nothing at the top of a tile waits for memory.
    float p[OWN][3], d[OWN][3];
    int32_t emb_row[OWN], dst_row[NT];
    auto take_over = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                p[o][i] = (x_next[o][i] - S.center[i]) * S.inv_extent[i];
                d[o][i] = x_next[o][3 + i];
            }
            emb_row[o] = 0;
            if (S.n_embeddings > 0) {
                asm volatile("" : "+v"(x_next[o][6]));  // (or the conversion moves up to the load, and waits for it there)
                const int idx = (int)x_next[o][6];
                emb_row[o] = idx < 0 ? 0 : (idx >= S.n_embeddings ? S.n_embeddings - 1 : idx);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(dst_row_next[nt]));  // (arrived: loads return in order, and x_next came later)
    };
    fetch_rows(tile_first);
    fetch_samples();
    take_over();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dst_row[nt] = dst_row_next[nt];

    int cluster = -1;  // the cluster whose weights are in LDS
    int tile_cluster = cluster_next;  // the cluster of the tile at hand (cluster_next runs one tile ahead)
    for (int tile = tile_first; tile < tile_end; ++tile) {
        if (tile_cluster != cluster) {  // (uniform)
            if (cluster >= 0) __syncthreads();  // every wavefront is done with the old weights
            cluster = tile_cluster;
            stage_weights(cluster);
            __syncthreads();
            MLP_CLOCK(0);
        }
        const uint16_t *emb[OWN];
#pragma unroll
        for (int o = 0; o < OWN; ++o)
            emb[o] = S.n_embeddings > 0 ? L.embeddings + ((size_t)cluster * S.n_embeddings + emb_row[o]) * S.embedding_dim : nullptr;
        // (the last tile of the range fetches itself again: loads under a condition would meet the other path's values in copies that the
        // compiler puts right behind the loads -- a wait for memory at the top of every tile)
        fetch_rows(tile + 1 < tile_end ? tile + 1 : tile);
        // Encoding of one HALF of a K tile (16 K slots) of this lane's samples into the wavefront's tile.  The position block, the direction
        // block and the embedding each start at a multiple of 16 slots (mnv_mlp.h), so a half tile holds features of ONE block at
        // compile-time places: octave, axis and phase of every feature are constants of the code -- seven or eight instructions per feature,
        // no scalar decoding and no branch (until round 5 the direction block started right behind the position block, at a run-time
        // slot, and its features went through a decoding loop with a branch each: 40 % of a wavefront's time; a branch-free decoder was
        // 41 instructions per feature, and at two wavefronts per SIMD every instruction, scalar or not, is four cycles of the wavefront).
        // The half that holds a block's END masks what lies past it (scalar masks).
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        typedef float float2v __attribute__((ext_vector_type(2)));
        auto pack2 = [](float x0, float x1) __attribute__((always_inline)) -> uint32_t {
            const float2v pr = {x0, x1};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, half2v));
        };
        // feature j of a block over the coordinates x: j < 3 the coordinate, then per octave k three phase-0 and three phase-1/4 triangle waves
        auto block_feature = [&](const float (&x)[3], auto j_tag) __attribute__((always_inline)) -> float {
            constexpr int j = decltype(j_tag)::value;
            if constexpr (j < 3) {
                return x[j];
            } else {
                constexpr int k = (j - 3) / 6, r = (j - 3) % 6, i = r % 3;
                const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
                return tri_of(x[i], scale, r >= 3);
            }
        };
        // features j0 .. j0 + 3 of a block -> 8 bytes of the tile; `masked`: features from n on are zeros (n is wave-uniform)
        auto block_group = [&](int o, const float (&x)[3], auto j0_tag, int gg, auto masked_tag, int n) __attribute__((always_inline)) {
            constexpr int j0 = decltype(j0_tag)::value;
            constexpr bool masked = decltype(masked_tag)::value;
            uint32_t lo = pack2(block_feature(x, std::integral_constant<int, j0>{}), block_feature(x, std::integral_constant<int, j0 + 1>{}));
            uint32_t hi = pack2(block_feature(x, std::integral_constant<int, j0 + 2>{}), block_feature(x, std::integral_constant<int, j0 + 3>{}));
            if constexpr (masked) {
                lo &= (j0 < n ? 0xffffu : 0u) | (j0 + 1 < n ? 0xffff0000u : 0u);
                hi &= (j0 + 2 < n ? 0xffffu : 0u) | (j0 + 3 < n ? 0xffff0000u : 0u);
            }
            s_enc2[(64 * o + lane) * 4 + (gg ^ own_swz)] = make_uint2(lo, hi);
        };
        auto block_half = [&](int o, const float (&x)[3], auto m_tag, int n) __attribute__((always_inline)) {  // features 16 m .. 16 m + 15 of a block of n
            constexpr int m = decltype(m_tag)::value;
            if (16 * m + 16 <= n) {
                block_group(o, x, std::integral_constant<int, 16 * m>{}, 0, std::false_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 4>{}, 1, std::false_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 8>{}, 2, std::false_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 12>{}, 3, std::false_type{}, n);
            } else {
                block_group(o, x, std::integral_constant<int, 16 * m>{}, 0, std::true_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 4>{}, 1, std::true_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 8>{}, 2, std::true_type{}, n);
                block_group(o, x, std::integral_constant<int, 16 * m + 12>{}, 3, std::true_type{}, n);
            }
        };
        // half m of a block, m wave-uniform and at most `last` (a block of 16 octaves is 99 features: seven halves)
        auto block_half_of = [&](int o, const float (&x)[3], int m, auto last_tag, int n) __attribute__((always_inline)) {
            constexpr int last = decltype(last_tag)::value < 6 ? decltype(last_tag)::value : 6;
            if (m == 0 || last == 0) {
                block_half(o, x, std::integral_constant<int, 0>{}, n);
            } else if (m == 1 || last == 1) {
                if constexpr (last >= 1) block_half(o, x, std::integral_constant<int, 1>{}, n);
            } else if (m == 2 || last == 2) {
                if constexpr (last >= 2) block_half(o, x, std::integral_constant<int, 2>{}, n);
            } else if (m == 3 || last == 3) {
                if constexpr (last >= 3) block_half(o, x, std::integral_constant<int, 3>{}, n);
            } else if (m == 4 || last == 4) {
                if constexpr (last >= 4) block_half(o, x, std::integral_constant<int, 4>{}, n);
            } else if (m == 5 || last == 5) {
                if constexpr (last >= 5) block_half(o, x, std::integral_constant<int, 5>{}, n);
            } else {
                if constexpr (last >= 6) block_half(o, x, std::integral_constant<int, 6>{}, n);
            }
        };
        // one half tile; static_tag: the half tile as a compile-time constant, or -1 (the K-outer loop of the kernels without a compile-time
        // tile count: p and d then pass through an empty statement per trip -- as invariants of that loop every feature, a function of p or
        // d alone, would be computed before it and held in registers)
        auto encode_half = [&](int half_tile, auto static_tag) __attribute__((always_inline)) {
            constexpr int T = decltype(static_tag)::value;
            const int pos_halves = S.dir_base >> 4, dir_halves = (S.emb_base - S.dir_base) >> 4;
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                float xp[3] = {p[o][0], p[o][1], p[o][2]}, xd[3] = {d[o][0], d[o][1], d[o][2]};
                if constexpr (T < 0) asm volatile("" : "+v"(xp[0]), "+v"(xp[1]), "+v"(xp[2]), "+v"(xd[0]), "+v"(xd[1]), "+v"(xd[2]));
                if (half_tile < pos_halves) {
                    if constexpr (T >= 0 && T <= 6) block_half(o, xp, std::integral_constant<int, (T >= 0 && T <= 6) ? T : 0>{}, S.n_pos);
                    else if constexpr (T < 0) block_half_of(o, xp, half_tile, std::integral_constant<int, 6>{}, S.n_pos);
                } else if (half_tile - pos_halves < dir_halves) {
                    block_half_of(o, xd, half_tile - pos_halves, std::integral_constant<int, T < 0 ? 6 : T>{}, S.n_dir);
                } else {  // the embedding (copied slot by slot) and the padding behind it
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) s_enc2[(64 * o + lane) * 4 + gg] = make_uint2(0u, 0u);
                    _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_enc2);
                    const int e_lo = 16 * half_tile - S.emb_base, e_hi = S.embedding_dim < e_lo + 16 ? S.embedding_dim : e_lo + 16;
                    for (int e = e_lo; e < e_hi; ++e)
                        tile_h[(64 * o + lane) * 16 + ((((e & 15) >> 2) ^ own_swz) << 2) + (e & 3)] = (_Float16)half_bits_to_float(emb[o][e]);
                }
            }
            __builtin_amdgcn_wave_barrier();  // LDS executes a wavefront's accesses in order; this only pins the compiler's order
        };

        // (the weights in LDS do not change between passes and the compiler knows: it would read the first layer's fragments and biases once,
        // before the pass loop, and hold them in 64 registers the hidden layers need -- the empty statement hides that the offset is the same;
        // an offset, not the pointers: hidden themselves they would stop being LDS pointers, and their loads become flat loads)
        int same_place = 0;
        asm volatile("" : "+s"(same_place));
        const half8 *w = s_frag + same_place;
        const float *b = s_bias + same_place;
        auto bias_tile = [&](int mt) __attribute__((always_inline)) -> f32x4 { return *reinterpret_cast<const f32x4 *>(b + 16 * mt + 4 * g); };

        // A layer with ReLU: the B fragments `bin` (KTIN K tiles x NT column tiles) in, the next layer's B fragments `nb` out.  One M tile (16
        // output features x COLS samples) at a time over the layer's K tiles, bias as the first MFMA's C operand; a finished PAIR of M tiles
        // is rounded and packed into the next layer's B fragment straight away (the C layout of two M tiles is the B layout of one K tile:
        // the permutation is in the weights), so that two accumulator tiles are live instead of MT, and the next M tile's weight fragments
        // and bias are read while the current one's MFMAs run.
        constexpr int KT = MT / 2;  // K tiles of a hidden layer
        auto dense_relu = [&](auto ktin_tag, const auto &bin, half8 (&nb)[KT][NT]) __attribute__((always_inline)) {
            constexpr int KTIN = decltype(ktin_tag)::value;
            half8 a[KTIN];
            f32x4 prev[NT], bv = bias_tile(0);
#pragma unroll
            for (int kk = 0; kk < KTIN; ++kk) a[kk] = w[kk * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                half8 an[KTIN];
                f32x4 bvn = bv;
                if (mt + 1 < MT) {
#pragma unroll
                    for (int kk = 0; kk < KTIN; ++kk) an[kk] = w[((mt + 1) * KTIN + kk) * 64 + lane];
                    bvn = bias_tile(mt + 1);
                }
                f32x4 cur[NT];
#pragma unroll
                for (int kk = 0; kk < KTIN; ++kk)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cur[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], bin[kk][nt], kk == 0 ? bv : cur[nt], 0, 0, 0);
                if (mt & 1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) nb[mt >> 1][nt] = relu_pack(prev[nt], cur[nt]);
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) prev[nt] = cur[nt];
                }
                if (mt + 1 < MT) {
#pragma unroll
                    for (int kk = 0; kk < KTIN; ++kk) a[kk] = an[kk];
                    bv = bvn;
                }
            }
            w += MT * KTIN * 64;
            b += 16 * MT;
        };

#ifdef MNV_MLP_CLOCKS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MLP_CLOCK(1);
        half8 bf[KT][NT];
        if constexpr (NKK0 > 0) {
            // ---- layer 0, the first layer's K tiles counted at compile time: ALL its B fragments are encoded first, half a K tile at a time through
            //      the tile (no accumulator is live yet: the encoding has the registers to itself), then it runs like any other layer
            half8 bf0[NKK0][NT];
            uint2 lo2[NT];
            static_for<2 * NKK0>([&](auto t_tag) __attribute__((always_inline)) {
                constexpr int T = decltype(t_tag)::value;
                encode_half(T, t_tag);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint2 part = s_enc2[(nt * 16 + col) * 4 + (g ^ read_swz(nt))];
                    if constexpr (T & 1) bf0[T >> 1][nt] = __builtin_bit_cast(half8, make_uint4(lo2[nt].x, lo2[nt].y, part.x, part.y));
                    else lo2[nt] = part;
                }
                __builtin_amdgcn_wave_barrier();
            });
            MLP_CLOCK(2);
            fetch_samples();
            dense_relu(std::integral_constant<int, NKK0>{}, bf0, bf);
            MLP_CLOCK(3);
        } else {
            // ---- layer 0 with any number of K tiles (more than four: large embeddings): K tile by K tile into ALL the layer's accumulators, the
            //      tile's MT weight fragments read before its encoding.  (Registers: the accumulators alone are half the budget of <8, 4, 8>,
            //      and its encoding spills around them -- the price of the rare shape.)
            f32x4 acc[MT][NT];
            half8 a8[MT];
            uint2 lo2[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {  // the accumulators start as the bias (as a C operand of the first K tile's MFMAs it would make them
                const f32x4 bv = bias_tile(mt);  // conditionally defined inside the loop: live, and spilled, from one pass to the next)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bv;
            }
#pragma unroll 1
            for (int half_tile = 0; half_tile < 2 * S.nkk0; ++half_tile) {
                if (!(half_tile & 1)) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a8[mt] = w[(mt * S.nkk0 + (half_tile >> 1)) * 64 + lane];
                }
                encode_half(half_tile, std::integral_constant<int, -1>{});
                if (!(half_tile & 1)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) lo2[nt] = s_enc2[(nt * 16 + col) * 4 + (g ^ read_swz(nt))];
                    __builtin_amdgcn_wave_barrier();
                } else {
                    half8 bt[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const uint2 hi2 = s_enc2[(nt * 16 + col) * 4 + (g ^ read_swz(nt))];
                        bt[nt] = __builtin_bit_cast(half8, make_uint4(lo2[nt].x, lo2[nt].y, hi2.x, hi2.y));
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[mt], bt[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            MLP_CLOCK(2);
            fetch_samples();
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[kk][nt] = relu_pack(acc[2 * kk][nt], acc[2 * kk + 1][nt]);
            w += MT * S.nkk0 * 64;
            b += 16 * MT;
            MLP_CLOCK(3);
        }

        // ---- hidden layers 1 .. hidden_layers - 1
        //      (two per trip, the B fragments going back and forth between two sets of registers: one per trip copied 64 registers a layer)
        {
            half8 nb[KT][NT];
            int layer = 1;
            for (; layer + 1 < S.hidden_layers; layer += 2) {
                dense_relu(std::integral_constant<int, KT>{}, bf, nb);
                dense_relu(std::integral_constant<int, KT>{}, nb, bf);
            }
            if (layer < S.hidden_layers) {
                dense_relu(std::integral_constant<int, KT>{}, bf, nb);
#pragma unroll
                for (int kk = 0; kk < KT; ++kk)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bf[kk][nt] = nb[kk][nt];
            }
        }

        MLP_CLOCK(4);
        take_over();  // the next tile's samples, before this tile's stores
        MLP_CLOCK(6);
        // ---- the output layer: lane (g, col) holds features 16 mt + 4 g + r of sample col + 16 nt; a 1 KB tile in LDS (the encode tile's
        //      place) turns that into the store's lane order, 16-byte slots swizzled so that neither side meets a bank twice
        f32x4 *s_out = reinterpret_cast<f32x4 *>(s_enc2);
        const int slot_w = col * 4 + (g ^ (col >> 2)), slot_r = (lane & ~3) | ((lane & 3) ^ ((lane >> 4) & 3));
        auto leave = [&](int mt, const f32x4 (&tile)[NT]) __attribute__((always_inline)) {
            static_assert(NT % 2 == 0, "column tiles leave in pairs");
#pragma unroll
            for (int nt0 = 0; nt0 < NT; nt0 += 2) {  // two column tiles per trip through LDS (the encode tile holds two 1 KB slots): half the round trips
                s_out[slot_w] = tile[nt0];
                s_out[64 + slot_w] = tile[nt0 + 1];
                __builtin_amdgcn_wave_barrier();
                const f32x4 v2[2] = {s_out[slot_r], s_out[64 + slot_r]};
                __builtin_amdgcn_wave_barrier();
                // a lane's four features: all inside the row (one 16-byte store), or the row's last one to three (ONE store of that width: the
                // width is the same for every lane that has a tail -- three 4-byte stores under three conditions were 0.1 ms of 1.29 for 29 outputs)
                const int f0 = 16 * mt + 4 * (lane & 3), here = S.out_dim - f0, tail = S.out_dim & 3;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nt = nt0 + h;
                    const f32x4 v = v2[h];
                    if (dst_row[nt] < 0 || here <= 0) continue;
                    float *out = L.results + (int64_t)dst_row[nt] * L.result_stride + f0;
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                    typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                    if (here >= 4) {
                        *reinterpret_cast<f4u *>(out) = v;
                    } else if (tail == 1) {
                        out[0] = v[0];
                    } else if (tail == 2) {
                        *reinterpret_cast<f2u *>(out) = f2u{v[0], v[1]};
                    } else {
                        *reinterpret_cast<f3u *>(out) = f3u{v[0], v[1], v[2]};
                    }
                }
            }
        };
        {   // tile mt's MFMAs are issued, then tile mt - 1 goes through the tile and out while they run
            f32x4 last[NT];
            half8 a[KT];
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) a[kk] = w[kk * 64 + lane];
            f32x4 bv = bias_tile(0);
#pragma unroll 1
            for (int mt = 0; mt < S.mt_out; ++mt) {
                f32x4 cur[NT];
#pragma unroll
                for (int kk = 0; kk < KT; ++kk)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cur[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], bf[kk][nt], kk == 0 ? bv : cur[nt], 0, 0, 0);
                if (mt + 1 < S.mt_out) {
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk) a[kk] = w[((mt + 1) * KT + kk) * 64 + lane];
                    bv = bias_tile(mt + 1);
                }
                if (mt > 0) leave(mt - 1, last);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) last[nt] = cur[nt];
            }
            leave(S.mt_out - 1, last);
        }
        MLP_CLOCK(5);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst_row[nt] = dst_row_next[nt];
        tile_cluster = cluster_next;
    }
#ifdef MNV_MLP_CLOCKS
    if (threadIdx.x == 0) {
        for (int i = 0; i < 14; ++i) atomicAdd(&L.clocks[i], clk_[i]);
        atomicAdd(&L.clocks[14], __builtin_readcyclecounter() - clk_begin_);  // shader-clock cycles and 100 MHz ticks of the workgroup's life:
        atomicAdd(&L.clocks[15], wall_clock64() - wall_begin_);               // their ratio is the clock the kernel ran at
    }
#endif
}

// ---------------------------------------------------------------- host side

static int fill_shape(MlpShape &S, const mnv_mlp_desc *d) {
    if (!d) return set_error(MNV_E_INVALID, "null MLP description");
    if (d->n_clusters < 1 || d->n_clusters > kMaxClusters) return set_error(MNV_E_INVALID, "n_clusters must be 1 .. 1024");
    if (d->pos_octaves < 0 || d->pos_octaves > 16 || d->dir_octaves < 0 || d->dir_octaves > 16)
        return set_error(MNV_E_INVALID, "octaves must be 0 .. 16");
    if (d->hidden_width != 64 && d->hidden_width != 128) return set_error(MNV_E_UNSUPPORTED, "hidden_width must be 64 or 128");
    if (d->hidden_layers < 1 || d->hidden_layers > 16) return set_error(MNV_E_INVALID, "hidden_layers must be 1 .. 16");
    if (d->out_dim < 1 || d->out_dim > d->hidden_width) return set_error(MNV_E_UNSUPPORTED, "out_dim must be 1 .. hidden_width");
    if (d->n_embeddings < 0 || (d->n_embeddings > 0 && (d->embedding_dim < 1 || d->embedding_dim > 64)))
        return set_error(MNV_E_INVALID, "embedding_dim must be 1 .. 64 when n_embeddings > 0");
    std::memset(&S, 0, sizeof(S));
    S.n_clusters = d->n_clusters;
    S.pos_octaves = d->pos_octaves;
    S.dir_octaves = d->dir_octaves;
    S.need_viewdir = d->need_viewdir ? 1 : 0;
    S.n_embeddings = d->n_embeddings;
    S.embedding_dim = d->n_embeddings > 0 ? d->embedding_dim : 0;
    S.hidden_width = d->hidden_width;
    S.hidden_layers = d->hidden_layers;
    S.out_dim = d->out_dim;
    S.n_pos = 3 + 6 * S.pos_octaves;
    S.n_dir = S.need_viewdir ? 3 + 6 * S.dir_octaves : 0;
    S.in_dim = S.n_pos + S.n_dir + S.embedding_dim;
    S.dir_base = (S.n_pos + 15) & ~15;
    S.emb_base = S.dir_base + ((S.n_dir + 15) & ~15);
    S.k_slots = S.emb_base + S.embedding_dim;
    S.nkk0 = (S.k_slots + 31) / 32;
    S.mt_hidden = S.hidden_width / 16;
    S.mt_out = (S.out_dim + 15) / 16;
    const int nkk_h = S.hidden_width / 32;
    S.frag_halfs = 512 * (S.mt_hidden * S.nkk0 + (S.hidden_layers - 1) * S.mt_hidden * nkk_h + S.mt_out * nkk_h);
    S.bias_floats = 16 * (S.mt_hidden * S.hidden_layers + S.mt_out);
    std::memcpy(S.center, d->center, sizeof(S.center));
    std::memcpy(S.inv_extent, d->inv_extent, sizeof(S.inv_extent));
    return MNV_OK;
}

static size_t param_count(const MlpShape &S) {
    const size_t w = (size_t)S.hidden_width;
    return w * S.in_dim + w + (size_t)(S.hidden_layers - 1) * (w * w + w) + (size_t)S.out_dim * w + S.out_dim +
           (size_t)S.n_embeddings * S.embedding_dim;
}

static float half_to_float_host(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal
            int e = -1;
            uint32_t m = man;
            do {
                ++e;
                m <<= 1;
            } while (!(m & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

size_t mnv_mlp_param_count(const mnv_mlp_desc *desc) {
    MlpShape S;
    if (fill_shape(S, desc)) return 0;
    return param_count(S);
}

int mnv_mlp_create(const mnv_mlp_desc *desc, const uint16_t *params, size_t n_halfs, void *hip_stream, mnv_mlp **out) {
    if (!out || !params) return set_error(MNV_E_INVALID, "null argument");
    *out = nullptr;
    MlpShape S;
    int rc = fill_shape(S, desc);
    if (rc) return rc;
    const size_t per_cluster = param_count(S);
    if (n_halfs != per_cluster * S.n_clusters) return set_error(MNV_E_INVALID, "parameter blob has the wrong size (see mnv_mlp_param_count)");
    const size_t lds_bytes = (size_t)S.frag_halfs * 2 + (size_t)S.bias_floats * 4 + 4 * 16 * 64 * 4;
    if (lds_bytes > 160 * 1024) return set_error(MNV_E_UNSUPPORTED, "network does not fit the 160 KB LDS of a CU (weights of all layers + 16 KB of encode tiles)");
    int dev = 0;
    hipDeviceProp_t prop;
    if ((rc = check_hip(hipGetDevice(&dev), "hipGetDevice"))) return rc;
    if ((rc = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties"))) return rc;

    // permute the row-major layers into MFMA A fragments: frag[(mt * nkk + kk) * 64 + lane][e] = W[16 mt + (lane & 15)][slot_feature(kk, lane >> 4, e)]
    std::vector<uint16_t> frags((size_t)S.n_clusters * S.frag_halfs, 0);
    std::vector<float> biases((size_t)S.n_clusters * S.bias_floats, 0.f);
    std::vector<uint16_t> emb((size_t)S.n_clusters * S.n_embeddings * S.embedding_dim, 0);
    const int W = S.hidden_width;
    for (int c = 0; c < S.n_clusters; ++c) {
        const uint16_t *src = params + (size_t)c * per_cluster;
        uint16_t *f = frags.data() + (size_t)c * S.frag_halfs;
        float *b = biases.data() + (size_t)c * S.bias_floats;
        for (int layer = 0; layer <= S.hidden_layers; ++layer) {
            const int in = layer == 0 ? S.in_dim : W, outd = layer == S.hidden_layers ? S.out_dim : W;
            const int nkk = layer == 0 ? S.nkk0 : W / 32, n_mt = layer == S.hidden_layers ? S.mt_out : S.mt_hidden;
            for (int mt = 0; mt < n_mt; ++mt)
                for (int kk = 0; kk < nkk; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int row = 16 * mt + (lane & 15), slot = slot_feature(kk, lane >> 4, e);
                            const int k = layer == 0 ? feature_of_slot(S, slot) : (slot < in ? slot : -1);  // hidden layers: slot = feature
                            f[((size_t)(mt * nkk + kk) * 64 + lane) * 8 + e] = (row < outd && k >= 0) ? src[(size_t)row * in + k] : 0;
                        }
            for (int i = 0; i < outd; ++i) b[i] = half_to_float_host(src[(size_t)outd * in + i]);
            f += (size_t)n_mt * nkk * 512;
            b += 16 * n_mt;
            src += (size_t)outd * in + outd;
        }
        if (!emb.empty()) std::memcpy(emb.data() + (size_t)c * S.n_embeddings * S.embedding_dim, src, (size_t)S.n_embeddings * S.embedding_dim * 2);
    }

    mnv_mlp *m = new mnv_mlp();
    m->shape = S;
    m->num_cus = prop.multiProcessorCount;
    hipStream_t stream = (hipStream_t)hip_stream;
    auto fail = [&](int code) {
        mnv_mlp_destroy(m);
        return code;
    };
    // + 8 fragments / 64 floats of slack: the fused guided kernel fetches every layer as 8 fragments and 4 bias tiles
    if ((rc = check_hip(hipMalloc((void **)&m->frags, frags.size() * 2 + 8 * 1024), "hipMalloc(mlp fragments)"))) return fail(rc);
    if ((rc = check_hip(hipMalloc((void **)&m->biases, biases.size() * 4 + 256), "hipMalloc(mlp biases)"))) return fail(rc);
    if ((rc = check_hip(hipMemsetAsync(m->frags, 0, frags.size() * 2 + 8 * 1024, stream), "clear"))) return fail(rc);
    if ((rc = check_hip(hipMemsetAsync(m->biases, 0, biases.size() * 4 + 256, stream), "clear"))) return fail(rc);
    if ((rc = check_hip(hipMemcpyAsync(m->frags, frags.data(), frags.size() * 2, hipMemcpyHostToDevice, stream), "upload"))) return fail(rc);
    if ((rc = check_hip(hipMemcpyAsync(m->biases, biases.data(), biases.size() * 4, hipMemcpyHostToDevice, stream), "upload"))) return fail(rc);
    if (!emb.empty()) {
        if ((rc = check_hip(hipMalloc((void **)&m->embeddings, emb.size() * 2), "hipMalloc(mlp embeddings)"))) return fail(rc);
        if ((rc = check_hip(hipMemcpyAsync(m->embeddings, emb.data(), emb.size() * 2, hipMemcpyHostToDevice, stream), "upload"))) return fail(rc);
    }
    if ((rc = check_hip(hipStreamSynchronize(stream), "mlp upload"))) return fail(rc);  // the staging vectors die here
    *out = m;
    return MNV_OK;
}

void mnv_mlp_destroy(mnv_mlp *m) {
    if (!m) return;
    if (m->frags) (void)hipFree(m->frags);
    if (m->biases) (void)hipFree(m->biases);
    if (m->embeddings) (void)hipFree(m->embeddings);
    if (m->scratch) (void)hipFree(m->scratch);
    delete m;
}

int mnv_query_submodules(mnv_mlp *m, const int16_t *cluster_indices, const float *samples, int32_t samples_stride,
                         int64_t n, float *results, int32_t result_stride, void *hip_stream) {
    if (!m || !cluster_indices || !samples || !results) return set_error(MNV_E_INVALID, "null argument");
    const MlpShape &S = m->shape;
    const int need_cols = 3 + (S.need_viewdir ? 3 : 0) + (S.n_embeddings > 0 ? 1 : 0);
    if (n < 0 || n > 0x7fffffff || samples_stride < need_cols || result_stride < S.out_dim)
        return set_error(MNV_E_INVALID, "invalid sample / result shapes");
    if (n == 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    // A tile is one pass of a workgroup (256 or 512 rows of one cluster); the persistent workgroups of mlp_forward_kernel share the tiles evenly.
    const int64_t cus = m->num_cus > 0 ? m->num_cus : 256;
    const int32_t rows_per_tile = S.hidden_width == 64 ? kRowsPerPass : 2 * kRowsPerPass;  // <4, 4, 4>: 4 x 64 rows; <8, 4, 8>: 8 x 64 rows
    const int64_t max_tiles = n / rows_per_tile + S.n_clusters + 1;
    const size_t table = (size_t)(kMaxClusters + 64) * 4;
    const size_t o_counts = 0, o_start = o_counts + table, o_cursor = o_start + table, o_tiles = o_cursor + table, o_order = o_tiles + table;
    const size_t need = o_order + (size_t)n * 4;
    int rc;
    if (m->scratch_bytes < need) {
        if (m->scratch) {
            if ((rc = check_hip(hipStreamSynchronize(stream), "mlp scratch"))) return rc;
            (void)hipFree(m->scratch);
            m->scratch = nullptr;
            m->scratch_bytes = 0;
        }
        if ((rc = check_hip(hipMalloc((void **)&m->scratch, need + need / 4), "hipMalloc(mlp scratch)"))) return rc;
        m->scratch_bytes = need + need / 4;
    }
    int32_t *counts = reinterpret_cast<int32_t *>(m->scratch + o_counts), *seg_start = reinterpret_cast<int32_t *>(m->scratch + o_start);
    int32_t *cursor = reinterpret_cast<int32_t *>(m->scratch + o_cursor);
    int32_t *tile_start = reinterpret_cast<int32_t *>(m->scratch + o_tiles), *order = reinterpret_cast<int32_t *>(m->scratch + o_order);

    if ((rc = check_hip(hipMemsetAsync(counts, 0, kMaxClusters * 4, stream), "memset"))) return rc;
    // sort blocks: as many chunks of 2048 rows each as leaves about two blocks per compute unit (at most 16 chunks)
    const int64_t n_chunks = (n + kSortRows - 1) / kSortRows;
    const int32_t chunks = (int32_t)std::min<int64_t>(16, std::max<int64_t>(1, n_chunks / (2 * cus)));
    const unsigned nb = (unsigned)((n_chunks + chunks - 1) / chunks);
    hipLaunchKernelGGL(mlp_histogram, dim3(nb), dim3(256), 0, stream, cluster_indices, n, S.n_clusters, chunks, counts, results, result_stride, S.out_dim);
    hipLaunchKernelGGL(mlp_plan, dim3(1), dim3(256), 0, stream, counts, S.n_clusters, seg_start, cursor, tile_start, rows_per_tile);
    hipLaunchKernelGGL(mlp_scatter, dim3(nb), dim3(256), 0, stream, cluster_indices, n, S.n_clusters, chunks, cursor, order);

    MlpLaunch L;
    L.S = S;
    L.frags = m->frags;
    L.biases = m->biases;
    L.embeddings = m->embeddings;
    L.samples = samples;
    L.samples_stride = samples_stride;
    L.results = results;
    L.result_stride = result_stride;
    L.order = order;
    L.seg_start = seg_start;
    L.tile_start = tile_start;
#ifdef MNV_MLP_CLOCKS
    static unsigned long long *clocks = nullptr;
    if (!clocks) (void)hipMalloc((void **)&clocks, 128);
    (void)hipMemsetAsync(clocks, 0, 128, stream);
    L.clocks = clocks;
#endif
    const size_t lds_bytes = (size_t)S.frag_halfs * 2 + (size_t)S.bias_floats * 4 + 4 * 16 * 64 * 4;  // + the encode tiles: 256 samples x 64 bytes
    // one kernel per width and per number of K tiles of the first layer (1 .. 4 at compile time; 0: any number -- the large embeddings)
    auto launch = [&](auto kern, unsigned threads) -> int {
        int rc2 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes), "lds attr");
        if (rc2) return rc2;
        // as many workgroups as the device holds at once (registers and LDS decide: one per compute unit for the 128-wide networks)
        if (m->blocks_per_cu <= 0) {
            int per_cu = 0;
            rc2 = check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), (int)threads, lds_bytes), "occupancy");
            if (rc2) return rc2;
            m->blocks_per_cu = std::max(per_cu, 1);
        }
        const int64_t blocks = std::min<int64_t>(max_tiles, cus * m->blocks_per_cu);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds_bytes, stream, L);
        return MNV_OK;
    };
    if (S.hidden_width == 64) {
        switch (S.nkk0) {
            case 1: rc = launch(mlp_forward_kernel<4, 4, 4, 1>, 256); break;
            case 2: rc = launch(mlp_forward_kernel<4, 4, 4, 2>, 256); break;
            case 3: rc = launch(mlp_forward_kernel<4, 4, 4, 3>, 256); break;
            case 4: rc = launch(mlp_forward_kernel<4, 4, 4, 4>, 256); break;
            default: rc = launch(mlp_forward_kernel<4, 4, 4, 0>, 256); break;
        }
    } else {  // (until round 5 <8, 2, 8>: 32 columns per wavefront, two MFMAs per fragment read)
        switch (S.nkk0) {
            case 1: rc = launch(mlp_forward_kernel<8, 4, 8, 1>, 512); break;
            case 2: rc = launch(mlp_forward_kernel<8, 4, 8, 2>, 512); break;
            case 3: rc = launch(mlp_forward_kernel<8, 4, 8, 3>, 512); break;
            case 4: rc = launch(mlp_forward_kernel<8, 4, 8, 4>, 512); break;
            default: rc = launch(mlp_forward_kernel<8, 4, 8, 0>, 512); break;
        }
    }
    if (rc) return rc;
#ifdef MNV_MLP_CLOCKS
    {
        unsigned long long h[16];
        (void)hipMemcpyAsync(h, clocks, 128, hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        unsigned long long sum = 0;
        for (int i = 0; i < 14; ++i) sum += h[i];
        fprintf(stderr, "mlp clocks: shader clock %.3f GHz over the workgroups' lives\n", (double)h[14] / (double)h[15] * 0.1);
        static const char *const names[7] = {"stage weights", "top of a tile (nothing should wait)", "layer 0: fragment reads + encode", "layer 0: read back + MFMAs", "hidden layers", "output layer + stores", "next tile's samples taken over"};
        for (int i = 0; i < 7; ++i) fprintf(stderr, "mlp clocks: %-28s %5.1f %%\n", names[i], 100.0 * (double)h[i] / (double)sum);
    }
#endif
    return check_hip(hipGetLastError(), "mlp_forward_kernel");
}

}  // extern "C"
