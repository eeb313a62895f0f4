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
// Execution: counting sort of the rows by cluster (histogram + scan + scatter), then one workgroup per 256 rows
// of one cluster.  The whole network runs in registers: with the weights as the MFMA A operand and the samples
// as the B operand (D^T = W X^T), the C layout of one layer (lane: sample l & 15, features 4 (l >> 4) + r) is
// already a valid B layout for the next layer once the K index is permuted -- and the permutation is baked into
// the weight fragments at upload time, so nothing moves between layers but a float -> half convert.
// v_mfma_f32_16x16x32_f16, fragments staged in LDS once per workgroup.

#include <algorithm>
#include <cstring>
#include <type_traits>
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
    const int32_t *tile_start;  // [n_clusters + 1]: first tile (workgroup) of each cluster; [n_clusters] = number of tiles
    int32_t rows_per_block;     // rows of one cluster a workgroup takes (a multiple of kRowsPerPass, at most kRowsPerBlock)
};

// ---------------------------------------------------------------- counting sort by cluster

// Rows per block of the two sort passes; each thread handles kSortItems rows.
constexpr int kSortItems = 8;
constexpr int kSortRows = 256 * kSortItems;

// Wavefront-aggregated add of 1 to s_bins[c] for every active lane: one LDS atomic per distinct bin in the
// wavefront instead of one per lane (neighbouring samples mostly share a cluster).  Returns the lane's rank
// among the block's rows of its bin.
__device__ inline int aggregated_rank(int32_t *s_bins, int c, bool valid) {
    int rank = 0;
    unsigned long long todo = __ballot(valid);
    const unsigned lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c0 = __shfl(c, leader);
        const unsigned long long same = __ballot(valid && c == c0);
        int base = 0;
        if ((int)lane == leader) base = atomicAdd(&s_bins[c0], __popcll(same));
        base = __shfl(base, leader);
        if (valid && c == c0) rank = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return rank;
}

__global__ __launch_bounds__(256) void mlp_histogram(const int16_t *__restrict__ cluster, int64_t n, int32_t n_clusters,
                                                     int32_t *__restrict__ counts, float *__restrict__ results, int32_t result_stride,
                                                     int32_t out_dim) {
    __shared__ int32_t s_bins[kMaxClusters];
    for (int c = threadIdx.x; c < n_clusters; c += 256) s_bins[c] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortRows;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        const int c = i < n ? cluster[i] : -1;
        const bool valid = c >= 0 && c < n_clusters;
        (void)aggregated_rank(s_bins, c, valid);
        if (i < n && !valid)
            for (int o = 0; o < out_dim; ++o) results[i * result_stride + o] = 0.f;  // no sub-module: zeros
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_clusters; c += 256)
        if (s_bins[c]) atomicAdd(&counts[c], s_bins[c]);
}

// one block: exclusive scans of the counts -> segment starts and first tile of every cluster
__global__ void mlp_plan(const int32_t *__restrict__ counts, int32_t n_clusters, int32_t *__restrict__ seg_start,
                         int32_t *__restrict__ cursor, int32_t *__restrict__ tile_start, int32_t rows_per_block) {
    __shared__ int32_t s_start[kMaxClusters + 1], s_tile[kMaxClusters + 1];
    if (threadIdx.x == 0) {
        int32_t a = 0, t = 0;
        for (int c = 0; c < n_clusters; ++c) {
            s_start[c] = a;
            s_tile[c] = t;
            a += counts[c];
            t += (counts[c] + rows_per_block - 1) / rows_per_block;
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

__global__ __launch_bounds__(256) void mlp_scatter(const int16_t *__restrict__ cluster, int64_t n, int32_t n_clusters,
                                                   int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    __shared__ int32_t s_bins[kMaxClusters];
    for (int c = threadIdx.x; c < n_clusters; c += 256) s_bins[c] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortRows;
    int rank[kSortItems], bin[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        const int c = i < n ? cluster[i] : -1;
        const bool valid = c >= 0 && c < n_clusters;
        bin[k] = valid ? c : -1;
        rank[k] = aggregated_rank(s_bins, c, valid);
    }
    __syncthreads();
    // one global reservation per bin and block; s_bins becomes the block's base position
    for (int c = threadIdx.x; c < n_clusters; c += 256)
        if (s_bins[c]) s_bins[c] = atomicAdd(&cursor[c], s_bins[c]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k)
        if (bin[k] >= 0) order[s_bins[bin[k]] + rank[k]] = (int32_t)(base + k * 256 + threadIdx.x);
}

// ---------------------------------------------------------------- the network

// One workgroup = WAVES wavefronts x (16 * NT) rows per pass = 256 rows.  <4, 4, 4>: 64-wide networks, four column tiles per
// wavefront.  <8, 2, 8>: 128-wide networks -- their weights fill most of a CU's LDS (131 KB for 4 hidden layers), so one workgroup
// per CU is all that fits: eight wavefronts with two column tiles each share the weights (two per SIMD, 64 accumulator + 32 operand
// registers) instead of four wavefronts that hold 128 + 64 and leave every SIMD with one wavefront and nothing to overlap its
// LDS and matrix-pipe latencies with.
// <8, 4, 8> (round 5): the 128-wide network with 64 columns per wavefront -- every weight fragment read from LDS (1 KB, 8 cycles of the CU's LDS
// port) feeds FOUR 16-cycle MFMAs instead of two, so the port is busy half as long as the matrix pipes instead of as long; 128 accumulator
// + 64 operand registers at two wavefronts per SIMD (256-register budget), 512 rows per pass.  Its encode tiles hold HALF a K tile (16
// features: 2 KB per wavefront, written and read twice per K tile) -- with whole tiles the 131 KB of weights + 32 KB would not fit a CU.
template <int MT, int NT, int WAVES>  // hidden width = 16 * MT
__global__ __launch_bounds__(64 * WAVES, (MT * NT >= 32 ? 1 : 2)) void mlp_forward_kernel(const MlpLaunch L) {
    constexpr int ROWS = WAVES * 16 * NT;  // rows of one pass of the workgroup
    constexpr bool HALF = NT * 16 == 64 && WAVES == 8;  // half-K-tile encode tiles
    static_assert(ROWS % kRowsPerPass == 0, "a pass is a multiple of 256 rows");
    constexpr int COLS = 16 * NT;  // samples (MFMA columns) of one wavefront
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const MlpShape &S = L.S;
    if ((int)blockIdx.x >= L.tile_start[S.n_clusters]) return;
    // the cluster whose tile range holds this block: last c with tile_start[c] <= blockIdx.x (uniform: scalar loads)
    int lo = 0, hi = S.n_clusters;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (L.tile_start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const int cluster = lo;
    const int block_first = L.seg_start[cluster] + ((int)blockIdx.x - L.tile_start[cluster]) * L.rows_per_block;
    const int block_rows = min(L.rows_per_block, L.seg_start[cluster + 1] - block_first);

    // weights and biases of this cluster -> LDS
    const half8 *s_frag = reinterpret_cast<const half8 *>(lds);
    const float *s_bias = reinterpret_cast<const float *>(lds + (size_t)S.frag_halfs * 2);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(L.frags + (size_t)cluster * S.frag_halfs);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < S.frag_halfs / 8; i += blockDim.x) dst[i] = src[i];
        const float *bsrc = L.biases + (size_t)cluster * S.bias_floats;
        float *bdst = reinterpret_cast<float *>(lds + (size_t)S.frag_halfs * 2);
        for (int i = threadIdx.x; i < S.bias_floats; i += blockDim.x) bdst[i] = bsrc[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    // the wavefront's encode tile: one K tile (32 features) of its COLS samples, a sample's 64 bytes in operand order (the 8 halfs of
    // lane group gg at byte 16 gg: features 4 gg .. 4 gg + 3 and 16 + 4 gg .. 16 + 4 gg + 3): written as four 16-byte stores by the
    // lane that owns the sample, read as one 16-byte load per column tile by the lane that feeds the matrix pipe
    uint4 *s_enc = reinterpret_cast<uint4 *>(lds + (size_t)S.frag_halfs * 2 + (size_t)S.bias_floats * 4) + wave * (COLS * (HALF ? 2 : 4));
    [[maybe_unused]] uint2 *s_enc2 = reinterpret_cast<uint2 *>(s_enc);  // HALF: a sample's 32 bytes = four lane groups x 8 bytes (features 4 gg .. 4 gg + 3 of the half)
    const int emb_base = S.n_pos + S.n_dir;

    for (int pass_first = 0; pass_first < block_rows; pass_first += ROWS) {
        const int first = block_first + pass_first, rows = min(ROWS, block_rows - pass_first);
        // Two roles per lane.  Encoding: lane j < COLS owns sample j of the wavefront.  Matrix operand / output: lane (g, col) serves the
        // samples col + 16 nt.
        int32_t src_row[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int local = wave * COLS + nt * 16 + col;
            src_row[nt] = local < rows ? L.order[first + local] : -1;
        }
        float p[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        const uint16_t *emb = nullptr;
        if (lane < COLS) {
            const int local = wave * COLS + lane;
            const float *x = L.samples + (int64_t)L.order[first + (local < rows ? local : 0)] * L.samples_stride;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                p[i] = (x[i] - S.center[i]) * S.inv_extent[i];
                d[i] = S.need_viewdir ? x[3 + i] : 0.f;
            }
            if (S.n_embeddings > 0) {
                int idx = (int)x[S.need_viewdir ? 6 : 3];
                idx = idx < 0 ? 0 : (idx >= S.n_embeddings ? S.n_embeddings - 1 : idx);
                emb = L.embeddings + ((size_t)cluster * S.n_embeddings + idx) * S.embedding_dim;
            }
        }
        // Encoding of K tile kk (32 features) of this lane's sample.  Position features -- the bulk -- have compile-time places in the first
        // three K tiles (coordinate, then per octave three phase-0 and three phase-1/4 triangle waves), guarded per octave by the
        // network's octave count (wave-uniform) and stored as four 16-byte words; whatever else the tile holds (view direction,
        // embedding, position features past the 96th) is decoded at run time and stored half by half over the zeros.
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        typedef float float2v __attribute__((ext_vector_type(2)));
        auto pos_feature = [&](auto f_tag) __attribute__((always_inline)) -> float {
            constexpr int f = decltype(f_tag)::value;
            if constexpr (f < 3) {
                return p[f];
            } else {
                constexpr int k = (f - 3) / 6, r = (f - 3) % 6, i = r % 3;
                const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
                return k < S.pos_octaves ? tri_wave(p[i] * scale + (r >= 3 ? 0.25f : 0.f)) : 0.f;
            }
        };
        auto pair = [&](auto f_tag) __attribute__((always_inline)) -> uint32_t {
            constexpr int f = decltype(f_tag)::value;
            const float2v pr = {pos_feature(std::integral_constant<int, f>{}), pos_feature(std::integral_constant<int, f + 1>{})};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, half2v));
        };
        auto fast_group = [&](auto kk_tag, auto gg_tag) __attribute__((always_inline)) {
            constexpr int kk = decltype(kk_tag)::value, gg = decltype(gg_tag)::value, f0 = 32 * kk + 4 * gg;
            s_enc[lane * 4 + gg] = make_uint4(pair(std::integral_constant<int, f0>{}), pair(std::integral_constant<int, f0 + 2>{}),
                                              pair(std::integral_constant<int, f0 + 16>{}), pair(std::integral_constant<int, f0 + 18>{}));
        };
        auto fast_tile = [&](auto kk_tag) __attribute__((always_inline)) {
            fast_group(kk_tag, std::integral_constant<int, 0>{});
            fast_group(kk_tag, std::integral_constant<int, 1>{});
            fast_group(kk_tag, std::integral_constant<int, 2>{});
            fast_group(kk_tag, std::integral_constant<int, 3>{});
        };
        auto encode_tile = [&](int kk) __attribute__((always_inline)) {  // kk is wave-uniform
            if (lane < COLS) {
                int f_lo;  // the first feature of the tile the wide stores did not produce
                if (kk == 0) {
                    fast_tile(std::integral_constant<int, 0>{});
                    f_lo = S.n_pos;
                } else if (kk == 1) {
                    fast_tile(std::integral_constant<int, 1>{});
                    f_lo = S.n_pos;
                } else if (kk == 2) {
                    fast_tile(std::integral_constant<int, 2>{});
                    f_lo = S.n_pos;
                } else {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) s_enc[lane * 4 + gg] = make_uint4(0u, 0u, 0u, 0u);
                    f_lo = 32 * kk;
                }
                f_lo = f_lo > 32 * kk ? f_lo : 32 * kk;
                const int f_hi = S.in_dim < 32 * kk + 32 ? S.in_dim : 32 * kk + 32;
                _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_enc);
                for (int f = f_lo; f < f_hi; ++f) {  // wave-uniform bounds
                    const float v = (f >= emb_base) ? half_bits_to_float(emb[f - emb_base]) : encode_feature(S, f, p, d);
                    const int r = f & 31;
                    tile_h[lane * 32 + ((r & 15) >> 2) * 8 + (r >> 4) * 4 + (r & 3)] = (_Float16)v;
                }
            }
            __builtin_amdgcn_wave_barrier();  // LDS executes a wavefront's accesses in order; this only pins the compiler's order
        };

        // HALF: the same features, one half of the K tile (16 features) at a time into a tile half the size
        [[maybe_unused]] auto fast_half_group = [&](auto kk_tag, auto h_tag, auto gg_tag) __attribute__((always_inline)) {
            constexpr int kk = decltype(kk_tag)::value, h = decltype(h_tag)::value, gg = decltype(gg_tag)::value, f0 = 32 * kk + 16 * h + 4 * gg;
            s_enc2[lane * 4 + gg] = make_uint2(pair(std::integral_constant<int, f0>{}), pair(std::integral_constant<int, f0 + 2>{}));
        };
        [[maybe_unused]] auto fast_half = [&](auto kk_tag, auto h_tag) __attribute__((always_inline)) {
            fast_half_group(kk_tag, h_tag, std::integral_constant<int, 0>{});
            fast_half_group(kk_tag, h_tag, std::integral_constant<int, 1>{});
            fast_half_group(kk_tag, h_tag, std::integral_constant<int, 2>{});
            fast_half_group(kk_tag, h_tag, std::integral_constant<int, 3>{});
        };
        [[maybe_unused]] auto encode_half = [&](int kk, auto h_tag) __attribute__((always_inline)) {  // kk is wave-uniform
            constexpr int h = decltype(h_tag)::value;
            if (lane < COLS) {
                int f_lo;
                if (kk == 0) {
                    fast_half(std::integral_constant<int, 0>{}, h_tag);
                    f_lo = S.n_pos;
                } else if (kk == 1) {
                    fast_half(std::integral_constant<int, 1>{}, h_tag);
                    f_lo = S.n_pos;
                } else if (kk == 2) {
                    fast_half(std::integral_constant<int, 2>{}, h_tag);
                    f_lo = S.n_pos;
                } else {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) s_enc2[lane * 4 + gg] = make_uint2(0u, 0u);
                    f_lo = 32 * kk;
                }
                const int h_lo = 32 * kk + 16 * h, h_hi = h_lo + 16;
                f_lo = f_lo > h_lo ? f_lo : h_lo;
                const int f_hi = S.in_dim < h_hi ? S.in_dim : h_hi;
                _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_enc2);
                for (int f = f_lo; f < f_hi; ++f) {  // wave-uniform bounds
                    const float v = (f >= emb_base) ? half_bits_to_float(emb[f - emb_base]) : encode_feature(S, f, p, d);
                    const int r = f & 15;
                    tile_h[lane * 16 + (r >> 2) * 4 + (r & 3)] = (_Float16)v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        f32x4 acc[MT][NT];
        const half8 *w = s_frag;
        const float *b = s_bias;
        auto bias_tile = [&](int mt) __attribute__((always_inline)) -> f32x4 { return *reinterpret_cast<const f32x4 *>(b + 16 * mt + 4 * g); };

        // ---- layer 0: B fragments are computed from the raw sample on the fly, one K tile at a time; the bias enters as the C operand
        //      of the first K tile's MFMAs, and the next M tile's fragment travels under the current one's MFMAs
        for (int kk = 0; kk < S.nkk0; ++kk) {
            half8 bf[NT];
            if constexpr (HALF) {
                uint2 lo2[NT];
                encode_half(kk, std::integral_constant<int, 0>{});
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) lo2[nt] = s_enc2[(nt * 16 + col) * 4 + g];
                __builtin_amdgcn_wave_barrier();
                encode_half(kk, std::integral_constant<int, 1>{});
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint2 hi2 = s_enc2[(nt * 16 + col) * 4 + g];
                    bf[nt] = __builtin_bit_cast(half8, make_uint4(lo2[nt].x, lo2[nt].y, hi2.x, hi2.y));
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                encode_tile(kk);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = __builtin_bit_cast(half8, s_enc[(nt * 16 + col) * 4 + g]);
                __builtin_amdgcn_wave_barrier();
            }
            half8 a = w[kk * 64 + lane];
            if (kk == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half8 a_now = a;
                    if (mt + 1 < MT) a = w[((mt + 1) * S.nkk0 + kk) * 64 + lane];
                    const f32x4 bv = bias_tile(mt);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_now, bf[nt], bv, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half8 a_now = a;
                    if (mt + 1 < MT) a = w[((mt + 1) * S.nkk0 + kk) * 64 + lane];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_now, bf[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        w += MT * S.nkk0 * 64;
        b += 16 * MT;

        // ---- hidden layers 1 .. hidden_layers-1 and the output layer: B fragments are the previous accumulators
        for (int layer = 1; layer <= S.hidden_layers; ++layer) {
            const int n_mt = layer < S.hidden_layers ? MT : S.mt_out;
            half8 bf[MT / 2][NT];
#pragma unroll
            for (int kk = 0; kk < MT / 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[kk][nt] = relu_pack(acc[2 * kk][nt], acc[2 * kk + 1][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < n_mt) {
                    half8 a = w[(mt * (MT / 2)) * 64 + lane];
#pragma unroll
                    for (int kk = 0; kk < MT / 2; ++kk) {
                        const half8 a_now = a;
                        if (kk + 1 < MT / 2) a = w[(mt * (MT / 2) + kk + 1) * 64 + lane];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (kk == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_now, bf[0][nt], bias_tile(mt), 0, 0, 0);
                            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_now, bf[kk][nt], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
            w += n_mt * (MT / 2) * 64;
            b += 16 * n_mt;
        }

        // ---- store: lane holds features 16 mt + 4 g + r of row src_row[nt]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (src_row[nt] < 0) continue;
            float *out = L.results + (int64_t)src_row[nt] * L.result_stride;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt >= S.mt_out) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * mt + 4 * g + r;
                    if (f < S.out_dim) out[f] = acc[mt][nt][r];
                }
            }
        }
    }
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
    S.nkk0 = (S.in_dim + 31) / 32;
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
                            const int row = 16 * mt + (lane & 15), k = slot_feature(kk, lane >> 4, e);
                            f[((size_t)(mt * nkk + kk) * 64 + lane) * 8 + e] = (row < outd && k < in) ? src[(size_t)row * in + k] : 0;
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
    // A workgroup stages its cluster's weights once and runs up to kPasses passes of 256 rows under them.  A small batch (the 262 k rows of a
    // refinement step's 4096 splits: 134 workgroups of 2048 rows on 256 CUs, 51 us) takes fewer passes per workgroup, so that the device
    // holds about four workgroups per CU.
    const int64_t cus = m->num_cus > 0 ? m->num_cus : 256;
    const int rows_per_pass = S.hidden_width == 64 ? kRowsPerPass : 2 * kRowsPerPass;  // <4, 4, 4>: 4 x 64 rows; <8, 4, 8>: 8 x 64 rows
    int passes = kRowsPerBlock / rows_per_pass;
    while (passes > 1 && n / ((int64_t)rows_per_pass * passes) < cus * 4) passes >>= 1;
    const int32_t rows_per_block = rows_per_pass * passes;
    const int64_t max_tiles = n / rows_per_block + S.n_clusters + 1;
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
    const unsigned nb = (unsigned)((n + kSortRows - 1) / kSortRows);
    hipLaunchKernelGGL(mlp_histogram, dim3(nb), dim3(256), 0, stream, cluster_indices, n, S.n_clusters, counts, results, result_stride, S.out_dim);
    hipLaunchKernelGGL(mlp_plan, dim3(1), dim3(256), 0, stream, counts, S.n_clusters, seg_start, cursor, tile_start, rows_per_block);
    hipLaunchKernelGGL(mlp_scatter, dim3(nb), dim3(256), 0, stream, cluster_indices, n, S.n_clusters, cursor, order);

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
    L.rows_per_block = rows_per_block;
    const size_t lds_bytes = (size_t)S.frag_halfs * 2 + (size_t)S.bias_floats * 4 + 4 * 16 * 64 * 4;  // + the encode tiles: 256 samples x 64 bytes
    if (S.hidden_width == 64) {
        auto kern = mlp_forward_kernel<4, 4, 4>;
        if ((rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes), "lds attr"))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)max_tiles), dim3(256), lds_bytes, stream, L);
    } else {
        auto kern = mlp_forward_kernel<8, 4, 8>;  // (until round 5 <8, 2, 8>: 32 columns per wavefront, two MFMAs per fragment read)
        if ((rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes), "lds attr"))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)max_tiles), dim3(512), lds_bytes, stream, L);
    }
    return check_hip(hipGetLastError(), "mlp_forward_kernel");
}

}  // extern "C"
