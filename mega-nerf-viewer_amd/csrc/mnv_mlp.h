// mnv_mlp.h -- shapes and device helpers of the per-sample sub-module network shared by its two users:
//   csrc/mnv_mlp.hip            mnv_query_submodules: the network as a pass over a sample buffer
//   csrc/mnv_guided_fused.h     the same network evaluated inside the guided-sampling march (BASELINE.json configs[4])
// Both run the identical MFMA sequence on the identical weight fragments, so their outputs are equal bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnv.h"
#include "mnv_device.h"

namespace mnv {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRowsPerPass = 256;   // 4 wavefronts x 64 rows
constexpr int kNT = 4;              // 16-row MFMA column tiles per wavefront
constexpr int kMaxClusters = 1024;

struct MlpShape {
    int32_t n_clusters, pos_octaves, dir_octaves, need_viewdir, n_embeddings, embedding_dim;
    int32_t hidden_width, hidden_layers, out_dim;
    int32_t in_dim, n_pos, n_dir;   // encoded widths
    // K slots of the first layer (round 5): the position block, the direction block and the embedding each START at a multiple of 16 -- a half K
    // tile never mixes blocks, so that every block's features sit at compile-time places of its half tiles (slot_of_feature / feature_of_slot;
    // the gaps carry zero weights).  The parameter blob keeps the plain feature order.
    int32_t dir_base, emb_base, k_slots;
    int32_t nkk0;                   // K tiles (32) of the first layer: ceil(k_slots / 32)
    int32_t mt_hidden, mt_out;      // M tiles (16) of hidden / output layers
    int32_t frag_halfs;             // per cluster: all weight fragments
    int32_t bias_floats;            // per cluster: all biases, padded per layer
    float center[3], inv_extent[3];
};

}  // namespace mnv

struct mnv_mlp {
    mnv::MlpShape shape;
    uint16_t *frags = nullptr;      // [n_clusters][frag_halfs] binary16, fragment order
    float *biases = nullptr;        // [n_clusters][bias_floats]
    uint16_t *embeddings = nullptr; // [n_clusters][n_embeddings][embedding_dim] binary16
    uint8_t *scratch = nullptr;     // grow-only: order, tiles, counters
    size_t scratch_bytes = 0;
    int num_cus = 0;
    int blocks_per_cu = 0;          // workgroups of this network's forward kernel a compute unit holds (asked once)
};

namespace mnv {

// K slot of the first layer handled by element e of lane group g in K tile kk (see the header comment)
__host__ __device__ inline int slot_feature(int kk, int g, int e) { return 32 * kk + 16 * (e >> 2) + 4 * g + (e & 3); }
// ... and the input feature that lives in K slot s of the first layer (-1: a gap or padding -- zero weights)
__host__ __device__ inline int feature_of_slot(const MlpShape &S, int s) {
    if (s < S.dir_base) return s < S.n_pos ? s : -1;
    if (s < S.emb_base) return s - S.dir_base < S.n_dir ? S.n_pos + (s - S.dir_base) : -1;
    return s - S.emb_base < S.embedding_dim ? S.n_pos + S.n_dir + (s - S.emb_base) : -1;
}

__host__ __device__ inline float tri_wave(float t) {
    const float r = t - floorf(t + 0.5f);
    return 4.f * fabsf(r) - 1.f;
}

// value of encoded feature f for one sample (before the binary16 rounding); 0 for padding features
__host__ __device__ inline float encode_feature(const MlpShape &S, int f, const float p[3], const float d[3]) {
    if (f < S.n_pos) {
        if (f < 3) return p[f];
        const int q = f - 3, k = q / 6, r = q - 6 * k, i = r % 3;
        const float scale = (float)(1u << k);
        return tri_wave(p[i] * scale + (r >= 3 ? 0.25f : 0.f));
    }
    f -= S.n_pos;
    if (f < S.n_dir) {
        if (f < 3) return d[f];
        const int q = f - 3, k = q / 6, r = q - 6 * k, i = r % 3;
        const float scale = (float)(1u << k);
        return tri_wave(d[i] * scale + (r >= 3 ? 0.25f : 0.f));
    }
    return 0.f;  // embedding features are looked up by the caller
}

// ReLU + binary16 rounding of eight accumulators.  Rounding first and clamping the packed halves as signed 16-bit integers (a
// half with its sign bit set -- negative or -0 -- is a negative integer) gives the bits of half(max(x, 0)) for every non-NaN x:
// rounding is monotonic and keeps the sign.  Two instructions per pair (v_cvt_pk_f16_f32, v_pk_max_i16) instead of five
// (fmaxf lowers to a canonicalising v_max plus the v_max itself, per value).
__device__ inline half8 relu_pack(const f32x4 &lo, const f32x4 &hi) {
    typedef short short2v __attribute__((ext_vector_type(2)));
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    union {
        half8 h;
        short2v s[4];
    } u;
    const short2v zero = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float2v fa = {lo[2 * r], lo[2 * r + 1]}, fb = {hi[2 * r], hi[2 * r + 1]};
        const half2v a = __builtin_convertvector(fa, half2v), b = __builtin_convertvector(fb, half2v);  // round to nearest even, as (_Float16)x
        u.s[r] = __builtin_elementwise_max(__builtin_bit_cast(short2v, a), zero);
        u.s[2 + r] = __builtin_elementwise_max(__builtin_bit_cast(short2v, b), zero);
    }
    return u.h;
}

}  // namespace mnv
