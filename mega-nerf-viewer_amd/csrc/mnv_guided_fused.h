// mnv_guided_fused.h -- the guided-sampling frame as ONE kernel (BASELINE.json configs[4]: "per-sample tiny-MLP fused into
// the HIP march kernel").  Instantiated by mnv_accel_fused.hip (it shares AccelLaunch, the ray queues and the launch slots with the march).
//
// What the reference does with four steps and three global buffers per frame (src/renderer/cuda_renderer.cpp:107-139):
//     get_samples_from_voxels   rt_core.cuh:418-576     every dense march step emits (z, world xyz[, dir][, embedding])
//     cumsum + boolean masks    cuda_renderer.cpp:116-121  compaction of the [rays][max_guided_samples] buffer
//     query_submodules          cuda_renderer.cpp:165-203  per-cluster network over the compacted samples
//     render_nerf_results       rt_core.cuh:334-416     CSR composite of the network outputs along every ray
// happens here inside one persistent wavefront per 8x8-pixel tile, and no sample ever reaches global memory:
//   * every lane marches its ray on the packed accel (same traversal and arithmetic as march_accel_kernel).  The composite of
//     sample i needs z_{i+1} (rt_core.cuh:358-362), so a lane holds its newest sample back in registers and releases the
//     previous one -- now complete with its delta z, or flagged as the ray's last -- into a per-wavefront pool in LDS (a ring
//     of 128 entries: world position, delta z, owner lane, cluster, and the pool slot of the owner's next sample);
//   * when the pool holds 64 samples (or nothing more can arrive), the wavefront runs the network on the matrix cores for the
//     pool's first 64 entries (the leading run that shares a cluster): lane j encodes entry j into column j of the MFMA B
//     operand through a 4 KB LDS tile; the weight fragments of that cluster are the A operand, read from global memory
//     (L2-resident, 16 KB for the 64x2 network) one layer ahead of their use -- the identical v_mfma_f32_16x16x32_f16 sequence
//     as mlp_forward_kernel, so the outputs equal mnv_query_submodules' bit for bit;
//   * the outputs cross to LDS ([feature][column]); lane j turns column j into the sample's transmittance factor and the three
//     colour denominators (render_nerf_results' arithmetic, SH basis of the OWNING ray read from LDS), and every owner then
//     walks the chain of its samples in this pass in ray order and accumulates weight / denominator: frames equal the
//     four-kernel path's bit for bit (tests/test_guided_fused_gpu.py), which remains the checker and the path for frames that
//     also track refinement.
// Lanes never wait for the network: a ray with many samples does not hold back its tile's passes (one sample per lane and pass
// -- the first version -- filled 23 of 64 columns on average: passes per tile = samples of its longest ray).
#pragma once

#include "mnv_accel_launch.h"
#include "mnv_mlp.h"

#pragma clang fp contract(off)

namespace mnv {

#ifndef MNV_FUSED_WAVES
#define MNV_FUSED_WAVES (MNV_FUSED_NT >= 4 ? 2 : (MNV_FUSED_NT == 2 ? 3 : 4))  // workgroups per CU = wavefronts per SIMD
#endif

constexpr int kFRayRows = 3 + 3 + 5;  // per-ray LDS rows besides the SH basis: view direction, world-space unit direction, held-back sample
// LDS of one 256-thread workgroup: exp table (256 B) | top-of-tree grid ((2^lds_level)^3 words) | per-ray constants
// [NB + 3][256] floats (SH basis, view direction) | per wavefront: network tile of 16 * mt_out features x 64 columns (the first
// 4 KB double as the encode tile), sample pool 6 x 128 words, per-column results 4 x 64 floats
constexpr int kPool = 128;  // ring capacity: a pass is due at 64 entries and one march step adds at most 64
__host__ __device__ inline int fused_tile_words(int mt_out, int nkk0) { return (mt_out > nkk0 ? mt_out : nkk0) * 16 * kFW; }  // outputs, or one encode tile per K tile
__host__ __device__ inline size_t fused_wave_words(int mt_out, int nkk0) { return (size_t)fused_tile_words(mt_out, nkk0) + 6 * kPool + 4 * kFW; }
__host__ __device__ inline size_t fused_lds_bytes(int nb, int lds_level, int mt_out, int nkk0) {
    return 256 + ((size_t)4 << (3 * lds_level)) + (size_t)(nb + kFRayRows) * 256 * 4 + 4 * fused_wave_words(mt_out, nkk0) * 4;
}

template <int BASIS, int NKK0 /* 32-feature K tiles of the encoded input: 1 or 2 */, bool TRACK /* refinement trackers + visit marks as well (rt_core.cuh:475-507,561-574) */>
__global__ __launch_bounds__(256, MNV_FUSED_WAVES) void guided_fused_kernel(const AccelLaunch K, const FusedGuided F) {
    constexpr int BLOCK = 256, MT = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint64_t *s_exp = reinterpret_cast<uint64_t *>(s_mem);
    constexpr int NB = BASIS > 0 ? BASIS : 1;
    const FrameParams &P = K.P;
    const AccelView &A = K.A;
    const MlpShape &S = F.S;
    const int LL = K.lds_level;
    const int cells = 1 << (3 * LL);
    uint32_t *s_grid = s_mem + 64;
    float *s_ray = reinterpret_cast<float *>(s_grid + cells);  // [k][thread]: k < NB basis, then vdir[3], true_dir[3], held sample[5]
    const int tile_words = fused_tile_words(S.mt_out, NKK0);
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(s_ray + (NB + kFRayRows) * BLOCK) + (threadIdx.x >> 6) * fused_wave_words(S.mt_out, NKK0);  // this wavefront's LDS
    float *s_out = reinterpret_cast<float *>(s_tile);
    float *s_px = s_out + tile_words, *s_py = s_px + kPool, *s_pz = s_py + kPool, *s_pd = s_pz + kPool;  // pool: world xyz, delta z
    uint32_t *s_pm = reinterpret_cast<uint32_t *>(s_pd + kPool), *s_pn = s_pm + kPool;  // owner | last << 6 | cluster << 8; owner's next slot
    float *s_ra = reinterpret_cast<float *>(s_pn + kPool), *s_r0 = s_ra + kFW, *s_r1 = s_r0 + kFW, *s_r2 = s_r1 + kFW;  // per-column results
    constexpr uint32_t kNone = 0xffffffffu;

    if (threadIdx.x < 32) s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
    for (int i = threadIdx.x; i < cells; i += BLOCK) {
        const int G = 1 << LL;
        const int iz = i & (G - 1), iy = (i >> LL) & (G - 1), ix = i >> (2 * LL);
        uint32_t chunk = 0, word = 0;
        for (int l = 1; l <= LL; ++l) {
            const int s = LL - l;
            const int cidx = (((ix >> s) & 1) << 2) | (((iy >> s) & 1) << 1) | ((iz >> s) & 1);
            word = A.nodes[(int64_t)chunk * 8 + cidx];
            if (word & kLeafBit) break;
            chunk = word;
        }
        s_grid[i] = word;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int Lq = A.max_depth;
    const float qscale = __uint_as_float((uint32_t)(127 + Lq) << 23);
    const int sh1 = Lq - LL, L2 = A.grid2_level, sh2 = Lq - L2;
    float *my_ray = s_ray + threadIdx.x;

    // per-lane ray state
    bool has_ray = false, done = true, held = false;
    float t = 0.f, T = 1.f, tmax = 0.f, dir0 = 0.f, dir1 = 0.f, dir2 = 0.f, inv0 = 0.f, inv1 = 0.f, inv2 = 0.f, delta_scale = 0.f;
    uint32_t pix = 0;
    int ns = 0;                              // samples emitted by the ray
    // world-space unit direction and the held-back (newest) sample (z, world xyz, cluster) live in LDS rows NB + 3 .. NB + 10 of the ray
    float *my_td = my_ray + (NB + 3) * BLOCK, *my_held = my_ray + (NB + 6) * BLOCK;
    uint32_t first_pending = kNone, prev_slot = kNone;  // pool slots (monotonic numbers): oldest sample not yet composited, last one pushed
    float ti = 1.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;  // composite state (render_nerf_results)
    // TRACK: per-ray tracker state, as in march_accel_kernel MODE 2 / 3
    float max_weight = -1.f, max_sample_weight = -1.f, sp_prio = 0.f, sa_prio = 0.f;
    int32_t sp_vox = -1, sa_vox = -1;
    int n_eval = 0, n_batches = 0, n_steps = 0, n_cut = 0, n_drain = 0;
    unsigned long long t_net = 0, t_enc = 0, t_hid = 0, t_eval = 0, t_apply = 0, t_all = F.diag ? wall_clock64() : 0;  // diagnostics: 100 MHz ticks
    uint32_t head = 0, tail = 0;             // pool bounds (wave-uniform, monotonic; slot = number & (kPool - 1))

    const uint32_t home = blockIdx.x % kNumQueues;
    uint32_t qsel = 0;
    bool drained = false;
    const CamBlock *__restrict__ Cp = K.cams;
    const float cen0 = Cp->cen[0], cen1 = Cp->cen[1], cen2 = Cp->cen[2];
    const int wave_base = threadIdx.x & ~63;

    for (;;) {
        // ---- a new 8x8 tile once every lane has written its pixel
        if (__ballot(has_ray) == 0) {
            if (drained) break;
            if (qsel >= kNumQueues) {
                drained = true;
                continue;
            }
            const uint32_t q = (home + qsel) % kNumQueues;
            const uint32_t begin = K.band_begin[q] * 64u, span = (K.band_begin[q + 1] - K.band_begin[q]) * 64u;
            uint32_t off = 0;
            if (lane == 0) off = atomicAdd(&K.queue[q * 16], 64u);
            off = __builtin_amdgcn_readfirstlane(off);
            if (off >= span) {
                ++qsel;
                continue;
            }
            const uint32_t id = begin + off + (uint32_t)lane;
            int bx, by;
            uint32_t p;
            if (ray_pixel(K, id, bx, by, p)) {
                pix = p;
                has_ray = true;
                done = true;
                held = false;
                ns = 0;
                first_pending = prev_slot = kNone;
                ti = 1.f;
                o0 = o1 = o2 = 0.f;
                if constexpr (TRACK) {
                    max_weight = max_sample_weight = -1.f;
                    sp_prio = (float)(K.max_depth + 1);
                    sa_prio = (float)(K.max_sample_count + 1);
                    sp_vox = sa_vox = -1;
                }
                RaySetup<NB> r;
                setup_ray<(BASIS > 0 ? BASIS : 0)>(P, *Cp, P.x0 + bx, P.y0 + by, r, frame_tmax(P, p));
                if constexpr (BASIS == 0) r.basis[0] = (0 < P.basis_min || 0 > P.basis_max) ? 0.f : (float)0.28209479177387814;
                float true_dir[3], vdir[3];
                world_ray_dirs(P, *Cp, P.x0 + bx, P.y0 + by, true_dir, vdir);
#pragma unroll
                for (int k = 0; k < 3; ++k) my_td[k * BLOCK] = true_dir[k];
#pragma unroll
                for (int k = 0; k < NB; ++k) my_ray[k * BLOCK] = r.basis[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) my_ray[(NB + k) * BLOCK] = vdir[k];
                if (r.in_bbox) {
                    done = false;
                    t = r.tmin;
                    T = 1.f;
                    tmax = r.tmax;
                    dir0 = r.dir[0]; dir1 = r.dir[1]; dir2 = r.dir[2];
                    inv0 = r.invdir[0]; inv1 = r.invdir[1]; inv2 = r.invdir[2];
                    delta_scale = r.delta_scale;
                }
            }
            continue;
        }

        // ---- one march step (rt_core.cuh:452-560) for the lanes whose ray is still under way
        bool fresh = false;                      // this step emitted a sample
        float sz = 0.f, sx = 0.f, sy = 0.f, sw = 0.f;
        int scl = -1;
        if (__ballot(has_ray && !done) != 0) ++n_steps;
        if (has_ray && !done) {
            if (!(t < tmax)) {
                done = true;
            } else {
                float pos[3];
                uint32_t q[3];
                pos[0] = cen0 + t * dir0;
                pos[1] = cen1 + t * dir1;
                pos[2] = cen2 + t * dir2;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pos[i] = __builtin_amdgcn_fmed3f(pos[i], 0.f, 1.f - 1e-6f);
                    q[i] = (uint32_t)(pos[i] * qscale);
                }
                uint32_t word = s_grid[((((q[0] >> sh1) << LL) | (q[1] >> sh1)) << LL) | (q[2] >> sh1)];
                int src = 0;       // TRACK: where the leaf word came from (0 LDS grid, 1 grid2, 2 node array) ...
                uint32_t vox = 0;  // ... and the leaf's voxel index (grid cell number until it is looked up)
                if (!(word & kLeafBit)) {
                    word = descend_to_leaf(A, q, word, sh1, sh2, L2, LL, src, vox);
                }
                const int depth = (int)((word >> 16) & 0x7fu);
                const float sc = __uint_as_float((uint32_t)(127 + depth) << 23);
                const float inv_cube = __uint_as_float((uint32_t)(127 - depth) << 23);
                float tu = 1e4f;
                const float invd[3] = {inv0, inv1, inv2};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float x = __builtin_amdgcn_fractf(pos[i] * sc);
                    const float t1 = -x * invd[i];
                    const float t2 = t1 + invd[i];
                    tu = fminf(tu, fmaxf(t1, t2));
                }
                const float delta_t = tu * inv_cube + P.step_size;
                const float sigma = half_bits_to_float((uint16_t)word);
                const bool is_dense = sigma > P.sigma_thresh;
                bool need_vox = false;
                if constexpr (TRACK) {
                    need_vox = is_dense || max_weight == -1.f || max_sample_weight == -1.f || K.visited != nullptr;
                    if (need_vox) {
                        const int shg = Lq - A.grid_level;
                        if (src == 0) vox = A.grid_vox[((((q[0] >> shg) << A.grid_level) + (q[1] >> shg)) << A.grid_level) + (q[2] >> shg)];
                        else if (src == 1) vox = A.grid2_vox[vox];
                        if (K.visited && __hip_atomic_load(&K.visited[vox >> 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) K.visited[vox >> 3] = 1;
                    }
                }
                if constexpr (TRACK) {
                    if (need_vox && !is_dense) {  // first leaf before any dense one, rt_core.cuh:561-574
                        if (depth < K.max_depth && max_weight == -1.f) {
                            sp_vox = (int32_t)vox;
                            sp_prio = (float)depth;
                        }
                        if (K.sample_counts && max_sample_weight == -1.f) {
                            const int16_t scn = K.sample_counts[vox];
                            if (scn < K.max_sample_count) {
                                sa_vox = (int32_t)vox;
                                sa_prio = (float)scn;
                            }
                        }
                    }
                }
                if (is_dense) {
                    const float att = exact_expf(-delta_t * delta_scale * sigma, s_exp);
                    if constexpr (TRACK) {  // best dense leaf so far, rt_core.cuh:475-507
                        const float weight = T * (1.f - att);
                        if (depth < K.max_depth && weight > max_weight) {
                            sp_vox = (int32_t)vox;
                            sp_prio = (float)depth;
                            max_weight = weight;
                        }
                        if (K.sample_counts && weight > max_sample_weight) {
                            const int16_t scn = K.sample_counts[vox];
                            if (scn < K.max_sample_count) {
                                sa_vox = (int32_t)vox;
                                sa_prio = (float)scn;
                                max_sample_weight = weight;
                            }
                        }
                    }
                    // rt_core.cuh:508-549: one sample per dense step while there is room
                    if (ns < F.max_guided_samples) {
                        const float tz0 = t * dir0 / P.scale[0], tz1 = t * dir1 / P.scale[1], tz2 = t * dir2 / P.scale[2];
                        sz = sqrtf(tz0 * tz0 + tz1 * tz1 + tz2 * tz2);
                        const float *m = Cp->c2w;
                        sx = m[9] + my_td[0] * sz;
                        sy = m[10] + my_td[BLOCK] * sz;
                        sw = m[11] + my_td[2 * BLOCK] * sz;
                        const int g1 = (int)fmaxf(fminf((sy - F.min_position[1]) / F.range[1] * (float)F.grid_dim[0], (float)F.grid_dim[0] - 1.0f), 0.0f);
                        const int g2 = (int)fmaxf(fminf((sw - F.min_position[2]) / F.range[2] * (float)F.grid_dim[1], (float)F.grid_dim[1] - 1.0f), 0.0f);
                        scl = (int)(int16_t)(g1 * F.grid_dim[1] + g2);
                        fresh = true;
                        ++ns;
                    }
                    T *= att;
                    if (T < P.stop_thresh) done = true;
                }
                t += delta_t;
                // a ray that has emitted its quota contributes nothing more to the picture: its remaining steps are skipped
                // (the trackers and visit marks do follow the remaining steps)
                if constexpr (!TRACK) {
                    if (ns >= F.max_guided_samples) done = true;
                }
            }
        }

        // ---- release complete samples into the pool: the held one once its successor exists (delta z known), or as the ray's
        //      last sample one step after the ray ended
        {
            bool push = false, last = false;
            float pz_ = 0.f, px_ = 0.f, py_ = 0.f, pw_ = 0.f, pdz = 0.f;
            int pcl = -1;
            if (fresh) {
                if (held) {
                    push = true;
                    pz_ = my_held[0]; px_ = my_held[BLOCK]; py_ = my_held[2 * BLOCK]; pw_ = my_held[3 * BLOCK]; pcl = __float_as_int(my_held[4 * BLOCK]);
                    pdz = sz - pz_;  // delta_i = z[i + 1] - z[i], rt_core.cuh:359
                }
                my_held[0] = sz; my_held[BLOCK] = sx; my_held[2 * BLOCK] = sy; my_held[3 * BLOCK] = sw; my_held[4 * BLOCK] = __int_as_float(scl);
                held = true;
            } else if (has_ray && done && held) {
                push = true;
                last = true;
                pz_ = my_held[0]; px_ = my_held[BLOCK]; py_ = my_held[2 * BLOCK]; pw_ = my_held[3 * BLOCK]; pcl = __float_as_int(my_held[4 * BLOCK]);
                held = false;
            }
            (void)pz_;
            const uint64_t pm = __ballot(push);
            if (pm != 0) {
                if (push) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    const uint32_t slot = tail + rank, e = slot & (kPool - 1);
                    s_px[e] = px_;
                    s_py[e] = py_;
                    s_pz[e] = pw_;
                    s_pd[e] = pdz;
                    s_pm[e] = (uint32_t)lane | (last ? 64u : 0u) | ((uint32_t)(pcl & 0xffff) << 8);
                    s_pn[e] = kNone;
                    if (prev_slot != kNone && (int32_t)(prev_slot - head) >= 0) s_pn[prev_slot & (kPool - 1)] = slot;  // still in the pool: chain it
                    if (first_pending == kNone) first_pending = slot;
                    prev_slot = slot;
                }
                tail += (uint32_t)__popcll(pm);
                __builtin_amdgcn_wave_barrier();
            }
        }

        // ---- the network for the head of the pool: when 64 samples wait, or when nothing more can arrive
        //      (a pass may take fewer than 64 entries -- a cluster boundary -- so passes repeat until fewer than batch_min <= 64 wait:
        //      the next march step then adds at most 64 and the ring of 128 cannot overflow)
        const bool more_to_come = __ballot(has_ray && (!done || held)) != 0;
        while (tail - head >= (uint32_t)F.batch_min || (tail != head && !more_to_come)) {
            const unsigned long long t_w0 = F.diag ? wall_clock64() : 0;
            const uint32_t size = tail - head;
            if (size < (uint32_t)F.batch_min) ++n_drain;
            const int n = size < (uint32_t)kFW ? (int)size : kFW;
            const int cj = lane & (kFW - 1), part = lane / kFW;  // this lane's column and which share of the per-column work it does
            const uint32_t e = (head + (uint32_t)cj) & (kPool - 1);
            const bool col_on = cj < n;
            const uint32_t meta = col_on ? s_pm[e] : 0u;
            const int owner = (int)(meta & 63u), my_cl = (int)(int16_t)(meta >> 8);
            // The window's samples may belong to several sub-modules (a ray that crosses the front and the back of a surface changes
            // cluster on the way, and its neighbours do so a few steps apart): the network runs once per distinct cluster of the
            // window, every run fills the columns of its own cluster, and the pool stays first-in first-out.
            uint64_t todo = __ballot(col_on && part == 0);  // one bit per column (lanes 0 .. W-1)
            while (todo != 0) {
            const int c_star = __builtin_amdgcn_readfirstlane(__shfl(my_cl, (int)__builtin_ctzll(todo)));
            const bool col_sel = col_on && my_cl == c_star;
            const uint64_t sel = __ballot(col_sel && part == 0);
            todo &= ~sel;
            ++n_batches;
            if (todo != 0) ++n_cut;
            const bool valid_cluster = c_star >= 0 && c_star < S.n_clusters;
            if (valid_cluster) {
                float p[3], d[3];
                p[0] = ((col_on ? s_px[e] : 0.f) - S.center[0]) * S.inv_extent[0];
                p[1] = ((col_on ? s_py[e] : 0.f) - S.center[1]) * S.inv_extent[1];
                p[2] = ((col_on ? s_pz[e] : 0.f) - S.center[2]) * S.inv_extent[2];
#pragma unroll
                for (int i = 0; i < 3; ++i) d[i] = S.need_viewdir ? s_ray[(NB + i) * BLOCK + wave_base + owner] : 0.f;
                const uint16_t *emb = nullptr;
                if (S.n_embeddings > 0) {
                    int idx = (int)(float)F.appearance_embedding;
                    idx = idx < 0 ? 0 : (idx >= S.n_embeddings ? S.n_embeddings - 1 : idx);
                    emb = F.embeddings + ((size_t)c_star * S.n_embeddings + idx) * S.embedding_dim;
                }
                const half8 *w = reinterpret_cast<const half8 *>(F.frags + (size_t)c_star * S.frag_halfs);
                const float *b = F.biases + (size_t)c_star * S.bias_floats;
                // The weight fragments come from global memory (L2, 16 KB for the 64x2 network): a layer's 8 fragments are requested
                // before the work that precedes its MFMAs (layer 0: the encode; later layers: the ReLU + float -> half conversion of
                // the previous accumulators) and wait in 32 registers.  (Requesting one layer further ahead cost 48 more live
                // registers: the allocator then spilled ray state that the encode loop reloads in every iteration.)
                constexpr int kFr = MT * (MT / 2);  // fragments of a hidden layer; the output layer's (2 * mt_out) are read as 8 as well
                                                    // (the arrays are padded, mnv_mlp_create), so every fetch is 8 unconditional loads
                half8 cur[kFr];
                auto fetch = [&](half8 *dst, const half8 *src, int count) {  // count is a compile-time constant at every call
#pragma unroll
                    for (int i = 0; i < kFr; ++i)
                        if (i < count) dst[i] = src[i * 64 + lane];
                };
                // the layer's bias tiles travel with its fragments (16 registers) when the register budget allows: 64-column runs
                f32x4 bias_pre[kFNT >= 4 ? MT : 1];
                auto fetch_bias = [&](const float *bsrc) {
                    if constexpr (kFNT >= 4) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) bias_pre[mt] = *reinterpret_cast<const f32x4 *>(bsrc + 16 * mt + 4 * g);
                    }
                };
                auto bias_of = [&](const float *bsrc, int mt) -> f32x4 {
                    if constexpr (kFNT >= 4) return bias_pre[mt];
                    else return *reinterpret_cast<const f32x4 *>(bsrc + 16 * mt + 4 * g);
                };
                constexpr int n0 = MT * NKK0;  // layer 0: fragment (mt, kk) at index mt * NKK0 + kk
                static_assert(n0 <= kFr, "layer 0 is prefetched whole");
                fetch(cur, w, n0);
                fetch_bias(b);
                const unsigned long long t_e0 = F.diag ? wall_clock64() : 0;
                f32x4 acc[MT][kFNT];
                // layer 0.  Every lane encodes its own column: feature f goes, as a half, to the slot of the MFMA B operand that
                // slot_feature() assigns it (K tile f >> 5, lane group and element from f & 31), one 4 KB tile per K tile.  Straight-line
                // per octave -- the generic encode_feature() loop of mlp_forward_kernel re-reads the network shape from the kernel
                // arguments in every iteration, which this kernel's scalar-register pressure turned into 12 k cycles per run.
                {
                    _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_tile);
                    auto put = [&](int f, float v) {  // f is uniform within a lane share
                        const int r = f & 31;
                        const int dw = ((((r & 15) >> 2) * 4 + (((r >> 4) * 4 + (r & 3)) >> 1)) * kFW) + (f >> 5) * (16 * kFW);
                        tile_h[(dw + cj) * 2 + (r & 1)] = (_Float16)v;  // element e = (r >> 4) * 4 + (r & 3): its low bit is r & 1
                    };
                    // the 64 / W lanes of a column share its features: octave k belongs to lane share k % parts, the rest to share 0
                    auto octaves = [&](int base, int n_oct, const float x[3]) {
                        if (part == 0) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) put(base + i, x[i]);
                        }
                        for (int k = part; k < n_oct; k += kFParts) {
                            const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                put(base + 3 + 6 * k + i, tri_wave(x[i] * scale + 0.f));
                                put(base + 3 + 6 * k + 3 + i, tri_wave(x[i] * scale + 0.25f));
                            }
                        }
                    };
                    octaves(0, S.pos_octaves, p);
                    if (S.need_viewdir) octaves(S.dir_base, S.dir_octaves, d);  // (every block starts at a multiple of 16 slots: mnv_mlp.h)
                    if (part == 0) {
                        for (int j = 0; j < S.embedding_dim; ++j) put(S.emb_base + j, half_bits_to_float(emb[j]));
                        // gaps and padding: finite (their weights are zero)
                        for (int f = S.n_pos; f < S.dir_base; ++f) put(f, 0.f);
                        for (int f = S.dir_base + S.n_dir; f < S.emb_base; ++f) put(f, 0.f);
                        for (int f = S.k_slots; f < 32 * NKK0; ++f) put(f, 0.f);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int kk = 0; kk < NKK0; ++kk) {
                    half8 bf[kFNT];
#pragma unroll
                    for (int nt = 0; nt < kFNT; ++nt) {
                        union {
                            uint32_t u[4];
                            half8 h;
                        } rd;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) rd.u[q4] = s_tile[kk * (16 * kFW) + (g * 4 + q4) * kFW + nt * 16 + col];
                        bf[nt] = rd.h;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (kk == 0) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const f32x4 bv = bias_of(b, mt);
#pragma unroll
                            for (int nt = 0; nt < kFNT; ++nt) acc[mt][nt] = bv;
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int nt = 0; nt < kFNT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[mt * NKK0 + kk], bf[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
                if (F.diag) t_enc += wall_clock64() - t_e0;
                const unsigned long long t_h0 = F.diag ? wall_clock64() : 0;
                w += n0 * 64;
                b += 16 * MT;
                for (int layer = 1; layer <= S.hidden_layers; ++layer) {
                    const int n_mt = layer < S.hidden_layers ? MT : S.mt_out;
                    fetch(cur, w, kFr);  // this layer's fragments travel while the activations are converted
                    fetch_bias(b);
                    half8 bf[MT / 2][kFNT];
#pragma unroll
                    for (int kk = 0; kk < MT / 2; ++kk)
#pragma unroll
                        for (int nt = 0; nt < kFNT; ++nt) bf[kk][nt] = relu_pack(acc[2 * kk][nt], acc[2 * kk + 1][nt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (mt < n_mt) {
                            const f32x4 bv = bias_of(b, mt);
#pragma unroll
                            for (int nt = 0; nt < kFNT; ++nt) acc[mt][nt] = bv;
#pragma unroll
                            for (int kk = 0; kk < MT / 2; ++kk) {
#pragma unroll
                                for (int nt = 0; nt < kFNT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[mt * (MT / 2) + kk], bf[kk][nt], acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                    w += n_mt * (MT / 2) * 64;
                    b += 16 * n_mt;
                }
                if (F.diag) t_hid += wall_clock64() - t_h0;
                // outputs to the tile: feature f of column j at s_out[f * 64 + j]
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt < S.mt_out) {
#pragma unroll
                        for (int nt = 0; nt < kFNT; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s_out[(16 * mt + 4 * g + r) * kFW + nt * 16 + col] = acc[mt][nt][r];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            const unsigned long long t_c0 = F.diag ? wall_clock64() : 0;
            // ---- column j: transmittance factor and colour denominators of its sample (rt_core.cuh:356-392), SH basis of the owner
            if (col_sel) {
                auto sv = [&](int f) -> float { return valid_cluster ? s_out[f * kFW + cj] : 0.f; };  // no sub-module: zeros (mlp_histogram)
                // the four quantities of a column are dealt to its 64 / W lanes: quantity q belongs to lane share q % parts
                if (0 % kFParts == part) {
                    const bool last = (meta & 64u) != 0;
                    s_ra[cj] = last ? 0.f : exact_expf(-sv(3) * s_pd[e], s_exp);
                }
                if constexpr (BASIS >= 0) {
                    float basis[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) basis[k] = s_ray[k * BLOCK + wave_base + owner];
                    const int stride = BASIS > 0 ? BASIS : 0;
                    if (1 % kFParts == part) s_r0[cj] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, 0), s_exp);
                    if (2 % kFParts == part) s_r1[cj] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, stride), s_exp);
                    if (3 % kFParts == part) s_r2[cj] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, 2 * stride), s_exp);
                } else {
                    if (part == 0) {
                        s_r0[cj] = sv(0);
                        s_r1[cj] = sv(1);
                        s_r2[cj] = sv(2);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next cluster's run rewrites the tile
            if (F.diag) t_eval += wall_clock64() - t_c0;
            }
            const unsigned long long t_a0 = F.diag ? wall_clock64() : 0;
            // ---- every owner walks its samples of this pass in ray order
            while (first_pending != kNone && (int32_t)(first_pending - (head + (uint32_t)n)) < 0) {
                const int j = (int)(first_pending - head);
                const uint32_t ee = first_pending & (kPool - 1);
                const bool last = (s_pm[ee] & 64u) != 0;
                const float wc = s_ra[j];
                const float weight = last ? ti : ti * (1.0f - wc);
                if constexpr (BASIS >= 0) {
                    o0 += weight / s_r0[j];
                    o1 += weight / s_r1[j];
                    o2 += weight / s_r2[j];
                } else {
                    o0 += weight * s_r0[j];
                    o1 += weight * s_r1[j];
                    o2 += weight * s_r2[j];
                }
                ti *= wc;
                first_pending = s_pn[ee];
                ++n_eval;
            }
            head += (uint32_t)n;
            __builtin_amdgcn_wave_barrier();  // tile, results and the freed pool slots are rewritten from here on
            if (F.diag) t_apply += wall_clock64() - t_a0;
            if (F.diag) t_net += wall_clock64() - t_w0;
        }
        // ---- rays that have ended and whose samples are all composited: write the pixel (alpha 1, renderer_kernel.cu:316)
        if (has_ray && done && !held && first_pending == kNone) {
            composite_and_write(P, (int64_t)pix, o0, o1, o2, 1.0f);
            if constexpr (TRACK) {
                if (K.split_track) {
                    K.split_track[(int64_t)pix * 3 + 0] = sp_prio;
                    K.split_track[(int64_t)pix * 3 + 1] = sp_vox < 0 ? -1.f : (float)(sp_vox >> 3);
                    K.split_track[(int64_t)pix * 3 + 2] = sp_vox < 0 ? -1.f : (float)(sp_vox & 7);
                }
                if (K.sample_track) {
                    K.sample_track[(int64_t)pix * 3 + 0] = sa_prio;
                    K.sample_track[(int64_t)pix * 3 + 1] = sa_vox < 0 ? -1.f : (float)(sa_vox >> 3);
                    K.sample_track[(int64_t)pix * 3 + 2] = sa_vox < 0 ? -1.f : (float)(sa_vox & 7);
                }
            }
            has_ray = false;
        }
    }
    if (F.sample_counter) {
        // one atomic per wavefront
        int tot = n_eval;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (lane == 0 && tot) atomicAdd(F.sample_counter, (unsigned long long)tot);
    }
    if (F.diag && lane == 0) {  // diagnostics buffer of mnv_set_fused_diag (32 words): network passes, march iterations, phase times
        atomicAdd(F.diag + 1, (unsigned long long)n_batches);
        atomicAdd(F.diag + 2, (unsigned long long)n_steps);
        atomicAdd(F.diag + 3, (unsigned long long)n_cut);
        atomicAdd(F.diag + 4, (unsigned long long)n_drain);
        atomicAdd(F.diag + 5, t_net);
        atomicAdd(F.diag + 6, wall_clock64() - t_all);
        atomicAdd(F.diag + 7, t_enc);
        atomicAdd(F.diag + 8, t_hid);
        atomicAdd(F.diag + 9, t_eval);
        atomicAdd(F.diag + 10, t_apply);
    }
}

}  // namespace mnv
