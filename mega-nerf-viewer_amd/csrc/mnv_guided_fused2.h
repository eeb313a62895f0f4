// mnv_guided_fused2.h -- the guided-sampling frame as ONE kernel with SPECIALISED wavefronts (BASELINE.json configs[4]).
// Included by mnv_march_accel.hip after mnv_guided_fused.h, whose kernel (one wavefront does both jobs) it replaces wherever the
// network's weights fit a workgroup's LDS; that kernel stays as the path for deeper networks and as a second checker.
//
// Same four reference steps (src/renderer/cuda_renderer.cpp:107-139: get_samples_from_voxels rt_core.cuh:418-576, cumsum / masks,
// query_submodules :165-203, render_nerf_results rt_core.cuh:334-416), same arithmetic, same frames bit for bit -- but the march and
// the network no longer share one wavefront's registers and time:
//   * a workgroup is NP PRODUCER wavefronts + one CONSUMER wavefront.  A producer marches an 8x8 tile, one ray per lane, on the packed
//     accel exactly as guided_fused_kernel does and pushes every complete sample (world position, delta z, owner lane, cluster) into
//     ITS ring in LDS (128 entries); it never touches the matrix cores.  The composite of a ray's samples happens in the owning lane,
//     in ray order, whenever results have arrived: one LDS poll per march iteration;
//   * the consumer gathers up to 64 waiting entries from the NP rings (oldest first per ring, rings that asked for a flush first),
//     encodes them, runs the network on the matrix cores -- the identical v_mfma_f32_16x16x32_f16 sequence on the identical
//     fragments as mlp_forward_kernel -- with the A operands read from LDS, where the weights of the workgroup's current sub-module
//     stay between runs (a run of a minority sub-module reads its fragments from L2 instead, as guided_fused_kernel always does),
//     turns every column into the sample's transmittance factor and colour denominators (SH basis of the owning ray from LDS) and
//     writes those four floats over the sample's ring entry;
//   * rings, results and the three counters per ring (pushed / flush request / evaluated) live in LDS; waves of one workgroup are
//     co-resident by construction, so the spin-waits (s_sleep) cannot deadlock: a producer waits only when its ring holds more than
//     64 entries (the consumer then has a full window) or after it has asked for a flush; the consumer never waits for a producer.
// What this buys (cfg2, 1080p, 9.4 M samples): the march runs in 12 wavefronts per CU that do nothing else (was: 8 that spent 44 %
// of their time in the network), the network's latency chain loses its three L2 round trips per run, and the register budget is
// 128 instead of 248 (4 wavefronts per SIMD instead of 2).
#pragma once

#include <type_traits>

#include "mnv_guided_fused.h"

#pragma clang fp contract(off)

namespace mnv {

#ifndef MNV_F2_NP
#define MNV_F2_NP 3  // producer wavefronts per workgroup (+ 1 consumer): 4 workgroups of 256 threads per CU at 128 VGPRs
#endif
constexpr int kF2NP = MNV_F2_NP;
constexpr int kF2Block = 64 * (kF2NP + 1);
constexpr int kF2Ring = 128;                       // entries per producer ring: a window is due at 64 and one march step adds at most 64
constexpr int kF2RingWords = kF2Ring * (4 + 1 + 1);  // float4 {x, y, z, dz} -> {att, d0, d1, d2} | meta | owner's next slot
constexpr int kF2WavesPerSimd = 4;                 // register budget: 128 VGPRs
constexpr bool kF2Default = false;                 // mnv_set_fused_kernel(0) picks this kernel when it fits (until it is the faster one: no)
// Watchdog of the spin-waits: a wait that lasts this many polls (s_sleep 1-2 each: tens of milliseconds; a healthy wait is a few
// microseconds) is abandoned and the wavefront leaves -- wrong pixels and a count in the diagnostics buffer instead of a hung device.
// No schedule of co-resident waves reaches it (header comment); it exists so that a bug cannot take the machine down.
constexpr uint32_t kF2SpinLimit = 1u << 18;

struct F2Layout {  // word offsets into the dynamic LDS block
    int grid, ray, rings, ctrl, frags, bias, tile, total;
    int ray_rows;
};
__host__ __device__ inline F2Layout f2_layout(int nb, int lds_level, const MlpShape &S) {
    F2Layout L;
    L.ray_rows = nb + (S.need_viewdir ? 3 : 0);
    L.grid = 64;                                          // after the exp table
    L.ray = L.grid + (1 << (3 * lds_level));
    L.rings = (L.ray + L.ray_rows * kF2NP * 64 + 3) & ~3;  // 16-byte aligned entries
    L.ctrl = L.rings + kF2NP * kF2RingWords;
    L.frags = L.ctrl + 16;                                // 4 words per ring, room for 4 rings
    L.bias = L.frags + S.frag_halfs / 2;
    L.tile = (L.bias + S.bias_floats + 3) & ~3;
    const int enc = S.nkk0 * 16 * 64, out = 16 * S.mt_out * 32;  // encode tiles (one per K tile); outputs of 32 columns at a time
    L.total = L.tile + (enc > out ? enc : out);
    return L;
}
static_assert(kF2NP >= 1 && kF2NP <= 4, "ctrl block holds 4 rings");

struct F2Diag {  // MNV_FUSED_DIAG: sums over wavefronts, 100 MHz ticks
    enum { kRuns = 1, kSteps, kWindows, kReloads, kGlobalRuns, kConsBusy, kConsTotal, kProdTotal, kProdRingWait, kProdFlushWait, kEnc, kLayers, kEval, kColumns, kWatchdog, kConsSimd /* 4 words: consumer wavefronts per SIMD id */, kProdSimd = kConsSimd + 4, kWords = kProdSimd + 4 };
};

template <int BASIS, int NKK0, bool TRACK>
__global__ __launch_bounds__(kF2Block, kF2WavesPerSimd) void guided_fused2_kernel(const AccelLaunch K, const FusedGuided F) {
    constexpr int MT = 4, RB = kF2NP * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint64_t *s_exp = reinterpret_cast<uint64_t *>(s_mem);
    constexpr int NB = BASIS > 0 ? BASIS : 1;
    const FrameParams &P = K.P;
    const AccelView &A = K.A;
    const MlpShape &S = F.S;
    const int LL = K.lds_level;
    const F2Layout Lo = f2_layout(NB, LL, S);
    uint32_t *s_grid = s_mem + Lo.grid;
    float *s_ray = reinterpret_cast<float *>(s_mem + Lo.ray);  // [row][producer thread]: SH basis, then the view direction
    uint32_t *s_ctrl = s_mem + Lo.ctrl;                        // per ring: pushed, flush request, evaluated, producer has left
    constexpr uint32_t kNone = 0xffffffffu;

    {
        const int cells = 1 << (3 * LL);
        if (threadIdx.x < 32) s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
        if (threadIdx.x < 16) s_ctrl[threadIdx.x] = 0u;
        for (int i = threadIdx.x; i < cells; i += kF2Block) {
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
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: the role branch below is a scalar branch
    const unsigned long long t_begin = F.diag ? wall_clock64() : 0;
    auto ld_relaxed = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto st_release = [&](uint32_t *p, uint32_t v) {
        if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (wave < kF2NP) {
        // =================================================================================== producer: march, push, composite
        float4 *r_data = reinterpret_cast<float4 *>(s_mem + Lo.rings + wave * kF2RingWords);
        uint32_t *r_meta = reinterpret_cast<uint32_t *>(r_data + kF2Ring), *r_next = r_meta + kF2Ring;
        uint32_t *c_tail = s_ctrl + 4 * wave, *c_flush = c_tail + 1, *c_ready = c_tail + 2, *c_exit = c_tail + 3;
        float *my_ray = s_ray + wave * 64 + lane;  // [k * RB]

        const int Lq = A.max_depth;
        const float qscale = __uint_as_float((uint32_t)(127 + Lq) << 23);
        const int sh1 = Lq - LL, L2 = A.grid2_level, sh2 = Lq - L2;

        bool has_ray = false, done = true, held = false;
        float t = 0.f, T = 1.f, tmax = 0.f, dir0 = 0.f, dir1 = 0.f, dir2 = 0.f, inv0 = 0.f, inv1 = 0.f, inv2 = 0.f, delta_scale = 0.f;
        float td0 = 0.f, td1 = 0.f, td2 = 0.f;                       // world-space unit direction
        float hz = 0.f, hx = 0.f, hy = 0.f, hw = 0.f;                // the held-back (newest) sample: z, world xyz ...
        int hcl = -1;                                                // ... and its cluster
        uint32_t pix = 0;
        int ns = 0;
        uint32_t first_pending = kNone, prev_slot = kNone;  // ring slots (monotonic numbers): oldest sample not yet composited, last one pushed
        float ti = 1.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;       // composite state (render_nerf_results)
        float max_weight = -1.f, max_sample_weight = -1.f, sp_prio = 0.f, sa_prio = 0.f;
        int32_t sp_vox = -1, sa_vox = -1;
        int n_eval = 0, n_steps = 0;
        unsigned long long t_ring = 0, t_flush = 0;
        uint32_t spins = 0;                           // consecutive waits (watchdog)
        uint32_t tail = 0, seen = 0, flush_sent = 0;  // wave-uniform: entries pushed; evaluated entries this wave has composited; last flush request

        const uint32_t home = blockIdx.x % kNumQueues;
        uint32_t qsel = 0;
        bool drained = false;
        const CamBlock *__restrict__ Cp = K.cams;
        const float cen0 = Cp->cen[0], cen1 = Cp->cen[1], cen2 = Cp->cen[2];

        for (;;) {
            // ---- results that arrived since the last look: every owner walks its samples in ray order (rt_core.cuh:356-392)
            {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(c_ready));
                if (r != seen) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    while (first_pending != kNone && (int32_t)(first_pending - r) < 0) {
                        const uint32_t e = first_pending & (kF2Ring - 1);
                        const float4 res = r_data[e];
                        const bool last = (r_meta[e] & 64u) != 0;
                        const float wc = res.x;
                        const float weight = last ? ti : ti * (1.0f - wc);
                        if constexpr (BASIS >= 0) {
                            o0 += weight / res.y;
                            o1 += weight / res.z;
                            o2 += weight / res.w;
                        } else {
                            o0 += weight * res.y;
                            o1 += weight * res.z;
                            o2 += weight * res.w;
                        }
                        ti *= wc;
                        first_pending = r_next[e];
                        ++n_eval;
                    }
                    seen = r;
                }
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
                    setup_ray<(BASIS > 0 ? BASIS : 0)>(P, *Cp, P.x0 + bx, P.y0 + by, r);
                    if constexpr (BASIS == 0) r.basis[0] = (0 < P.basis_min || 0 > P.basis_max) ? 0.f : (float)0.28209479177387814;
                    float true_dir[3], vdir[3];
                    world_ray_dirs(P, *Cp, P.x0 + bx, P.y0 + by, true_dir, vdir);
                    td0 = true_dir[0];
                    td1 = true_dir[1];
                    td2 = true_dir[2];
#pragma unroll
                    for (int k = 0; k < NB; ++k) my_ray[k * RB] = r.basis[k];
                    if (S.need_viewdir) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) my_ray[(NB + k) * RB] = vdir[k];
                    }
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
            // ---- nothing more can be pushed: ask for the rest of the ring to be evaluated and wait for it
            if (__ballot(has_ray && (!done || held)) == 0) {
                if (tail != seen) {
                    if (flush_sent != tail) {
                        st_release(c_flush, tail);
                        flush_sent = tail;
                    }
                    const unsigned long long t0 = F.diag ? wall_clock64() : 0;
                    __builtin_amdgcn_s_sleep(2);
                    if (F.diag) t_flush += wall_clock64() - t0 + 1;
                    if (++spins > kF2SpinLimit) break;  // watchdog: never hang the device (see kF2SpinLimit)
                }
                continue;  // tail == seen: every lane's chain is empty, the pixels go out at the top of the next iteration
            }
            // ---- room for one more step's samples (at most 64)?
            if (tail - seen > (uint32_t)(kF2Ring - 64)) {
                const unsigned long long t0 = F.diag ? wall_clock64() : 0;
                __builtin_amdgcn_s_sleep(1);
                if (F.diag) t_ring += wall_clock64() - t0 + 1;
                if (++spins > kF2SpinLimit) break;
                continue;
            }
            spins = 0;

            // ---- one march step (rt_core.cuh:452-560) for the lanes whose ray is still under way
            bool fresh = false;  // this step emitted a sample
            float sz = 0.f, sx = 0.f, sy = 0.f, sw = 0.f;
            int scl = -1;
            ++n_steps;
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
                        int sh = sh1;
                        if (L2 > LL) {
                            const int LB = L2 - 2;
                            uint32_t gi = q[0] >> (sh2 + 2);
                            gi = (gi << LB) | (q[1] >> (sh2 + 2));
                            gi = (gi << LB) | (q[2] >> (sh2 + 2));
                            gi = (gi << 2) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh2, 2u);
                            gi = (gi << 2) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh2, 2u);
                            gi = (gi << 2) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh2, 2u);
                            word = A.grid2[gi];
                            sh = sh2;
                            src = 1;
                            vox = gi;
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
                            sx = m[9] + td0 * sz;
                            sy = m[10] + td1 * sz;
                            sw = m[11] + td2 * sz;
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

            // ---- release complete samples into the ring: the held one once its successor exists (delta z known), or as the ray's
            //      last sample one step after the ray ended
            {
                bool push = false, last = false;
                float px_ = 0.f, py_ = 0.f, pw_ = 0.f, pdz = 0.f;
                int pcl = -1;
                if (fresh) {
                    if (held) {
                        push = true;
                        px_ = hx; py_ = hy; pw_ = hw; pcl = hcl;
                        pdz = sz - hz;  // delta_i = z[i + 1] - z[i], rt_core.cuh:359
                    }
                    hz = sz; hx = sx; hy = sy; hw = sw; hcl = scl;
                    held = true;
                } else if (has_ray && done && held) {
                    push = true;
                    last = true;
                    px_ = hx; py_ = hy; pw_ = hw; pcl = hcl;
                    held = false;
                }
                const uint64_t pm = __ballot(push);
                if (pm != 0) {
                    if (push) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                        const uint32_t slot = tail + rank, e = slot & (kF2Ring - 1);
                        r_data[e] = make_float4(px_, py_, pw_, pdz);
                        r_meta[e] = (uint32_t)lane | (last ? 64u : 0u) | ((uint32_t)(pcl & 0xffff) << 8);
                        r_next[e] = kNone;
                        if (first_pending == kNone) first_pending = slot;
                        else r_next[prev_slot & (kF2Ring - 1)] = slot;  // the chain's last entry is still in the ring: nobody but its owner frees it
                        prev_slot = slot;
                    }
                    tail += (uint32_t)__popcll(pm);
                    st_release(c_tail, tail);
                }
            }
        }
        st_release(c_exit, 1u);
        if (spins > kF2SpinLimit && F.diag && lane == 0) atomicAdd(F.diag + F2Diag::kWatchdog, 1ull);
        if (F.sample_counter) {
            int tot = n_eval;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
            if (lane == 0 && tot) atomicAdd(F.sample_counter, (unsigned long long)tot);
        }
        if (F.diag && lane == 0) {
            atomicAdd(F.diag + F2Diag::kProdSimd + (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3), 1ull);  // HW_ID[5:4] = SIMD_ID
            atomicAdd(F.diag + F2Diag::kSteps, (unsigned long long)n_steps);
            atomicAdd(F.diag + F2Diag::kProdTotal, wall_clock64() - t_begin);
            atomicAdd(F.diag + F2Diag::kProdRingWait, t_ring);
            atomicAdd(F.diag + F2Diag::kProdFlushWait, t_flush);
        }
    } else {
        // =================================================================================== consumer: the network
        __builtin_amdgcn_s_setprio(2);
        const int g = lane >> 4, col = lane & 15;
        half8 *s_frag = reinterpret_cast<half8 *>(s_mem + Lo.frags);
        float *s_bias = reinterpret_cast<float *>(s_mem + Lo.bias);
        uint32_t *s_tile = s_mem + Lo.tile;
        float *s_out = reinterpret_cast<float *>(s_tile);
        uint32_t rdy[kF2NP];
#pragma unroll
        for (int p = 0; p < kF2NP; ++p) rdy[p] = 0u;
        int lds_cluster = -1;
        uint32_t spins = 0;
        unsigned long long n_runs = 0, n_windows = 0, n_reloads = 0, n_global = 0, n_cols = 0, t_busy = 0, t_enc = 0, t_lay = 0, t_eval = 0;

        for (;;) {
            // ---- what waits in the rings
            uint32_t pend[kF2NP];
            bool flush[kF2NP];
            uint32_t total = 0;
            bool any_flush = false, all_exited = true;
#pragma unroll
            for (int p = 0; p < kF2NP; ++p) {
                const uint32_t ex = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_ctrl + 4 * p + 3));  // read BEFORE the tail: a producer pushes nothing after it has left
                const uint32_t tl = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_ctrl + 4 * p));
                const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_ctrl + 4 * p + 1));
                pend[p] = tl - rdy[p];
                flush[p] = (int32_t)(fl - rdy[p]) > 0;
                total += pend[p];
                any_flush |= flush[p];
                all_exited &= ex != 0;
            }
            if (total == 0) {
                if (all_exited) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 64u * kF2SpinLimit) break;  // watchdog (a consumer legitimately idles through a whole tile of empty space)
                continue;
            }
            if (total < (uint32_t)F.batch_min && !any_flush) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 64u * kF2SpinLimit) break;
                continue;
            }
            spins = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const unsigned long long t_w0 = F.diag ? wall_clock64() : 0;
            // ---- the window: up to 64 entries, oldest first per ring; rings that asked for a flush come first
            uint32_t take[kF2NP], first_col[kF2NP];
            int n = 0;
#pragma unroll
            for (int p = 0; p < kF2NP; ++p) take[p] = first_col[p] = 0u;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int p = 0; p < kF2NP; ++p) {
                    if (flush[p] == (pass == 0) && pend[p] != 0u && n < 64) {
                        const uint32_t k = pend[p] < (uint32_t)(64 - n) ? pend[p] : (uint32_t)(64 - n);
                        take[p] = k;
                        first_col[p] = (uint32_t)n;
                        n += (int)k;
                    }
                }
            }
            ++n_windows;
            // column `lane` -> its ring and entry
            int ring = 0;
            uint32_t e = 0;
            bool col_on = false;
#pragma unroll
            for (int p = 0; p < kF2NP; ++p) {
                if (take[p] != 0u && (uint32_t)lane >= first_col[p] && (uint32_t)lane < first_col[p] + take[p]) {
                    ring = p;
                    e = (rdy[p] + ((uint32_t)lane - first_col[p])) & (kF2Ring - 1);
                    col_on = true;
                }
            }
            float4 *e_data = reinterpret_cast<float4 *>(s_mem + Lo.rings + ring * kF2RingWords) + e;
            const uint32_t meta = col_on ? reinterpret_cast<const uint32_t *>(s_mem + Lo.rings + ring * kF2RingWords + 4 * kF2Ring)[e] : 0u;
            const int owner_thread = ring * 64 + (int)(meta & 63u), my_cl = (int)(int16_t)(meta >> 8);
            const float4 smp = col_on ? *e_data : make_float4(0.f, 0.f, 0.f, 0.f);
            // The window's samples may belong to several sub-modules (a ray that crosses the front and the back of a surface changes
            // cluster on the way): the network runs once per distinct cluster of the window, the resident one first.
            uint64_t todo = __ballot(col_on);
            while (todo != 0) {
                int c_star;
                if (lds_cluster >= 0 && __ballot(col_on && my_cl == lds_cluster && ((todo >> lane) & 1ull)) != 0) c_star = lds_cluster;
                else c_star = __builtin_amdgcn_readfirstlane(__shfl(my_cl, (int)__builtin_ctzll(todo)));
                const bool col_sel = col_on && my_cl == c_star;
                const uint64_t sel = __ballot(col_sel);
                todo &= ~sel;
                ++n_runs;
                n_cols += (unsigned long long)__popcll(sel);
                const bool valid_cluster = c_star >= 0 && c_star < S.n_clusters;
                // ---- per column: transmittance factor and colour denominators of its sample (rt_core.cuh:356-392), SH basis of the
                //      owner; 32 columns at a time (the output tile holds 32), two lanes per column: lane share 0 takes the opacity and
                //      the first channel, share 1 the other two; the four floats replace the sample in its ring entry
                auto evaluate = [&](int half, auto valid_tag) {
                    constexpr bool kValid = decltype(valid_tag)::value;
                    const int c32 = lane & 31, share = lane >> 5, jc = half * 32 + c32;
                    const bool on = __shfl((int)col_sel, jc) != 0;
                    const uint32_t meta_j = (uint32_t)__shfl((int)meta, jc);
                    const int owner_j = __shfl(owner_thread, jc);
                    const float dz_j = __shfl(smp.w, jc);
                    const int ring_j = __shfl(ring, jc);
                    const uint32_t e_j = (uint32_t)__shfl((int)e, jc);
                    if (on) {
                        float *dst = reinterpret_cast<float *>(reinterpret_cast<float4 *>(s_mem + Lo.rings + ring_j * kF2RingWords) + e_j);
                        auto sv = [&](int f) -> float { return kValid ? s_out[f * 32 + c32] : 0.f; };  // no sub-module: zeros (mlp_histogram)
                        if constexpr (BASIS >= 0) {
                            float basis[NB];
#pragma unroll
                            for (int k = 0; k < NB; ++k) basis[k] = s_ray[k * RB + owner_j];
                            const int stride = BASIS > 0 ? BASIS : 0;
                            if (share == 0) {
                                const bool last = (meta_j & 64u) != 0;
                                dst[0] = last ? 0.f : exact_expf(-sv(3) * dz_j, s_exp);
                                dst[1] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, 0), s_exp);
                            } else {
                                dst[2] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, stride), s_exp);
                                dst[3] = 1.f + exact_expf(-sh_channel<BASIS>(basis, sv, 2 * stride), s_exp);
                            }
                        } else {
                            if (share == 0) {
                                const bool last = (meta_j & 64u) != 0;
                                dst[0] = last ? 0.f : exact_expf(-sv(3) * dz_j, s_exp);
                                dst[1] = sv(0);
                            } else {
                                dst[2] = sv(1);
                                dst[3] = sv(2);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // the next half (or the next cluster's encode) rewrites the tile
                };
                if (valid_cluster) {
                    f32x4 acc[MT][4];
                    const unsigned long long t_e0 = F.diag ? wall_clock64() : 0;
                    // ---- whose weights: the LDS copy (refilled when a sub-module takes over a window), or L2 for a minority's run
                    bool from_lds = c_star == lds_cluster;
                    if (!from_lds && (lds_cluster < 0 || __popcll(sel) >= F.switch_min)) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(F.frags + (size_t)c_star * S.frag_halfs);
                        uint4 *dst = reinterpret_cast<uint4 *>(s_frag);
                        const int n16 = S.frag_halfs / 8;
#pragma unroll 4
                        for (int i = lane; i < n16; i += 64) dst[i] = src[i];
                        const float *bsrc = F.biases + (size_t)c_star * S.bias_floats;
                        for (int i = lane; i < S.bias_floats; i += 64) s_bias[i] = bsrc[i];
                        __builtin_amdgcn_wave_barrier();
                        lds_cluster = c_star;
                        from_lds = true;
                        ++n_reloads;
                    }
                    if (!from_lds) ++n_global;
                    // ---- encode: lane j writes column j of the B operand (one 4 KB tile per K tile), as guided_fused_kernel does
                    {
                        float p[3], d[3];
                        p[0] = (smp.x - S.center[0]) * S.inv_extent[0];
                        p[1] = (smp.y - S.center[1]) * S.inv_extent[1];
                        p[2] = (smp.z - S.center[2]) * S.inv_extent[2];
#pragma unroll
                        for (int i = 0; i < 3; ++i) d[i] = S.need_viewdir ? s_ray[(NB + i) * RB + owner_thread] : 0.f;
                        const uint16_t *emb = nullptr;
                        if (S.n_embeddings > 0) {
                            int idx = (int)(float)F.appearance_embedding;
                            idx = idx < 0 ? 0 : (idx >= S.n_embeddings ? S.n_embeddings - 1 : idx);
                            emb = F.embeddings + ((size_t)c_star * S.n_embeddings + idx) * S.embedding_dim;
                        }
                        _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_tile);
                        auto put = [&](int f, float v) {  // f is wave-uniform
                            const int r = f & 31;
                            const int dw = ((((r & 15) >> 2) * 4 + (((r >> 4) * 4 + (r & 3)) >> 1)) * 64) + (f >> 5) * (16 * 64);
                            tile_h[(dw + lane) * 2 + (r & 1)] = (_Float16)v;  // element e = (r >> 4) * 4 + (r & 3): its low bit is r & 1
                        };
                        auto octaves = [&](int base, int n_oct, const float x[3]) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) put(base + i, x[i]);
                            for (int k = 0; k < n_oct; ++k) {
                                const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
#pragma unroll
                                for (int i = 0; i < 3; ++i) {
                                    put(base + 3 + 6 * k + i, tri_wave(x[i] * scale + 0.f));
                                    put(base + 3 + 6 * k + 3 + i, tri_wave(x[i] * scale + 0.25f));
                                }
                            }
                        };
                        octaves(0, S.pos_octaves, p);
                        if (S.need_viewdir) octaves(S.n_pos, S.dir_octaves, d);
                        const int emb_base = S.n_pos + S.n_dir;
                        for (int j = 0; j < S.embedding_dim; ++j) put(emb_base + j, half_bits_to_float(emb[j]));
                        for (int f = S.in_dim; f < 32 * NKK0; ++f) put(f, 0.f);  // padding features: finite (their weights are zero)
                        __builtin_amdgcn_wave_barrier();
                    }
                    // ---- the layers: weights = A operand, 16 samples of a column tile = B; a layer's C layout is the next layer's B layout
                    auto network = [&](const half8 *w, const float *b) {
                        auto bias_tile = [&](int mt) -> f32x4 { return *reinterpret_cast<const f32x4 *>(b + 16 * mt + 4 * g); };
#pragma unroll
                        for (int kk = 0; kk < NKK0; ++kk) {
                            half8 bf[4];
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                union {
                                    uint32_t u[4];
                                    half8 h;
                                } rd;
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) rd.u[q4] = s_tile[kk * (16 * 64) + (g * 4 + q4) * 64 + nt * 16 + col];
                                bf[nt] = rd.h;
                            }
                            if (kk == 0) {
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) {
                                    const f32x4 bv = bias_tile(mt);
#pragma unroll
                                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bv;
                                }
                            }
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const half8 a = w[(mt * NKK0 + kk) * 64 + lane];
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[nt], acc[mt][nt], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the outputs
                        if (F.diag) t_enc += wall_clock64() - t_e0;
                        w += MT * NKK0 * 64;
                        b += 16 * MT;
                        for (int layer = 1; layer <= S.hidden_layers; ++layer) {
                            const int n_mt = layer < S.hidden_layers ? MT : S.mt_out;
                            half8 bf[MT / 2][4];
#pragma unroll
                            for (int kk = 0; kk < MT / 2; ++kk)
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt) bf[kk][nt] = relu_pack(acc[2 * kk][nt], acc[2 * kk + 1][nt]);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                if (mt < n_mt) {
                                    const f32x4 bv = bias_tile(mt);
#pragma unroll
                                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bv;
#pragma unroll
                                    for (int kk = 0; kk < MT / 2; ++kk) {
                                        const half8 a = w[(mt * (MT / 2) + kk) * 64 + lane];
#pragma unroll
                                        for (int nt = 0; nt < 4; ++nt)
                                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[kk][nt], acc[mt][nt], 0, 0, 0);
                                    }
                                }
                            }
                            w += n_mt * (MT / 2) * 64;
                            b += 16 * n_mt;
                        }
                    };
                    if (from_lds) network(s_frag, s_bias);
                    else network(reinterpret_cast<const half8 *>(F.frags + (size_t)c_star * S.frag_halfs), F.biases + (size_t)c_star * S.bias_floats);
                    if (F.diag) t_lay += wall_clock64() - t_e0;
                    const unsigned long long t_c0 = F.diag ? wall_clock64() : 0;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (mt < S.mt_out) {
#pragma unroll
                                for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) s_out[(16 * mt + 4 * g + r) * 32 + nn * 16 + col] = acc[mt][2 * half + nn][r];
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        evaluate(half, std::true_type{});
                    }
                    if (F.diag) t_eval += wall_clock64() - t_c0;
                } else {
                    evaluate(0, std::false_type{});
                    evaluate(1, std::false_type{});
                }
            }
            // ---- publish: the owners may composite these entries (and their slots come free once they have)
#pragma unroll
            for (int p = 0; p < kF2NP; ++p) {
                if (take[p] != 0u) {
                    rdy[p] += take[p];
                    st_release(s_ctrl + 4 * p + 2, rdy[p]);
                }
            }
            if (F.diag) t_busy += wall_clock64() - t_w0;
        }
        if (F.diag && lane == 0) {
            if (spins > 64u * kF2SpinLimit) atomicAdd(F.diag + F2Diag::kWatchdog, 1ull);
            atomicAdd(F.diag + F2Diag::kConsSimd + (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3), 1ull);
            atomicAdd(F.diag + F2Diag::kRuns, n_runs);
            atomicAdd(F.diag + F2Diag::kWindows, n_windows);
            atomicAdd(F.diag + F2Diag::kReloads, n_reloads);
            atomicAdd(F.diag + F2Diag::kGlobalRuns, n_global);
            atomicAdd(F.diag + F2Diag::kColumns, n_cols);
            atomicAdd(F.diag + F2Diag::kConsBusy, t_busy);
            atomicAdd(F.diag + F2Diag::kConsTotal, wall_clock64() - t_begin);
            atomicAdd(F.diag + F2Diag::kEnc, t_enc);
            atomicAdd(F.diag + F2Diag::kLayers, t_lay);
            atomicAdd(F.diag + F2Diag::kEval, t_eval);
        }
    }
}

}  // namespace mnv
