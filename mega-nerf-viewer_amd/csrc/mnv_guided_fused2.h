// mnv_guided_fused2.h -- the guided-sampling frame as ONE kernel with SPECIALISED wavefronts (BASELINE.json configs[4]).
// Included by mnv_accel_fused.hip after mnv_guided_fused.h, whose kernel (one wavefront does both jobs) it replaces wherever the
// network's weights fit a workgroup's LDS; that kernel stays as the path for deeper networks and as a second checker.
//
// Same four reference steps (src/renderer/cuda_renderer.cpp:107-139: get_samples_from_voxels rt_core.cuh:418-576, cumsum / masks,
// query_submodules :165-203, render_nerf_results rt_core.cuh:334-416), same arithmetic, same frames bit for bit -- but the march and
// the network no longer share one wavefront's registers and time:
//   * a workgroup is NP PRODUCER wavefronts + NC CONSUMER wavefronts (8 + 8: a whole CU's worth at 128 VGPRs).  A producer marches an
//     8x8 tile, one ray per lane, on the packed accel exactly as guided_fused_kernel does and pushes every complete sample (world
//     position, delta z, owner lane, cluster) into ITS ring in LDS (256 slots); it never touches the matrix cores.  The composite of a
//     ray's samples happens in the owning lane, in ray order, as results arrive (a per-entry READY flag; one LDS poll per march iteration);
//   * a consumer serves the ring(s) of its producer(s).  It keeps, per lane, the state of RING / 64 slots of every ring (waiting? which
//     sub-module? -- in LDS between windows), picks a sub-module -- the one it ran last or one whose weights sit in a slot while plenty of
//     its samples wait, otherwise that of the oldest waiting sample of the ring that asked for service -- gathers up to 64 waiting samples
//     OF THAT SUB-MODULE by ballot (so a run fills its columns although rays change sub-module between the front and the back of a
//     surface), takes the weights from a cache of NS slots in LDS shared by the workgroup's consumers (lock word, readers count, least
//     recently used victim; 16 KB from L2 on a miss), encodes, runs the network on the matrix cores (the identical
//     v_mfma_f32_16x16x32_f16 sequence on the identical fragments as mlp_forward_kernel), turns every column into the sample's
//     transmittance factor and colour denominators (SH basis of the owning ray from LDS; two lanes per column) and writes those four
//     floats over the sample's ring entry;
//   * rings, results and the counters per ring (pushed / flush request / evaluated / producer has left / stall request) live in LDS;
//     waves of one workgroup are co-resident by construction, so the spin-waits (s_sleep, under a watchdog) cannot deadlock: a producer
//     waits only after it has asked for service (its ring cannot take another step's samples, or its tile is finished), and the consumer
//     serves such a ring's oldest sample first; a consumer waits for another consumer only while that one fills a weight slot.
// Debug builds: -DMNV_F2_CHECK_RINGS / _WEIGHTS / _TILE (protocol, weight-slot and output-tile self-checks into the diagnostics words 28-30),
// -DMNV_F2_LOG (per-sample logs, tools/fused_log_diff.py); tools/fused_stress.py is the gate for every change to this file.
// The whole library is built WITHOUT VOP3P packed-FP32 instructions (Makefile: NOPK).  Round 3's "rare wrong colour denominator" of
// this kernel (one frame in ~100: the channel-1 sum of <= 16 columns of a window lacked its eighth term) was one such instruction in
// the column evaluation -- v_pk_add_f32 v[42:43], v[46:47], v[42:43] op_sel:[0,1] op_sel_hi:[1,0], formed by the SLP vectoriser from
// sh_channel's scalar sums -- whose low half read the HIGH dword of its second source as 0.0 in lanes 48-63, sporadically, beside the
// co-resident consumers' MFMAs (LAB_NOTEBOOK.md, "the rare wrong denominator: cause"; tools/f2lab/).
#pragma once

#include <type_traits>

#include "mnv_guided_fused.h"

#pragma clang fp contract(off)

namespace mnv {

// Shape of a workgroup: NP producer wavefronts, NC consumer wavefronts (consumer c serves the rings of producers c * NP / NC ...),
// NS weight slots in LDS shared by the consumers, rings of MNV_F2_RING slots.  The workgroup is a whole CU's worth of wavefronts at
// 128 VGPRs.  Measured on cfg2 at 1080p (tools/f2_variants.sh, LAB_NOTEBOOK.md): 8 + 8 with 256-slot rings and three weight slots
// (158.5 KB of LDS) 1.24 ms, with two 1.31; 10 + 5 (256 slots, 2 weight slots: a third does not fit) 1.38; 12 + 4 (256, 2) 1.57;
// 8 + 8 with 128-slot rings and four weight slots 1.45; the one-role kernel 1.43.  Deep rings matter more than resident weights: the rays
// of an 8x8 tile reach a surface together, so a producer emits its samples in bursts that a 128-slot ring cannot absorb.
#ifndef MNV_F2_NP
#define MNV_F2_NP 8
#endif
#ifndef MNV_F2_NC
#define MNV_F2_NC 8
#endif
#ifndef MNV_F2_NS
#define MNV_F2_NS 3  // as many as fit beside the rings: launch_accel takes fewer when a network's fragments are larger
#endif
#ifndef MNV_F2_COLS
#define MNV_F2_COLS 64  // columns (samples) of a network run: 64, or 32 (half the accumulators: a smaller register budget, more wavefronts)
#endif
#ifndef MNV_F2_WAVES
#define MNV_F2_WAVES 4  // wavefronts per SIMD the kernel is compiled for: 4 (128 VGPRs), 5 (96), 6 (80)
#endif
#ifndef MNV_F2_CONS_PRIO
#define MNV_F2_CONS_PRIO 0  // s_setprio of the network wavefronts (2 until round 4: their windows 6.6 instead of 8.6 us, but the march -- the
                            // bound of the 8 + 8 shape -- 3 % slower: 1.223 against 1.194 ms per frame) ...
#endif
#ifndef MNV_F2_PROD_PRIO
#define MNV_F2_PROD_PRIO 0  // ... and of the marching ones
#endif
#ifndef MNV_F2_SHARE
#define MNV_F2_SHARE 1  // consumers that serve one group of rings together (1, 2 or 4): slot s of a ring of the group belongs to the
#endif                  // consumer s % SHARE of the group -- a tile's burst of samples is drained by SHARE consumers instead of one
constexpr int kF2NP = MNV_F2_NP, kF2NC = MNV_F2_NC, kF2NS = MNV_F2_NS, kF2SH = MNV_F2_SHARE, kF2ShLog = kF2SH == 4 ? 2 : kF2SH - 1;
constexpr int kF2RPC = kF2NP * kF2SH / kF2NC;  // RPC: rings a consumer watches (those of its group)
constexpr int kF2Cols = MNV_F2_COLS, kF2NT = kF2Cols / 16, kF2Halves = kF2Cols / 32;
static_assert(kF2Cols == 64 || kF2Cols == 32, "a run is 64 or 32 columns");
constexpr int kF2Block = 64 * (kF2NP + kF2NC);
#ifndef MNV_F2_RING
#define MNV_F2_RING 256
#endif
constexpr int kF2Ring = MNV_F2_RING;               // slots per producer ring (128 or 256): one march step adds at most 64 samples
constexpr int kF2RingLog = kF2Ring == 256 ? 8 : 7, kF2RH = kF2Ring / 64 / kF2SH;  // RH: slots of a ring that one consumer lane watches
constexpr int kF2WCL = (kF2RH + 1) / 2;  // words of two 16-bit sub-module ids per lane and ring
static_assert(kF2Ring == 128 || kF2Ring == 256, "ring size");
#ifdef MNV_F2_LOG
constexpr int kF2RingWords = kF2Ring * (4 + 1 + 1 + 1);  // + a debug word per entry (where in which window the consumer evaluated it)
#else
constexpr int kF2RingWords = kF2Ring * (4 + 1 + 1);
#endif  // float4 {x, y, z, dz} -> {att, d0, d1, d2} | meta | owner's next slot
constexpr int kF2WavesPerSimd = MNV_F2_WAVES;      // register budget: 128 / 96 / 80 VGPRs
#ifndef MNV_F2_L0_BLOCKS
#define MNV_F2_L0_BLOCKS 1   // layer 0 one 16-row block at a time (a compiler barrier): see the comment at the barrier
#endif
#ifndef MNV_F2_OPAQUE_LANE
#define MNV_F2_OPAQUE_LANE 1 // lane-derived addresses are recomputed per window instead of being hoisted out of the loop and spilled
#endif
constexpr bool kF2Default = true;                  // mnv_set_fused_kernel(0) picks this kernel when it fits
constexpr uint32_t kF2Ready = 128u;                // meta bit: the entry holds its results
static_assert((kF2NP * kF2SH) % kF2NC == 0 && kF2NC % kF2SH == 0 && (kF2SH == 1 || kF2SH == 2 || kF2SH == 4) && kF2RH >= 1 && kF2RPC >= 1 && kF2RPC <= 4 &&
                  kF2Block <= 1024 && kF2NS >= 1 && kF2NS <= 7,
              "workgroup shape");
// Watchdog of the spin-waits: a wait that lasts this many polls (s_sleep 1-2 each: tens of milliseconds; a healthy wait is a few
// microseconds) is abandoned and the wavefront leaves -- wrong pixels and a count in the diagnostics buffer instead of a hung device.
// No schedule of co-resident waves reaches it (header comment); it exists so that a bug cannot take the machine down.
constexpr uint32_t kF2SpinLimit = 1u << 18;

struct F2Layout {  // word offsets into the dynamic LDS block
    int grid, ray, rings, ctrl, cols, watch, wcache, frags, bias, tile, total;
    int ray_rows, frag_words, bias_words, tile_words;
};
__host__ __device__ inline F2Layout f2_layout(int nb, int lds_level, const MlpShape &S, int slots) {
    F2Layout L;
    L.ray_rows = nb + (S.need_viewdir ? 3 : 0);
    L.grid = 64;                                          // after the exp table
    L.ray = L.grid + (1 << (3 * lds_level));
    L.rings = (L.ray + L.ray_rows * kF2NP * 64 + 3) & ~3;  // 16-byte aligned entries
    L.ctrl = L.rings + kF2NP * kF2RingWords;
    L.cols = L.ctrl + 8 * kF2NP;                          // 8 words per ring
    L.watch = L.cols + 64 * kF2NC;                        // column -> (ring, slot) of every consumer's current window
    L.wcache = L.watch + (kF2RPC * kF2WCL + 1) * 64 * kF2NC;       // per consumer and lane: what waits in the slots the lane watches (between windows the registers belong to the network)
    L.frags = L.wcache + 32;                              // weight cache: lock, clock, per slot {cluster, state, readers, stamp}
    L.frag_words = S.frag_halfs / 2;
    L.bias_words = (S.bias_floats + 3) & ~3;
    L.bias = L.frags + slots * L.frag_words;
    L.tile = L.bias + slots * L.bias_words;
    const int enc = S.nkk0 * 16 * kF2Cols, out = 16 * S.mt_out * 32;  // encode tiles (one per K tile); outputs of 32 columns at a time
    L.tile_words = enc > out ? enc : out;
    L.total = L.tile + kF2NC * L.tile_words;
    return L;
}

struct F2Diag {  // mnv_set_fused_diag: sums over wavefronts, 100 MHz ticks
    enum { kRuns = 1, kSteps, kWindows, kReloads, kGlobalRuns, kConsBusy, kConsTotal, kProdTotal, kProdRingWait, kProdFlushWait, kEnc, kLayers, kEval, kColumns, kWatchdog,
           kConsSimd /* 4 words: consumer wavefronts per SIMD id */, kProdSimd = kConsSimd + 4, kProdWalk = kProdSimd + 4, kProdSetup, kProdStep, kProdPush, kWords };
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
    const F2Layout Lo = f2_layout(NB, LL, S, F.weight_slots);
    uint32_t *s_grid = s_mem + Lo.grid;
    float *s_ray = reinterpret_cast<float *>(s_mem + Lo.ray);  // [row][producer thread]: SH basis, then the view direction
    uint32_t *s_ctrl = s_mem + Lo.ctrl;                        // per ring (8 words): pushed, flush request, evaluated, producer has left, stall request
    constexpr uint32_t kNone = 0xffffffffu;

    {
        const int cells = 1 << (3 * LL);
        if (threadIdx.x < 32) s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
        for (int i = threadIdx.x; i < 8 * kF2NP; i += kF2Block) s_ctrl[i] = 0u;
        for (int i = threadIdx.x; i < (kF2RPC * kF2WCL + 1) * 64 * kF2NC; i += kF2Block) s_mem[Lo.watch + i] = 0u;
        if (threadIdx.x < 32) s_mem[Lo.wcache + threadIdx.x] = threadIdx.x >= 2 && ((threadIdx.x - 2) & 3) == 0 ? 0xffffffffu : 0u;  // slots: no cluster
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
        if (MNV_F2_PROD_PRIO) __builtin_amdgcn_s_setprio(MNV_F2_PROD_PRIO);
        float4 *r_data = reinterpret_cast<float4 *>(s_mem + Lo.rings + wave * kF2RingWords);
        uint32_t *r_meta = reinterpret_cast<uint32_t *>(r_data + kF2Ring), *r_next = r_meta + kF2Ring;
        uint32_t *c_tail = s_ctrl + 8 * wave, *c_flush = c_tail + 1, *c_ready = c_tail + 2, *c_exit = c_tail + 3, *c_stall = c_tail + 4;
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
        // TRACK: the trackers' fallback leaves (the last leaf without a dense sample that qualifies, rt_core.cuh:561-574) by the t of their step;
        // named, and their sample count read, once per ray when the pixel is written (march_accel_kernel does the same: mnv_march_accel_kernel.h)
        [[maybe_unused]] float sp_t = -1.f, sa_t = -1.f, ray_tmin = 0.f;
        int n_eval = 0, n_steps = 0;
#ifdef MNV_F2_LOG
        int n_comp = 0, n_pushed = 0;
#endif
        unsigned long long t_ring = 0, t_flush = 0, t_walk = 0, t_setup = 0, t_step = 0, t_push = 0;
        uint32_t spins = 0;                           // consecutive waits (watchdog)
        uint32_t tail = 0, seen = 0, flush_sent = 0xffffffffu, stall_sent = 0xffffffffu, head = 0;  // wave-uniform: entries pushed; evaluated entries (counter last seen); last service request; lower bound of the oldest slot in use

        const uint32_t home = blockIdx.x % kNumQueues;
        uint32_t qsel = 0;
        bool drained = false;
        const CamBlock *__restrict__ Cp = K.cams;
        const float cen0 = Cp->cen[0], cen1 = Cp->cen[1], cen2 = Cp->cen[2];

        for (;;) {
            // ---- results that arrived since the last look: every owner walks its samples in ray order (rt_core.cuh:356-392) as far as
            //      they have been evaluated (entries of other sub-modules may still wait: evaluation is not first-in first-out)
            const unsigned long long t_i0 = F.diag ? wall_clock64() : 0;
            {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(c_ready));
                if (r != seen) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    while (first_pending != kNone) {
                        const uint32_t e = first_pending & (kF2Ring - 1);
                        // The results must be read AFTER the flag.  LDS serves a wavefront's requests in program order, so it is enough that
                        // the compiler keeps the three reads in this order (the barrier); they travel together instead of one round trip each.
                        const uint32_t m = __hip_atomic_load(&r_meta[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        asm volatile("" ::: "memory");
                        const float4 res = r_data[e];
                        const uint32_t nx = r_next[e];
                        if (!(m & kF2Ready)) break;
                        const bool last = (m & 64u) != 0;
#ifdef MNV_F2_LOG
                        if (F.diag && F.diag[31] && n_comp < 39) {
                            float *rec = reinterpret_cast<float *>(F.diag[31]) + ((size_t)pix * 40 + n_comp) * 8;
                            rec[0] = res.x; rec[1] = res.y; rec[2] = res.z; rec[3] = res.w;
                            rec[4] = __uint_as_float(m); rec[5] = __uint_as_float(first_pending); rec[6] = ti; rec[7] = __uint_as_float(r_next[kF2Ring + e] & 63u);
                        }
                        ++n_comp;
#endif
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
                        first_pending = nx;
                        ++n_eval;
                    }
                    seen = r;
                }
            }
            // ---- rays that have ended and whose samples are all composited: write the pixel (alpha 1, renderer_kernel.cu:316)
            if (has_ray && done && !held && first_pending == kNone) {
                composite_and_write(P, (int64_t)pix, o0, o1, o2, 1.0f);
#ifdef MNV_F2_LOG
                if (F.diag && F.diag[31]) {
                    float *rec = reinterpret_cast<float *>(F.diag[31]) + ((size_t)pix * 40 + 39) * 8;
                    rec[0] = (float)n_comp; rec[1] = o0; rec[2] = o1; rec[3] = o2; rec[4] = ti; rec[5] = (float)ns;
                }
#endif
                if constexpr (TRACK) {
                    {
                        const int Lq_ = A.max_depth, shg_ = Lq_ - A.grid_level;
                        auto leaf_at = [&](float tw, uint32_t &word, uint32_t &v, float &dt) {
                            float pos[3];
                            uint32_t q[3];
                            pos[0] = cen0 + tw * dir0;
                            pos[1] = cen1 + tw * dir1;
                            pos[2] = cen2 + tw * dir2;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                pos[i] = __builtin_amdgcn_fmed3f(pos[i], 0.f, 1.f - 1e-6f);
                                q[i] = (uint32_t)(pos[i] * qscale);
                            }
                            word = s_grid[((((q[0] >> sh1) << LL) | (q[1] >> sh1)) << LL) | (q[2] >> sh1)];
                            int src = 0;
                            v = 0;
                            if (!(word & kLeafBit)) word = descend_to_leaf(A, q, word, sh1, sh2, L2, LL, src, v);
                            if (src == 0) v = A.grid_vox[((((q[0] >> shg_) << A.grid_level) + (q[1] >> shg_)) << A.grid_level) + (q[2] >> shg_)];
                            else if (src == 1) v = A.grid2_vox[v];
                            const int depth = (int)((word >> 16) & 0x7fu);
                            const float sc = __uint_as_float((uint32_t)(127 + depth) << 23), inv_cube = __uint_as_float((uint32_t)(127 - depth) << 23);
                            const float invd[3] = {inv0, inv1, inv2};
                            float tu = 1e4f;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const float x = __builtin_amdgcn_fractf(pos[i] * sc);
                                const float t1 = -x * invd[i];
                                const float t2 = t1 + invd[i];
                                tu = fminf(tu, fmaxf(t1, t2));
                            }
                            dt = tu * inv_cube + P.step_size;
                        };
                        const bool want_split = max_weight == -1.f && sp_t >= 0.f, want_sample = K.sample_counts && max_sample_weight == -1.f && sa_t >= 0.f;
                        uint32_t word = 0, v = 0;
                        float dt = 0.f;
                        if (want_split) {
                            leaf_at(sp_t, word, v, dt);
                            sp_vox = (int32_t)v;
                            sp_prio = (float)((word >> 16) & 0x7fu);
                        }
                        if (want_sample) {
                            if (!(want_split && sa_t == sp_t)) leaf_at(sa_t, word, v, dt);
                            const int16_t sc_last = K.sample_counts[v];
                            if (sc_last < K.max_sample_count) {
                                sa_vox = (int32_t)v;
                                sa_prio = (float)sc_last;
                            } else {  // the last such leaf is saturated: the ray's steps once more, for the last one that is not
                                float tw = ray_tmin;
                                while (tw < t) {
                                    leaf_at(tw, word, v, dt);
                                    if (!(half_bits_to_float((uint16_t)word) > P.sigma_thresh)) {
                                        const int16_t c = K.sample_counts[v];
                                        if (c < K.max_sample_count) {
                                            sa_vox = (int32_t)v;
                                            sa_prio = (float)c;
                                        }
                                    }
                                    tw += dt;
                                }
                            }
                        }
                    }
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
            const unsigned long long t_i1 = F.diag ? wall_clock64() : 0;
            if (F.diag) t_walk += t_i1 - t_i0;
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
#ifdef MNV_F2_LOG
                    n_comp = n_pushed = 0;
#endif
                    ti = 1.f;
                    o0 = o1 = o2 = 0.f;
                    if constexpr (TRACK) {
                        max_weight = max_sample_weight = -1.f;
                        sp_prio = (float)(K.max_depth + 1);
                        sa_prio = (float)(K.max_sample_count + 1);
                        sp_vox = sa_vox = -1;
                        sp_t = sa_t = -1.f;
                    }
                    RaySetup<NB> r;
                    setup_ray<(BASIS > 0 ? BASIS : 0)>(P, *Cp, P.x0 + bx, P.y0 + by, r, frame_tmax(P, p));
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
                        if constexpr (TRACK) ray_tmin = r.tmin;
                    }
                }
                if (F.diag) t_setup += wall_clock64() - t_i1;
                continue;
            }
            // ---- nothing more can be pushed: ask for the rest of the ring to be evaluated and wait for it
            if (__ballot(has_ray && (!done || held)) == 0) {
                if (tail != seen) {  // `seen` counts evaluated entries: all of them once it reaches `tail`
                    if (flush_sent != tail) {
                        st_release(c_flush, tail);
                        flush_sent = tail;
                    }
                    const unsigned long long t0 = F.diag ? wall_clock64() : 0;
                    __builtin_amdgcn_s_sleep(2);
                    if (F.diag) t_flush += wall_clock64() - t_i0;
                    (void)t0;
                    if (++spins > kF2SpinLimit) break;  // watchdog: never hang the device (see kF2SpinLimit)
                }
                continue;  // tail == seen: every lane's chain is empty, the pixels go out at the top of the next iteration
            }
            // ---- room for one more step's samples (at most 64)?  A slot is free once its owner has composited it; rays composite in
            //      order, so the oldest slot in use is the smallest `first_pending` of the wavefront
            if (tail - head > (uint32_t)(kF2Ring - 64)) {
                uint32_t age = first_pending != kNone ? tail - first_pending : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)age, o);
                    age = other > age ? other : age;
                }
                head = tail - (uint32_t)__builtin_amdgcn_readfirstlane((int)age);
                if (tail - head > (uint32_t)(kF2Ring - 64)) {
                    // the ring cannot take another step: ask for service (the consumer then evaluates this ring's oldest entries first)
                    if (stall_sent != tail) {
                        st_release(c_stall, tail);
                        stall_sent = tail;
                    }
                    const unsigned long long t0 = F.diag ? wall_clock64() : 0;
                    __builtin_amdgcn_s_sleep(1);
                    if (F.diag) t_ring += wall_clock64() - t_i0;
                    (void)t0;
                    if (++spins > kF2SpinLimit) break;
                    continue;
                }
            }
            spins = 0;

            // ---- one march step (rt_core.cuh:452-560) for the lanes whose ray is still under way
            bool fresh = false;  // this step emitted a sample
            float sz = 0.f, sx = 0.f, sy = 0.f, sw = 0.f;
            int scl = -1;
            ++n_steps;
            const unsigned long long t_i2 = F.diag ? wall_clock64() : 0;
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
                        need_vox = is_dense || K.visited != nullptr;  // (leaves without a dense sample: by their t, sp_t / sa_t)
                        if (need_vox) {
                            const int shg = Lq - A.grid_level;
                            if (src == 0) vox = A.grid_vox[((((q[0] >> shg) << A.grid_level) + (q[1] >> shg)) << A.grid_level) + (q[2] >> shg)];
                            else if (src == 1) vox = A.grid2_vox[vox];
                            if (K.visited && __hip_atomic_load(&K.visited[vox >> 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) K.visited[vox >> 3] = 1;
                        }
                        if (!is_dense) {  // the last leaf before any dense one qualifies, rt_core.cuh:561-574
                            if (depth < K.max_depth && max_weight == -1.f) sp_t = t;
                            if (max_sample_weight == -1.f) sa_t = t;
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

            const unsigned long long t_i3 = F.diag ? wall_clock64() : 0;
            if (F.diag) t_step += t_i3 - t_i2;
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
#ifdef MNV_F2_CHECK_RINGS
                        if (slot >= (uint32_t)kF2Ring && !(r_meta[e] & kF2Ready) && F.diag) atomicAdd(F.diag + 29, 1ull);  // overwrites an entry that was never evaluated
#endif
                        r_data[e] = make_float4(px_, py_, pw_, pdz);
                        r_meta[e] = (uint32_t)lane | (last ? 64u : 0u) | ((uint32_t)(pcl & 0xffff) << 8);
                        r_next[e] = kNone;
#ifdef MNV_F2_LOG
                        r_next[kF2Ring + e] = pix * 64u + (uint32_t)(n_pushed++ & 63);
#endif
                        if (first_pending == kNone) first_pending = slot;
                        else r_next[prev_slot & (kF2Ring - 1)] = slot;  // the chain's last entry is still in the ring: nobody but its owner frees it
                        prev_slot = slot;
                    }
                    tail += (uint32_t)__popcll(pm);
                    st_release(c_tail, tail);
                }
            }
            if (F.diag) t_push += wall_clock64() - t_i3;
        }
        st_release(c_exit, 1u);
        if (spins > kF2SpinLimit && lane == 0) {  // the frame is wrong: say so whether or not anybody collects diagnostics
            atomicAdd(F.fault, 1u);
            if (F.diag) atomicAdd(F.diag + F2Diag::kWatchdog, 1ull);
        }
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
            atomicAdd(F.diag + F2Diag::kProdWalk, t_walk);
            atomicAdd(F.diag + F2Diag::kProdSetup, t_setup);
            atomicAdd(F.diag + F2Diag::kProdStep, t_step);
            atomicAdd(F.diag + F2Diag::kProdPush, t_push);
        }
    } else {
        // =================================================================================== consumer: the network
        __builtin_amdgcn_s_setprio(MNV_F2_CONS_PRIO);
        // Lane-derived constants (tile / fragment / bias addresses) are loop invariants the compiler hoists out of the window loop -- into
        // registers the 64 accumulators leave no room for, i.e. into scratch, to be reloaded (a ~0.3 us round trip each, eight in a row in
        // the evaluation alone) in every window.  An opaque copy of the lane number per window makes them cheap values again.
        auto opaque = [](int v) __attribute__((always_inline)) {
#if MNV_F2_OPAQUE_LANE
            asm volatile("" : "+v"(v));
#endif
            return v;
        };
        int g = lane >> 4, col = lane & 15;
        const int ci = wave - kF2NP, rbase = (ci / kF2SH) * kF2RPC;  // this consumer and the first ring of its group
        const uint32_t par = (uint32_t)(ci % kF2SH);                  // ... of whose rings it watches the slots s with s % SH == par
        auto slot_of = [&](int h) __attribute__((always_inline)) -> uint32_t { return (uint32_t)kF2SH * ((uint32_t)lane + 64u * (uint32_t)h) + par; };  // the h-th slot this lane watches
        // entries with numbers in [a, b) that are this consumer's (numbers are monotonic, the ring size is a multiple of SH)
        auto mine_in = [&](uint32_t a, uint32_t b) __attribute__((always_inline)) -> uint32_t { return ((b + (uint32_t)(kF2SH - 1) - par) >> kF2ShLog) - ((a + (uint32_t)(kF2SH - 1) - par) >> kF2ShLog); };
        uint32_t *s_cols = s_mem + Lo.cols + 64 * ci;
        uint32_t *s_tile = s_mem + Lo.tile + ci * Lo.tile_words;
        float *s_out = reinterpret_cast<float *>(s_tile);
        uint32_t *s_wc = s_mem + Lo.wcache;  // [0] lock, [1] clock, then per slot {cluster, state (1: being filled), readers, stamp}
        uint32_t *s_rctrl = s_ctrl + 8 * rbase;
        auto ring_data = [&](int p) { return reinterpret_cast<float4 *>(s_mem + Lo.rings + (rbase + p) * kF2RingWords); };  // p: ring of this consumer
        auto ring_meta = [&](int p) { return s_mem + Lo.rings + (rbase + p) * kF2RingWords + 4 * kF2Ring; };
        // this lane watches slots `lane + 64 h` (h < RH) of every ring: is an unevaluated sample there, and of which sub-module
        // (kept in LDS between windows -- s_watch[k * 64 + lane]: first the cluster words (two 16-bit clusters each), then the waiting
        // bits -- so that nothing of it occupies registers while the network runs)
        constexpr int WCL = kF2WCL, WWORDS = kF2RPC * WCL + 1;
        static_assert(kF2RPC * kF2RH <= 32, "waiting bits fit a word");
        uint32_t *s_watch = s_mem + Lo.watch + ci * WWORDS * 64 + lane;
        uint32_t scan[kF2RPC], mine[kF2RPC], evald[kF2RPC];  // per ring (wave-uniform): entries seen; of those, this consumer's; of those, evaluated
#pragma unroll
        for (int p = 0; p < kF2RPC; ++p) scan[p] = mine[p] = evald[p] = 0u;
        int lds_cluster = -1;  // the sub-module this consumer ran last (its weights are most likely still in a slot)
        int held_slot = -1;    // the weight slot this consumer holds a reader's reference on
        uint32_t spins = 0;
        unsigned long long n_runs = 0, n_reloads = 0, n_cols = 0, t_busy = 0, t_enc = 0, t_lay = 0, t_eval = 0, t_reload = 0;

        for (;;) {
            // ---- what has arrived: register the new entries of every ring with their watcher lanes
            uint32_t watch_cl[kF2RPC][WCL];  // clusters of the slots this lane watches in ring p: slot lane + 64 h in half (h & 1) of word h >> 1
            uint32_t watch_pend;             // bit RH * p + h: slot lane + 64 h of ring p waits
#pragma unroll
            for (int p = 0; p < kF2RPC; ++p)
#pragma unroll
                for (int k = 0; k < WCL; ++k) watch_cl[p][k] = s_watch[(p * WCL + k) * 64];
            watch_pend = s_watch[kF2RPC * WCL * 64];
            auto cl_of = [&](int p, int h) __attribute__((always_inline)) -> int { return (int)(int16_t)((h & 1) ? watch_cl[p][h >> 1] >> 16 : watch_cl[p][h >> 1] & 0xffffu); };
            uint32_t total = 0;
            bool all_exited = true;
            int service = -1;  // a ring whose producer waits for it (tile finished, or no room for another step)
            bool service_stall = false;
            {
                uint32_t tl[kF2RPC], fl[kF2RPC], sl[kF2RPC];
#pragma unroll
                for (int p = 0; p < kF2RPC; ++p) {
                    all_exited &= __builtin_amdgcn_readfirstlane((int)ld_relaxed(s_rctrl + 8 * p + 3)) != 0;  // read BEFORE the tail: a producer pushes nothing after it has left
                    tl[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_rctrl + 8 * p));
                    fl[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_rctrl + 8 * p + 1));
                    sl[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_relaxed(s_rctrl + 8 * p + 4));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int p = 0; p < kF2RPC; ++p) {
                    const uint32_t fresh = tl[p] - scan[p];
                    if (fresh != 0u) {
                        const uint32_t *meta_p = ring_meta(p);
#pragma unroll
                        for (int h = 0; h < kF2RH; ++h) {
                            if (((slot_of(h) - scan[p]) & (kF2Ring - 1)) < fresh) {
#ifdef MNV_F2_CHECK_RINGS
                                if (((watch_pend >> (kF2RH * p + h)) & 1u) && F.diag) atomicAdd(F.diag + 29, 1ull << 32);  // the slot's previous entry still waits
                                if ((meta_p[slot_of(h)] & kF2Ready) && F.diag) atomicAdd(F.diag + 29, 1ull << 40);      // a fresh entry that is already marked
#endif
                                const uint32_t c16 = (meta_p[slot_of(h)] >> 8) & 0xffffu;
                                watch_cl[p][h >> 1] = (h & 1) ? (watch_cl[p][h >> 1] & 0xffffu) | (c16 << 16) : (watch_cl[p][h >> 1] & 0xffff0000u) | c16;
                                watch_pend |= 1u << (kF2RH * p + h);
                            }
                        }
                        mine[p] += mine_in(scan[p], tl[p]);
                        scan[p] = tl[p];
                    }
                    const uint32_t waiting = mine[p] - evald[p];
                    total += waiting;
                    // a stalled producer (no room for another step) needs its OLDEST samples; one that has finished its tile needs all of them
                    if (waiting != 0u && sl[p] == tl[p] && (service < 0 || !service_stall)) {
                        service = p;
                        service_stall = true;
                    } else if (service < 0 && waiting != 0u && fl[p] == tl[p]) {
                        service = p;
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < kF2RPC; ++p)
#pragma unroll
                for (int k = 0; k < WCL; ++k) s_watch[(p * WCL + k) * 64] = watch_cl[p][k];
            s_watch[kF2RPC * WCL * 64] = watch_pend;
            if (total == 0) {
                if (held_slot >= 0) {  // idle: let the others replace the slot
                    if (lane == 0) __hip_atomic_fetch_sub(s_wc + 2 + 4 * held_slot + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    held_slot = -1;
                }
                if (all_exited) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 64u * kF2SpinLimit) break;  // watchdog (a consumer legitimately idles through a whole tile of empty space)
                continue;
            }
            if (total < (uint32_t)F.batch_min && service < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 64u * kF2SpinLimit) break;
                continue;
            }
            spins = 0;
            const unsigned long long t_w0 = F.diag ? wall_clock64() : 0;
            {
                const int lw = opaque(lane);
                g = lw >> 4;
                col = lw & 15;
            }
            // ---- the window: up to 64 waiting samples of ONE sub-module, gathered from all rings of this consumer.  select(c, first, half)
            //      lists them in s_cols (ring `first` first, and of that ring the half `half` first) without taking them yet.
            uint32_t taken[kF2RPC];
            uint32_t sel_bits = 0u;  // the watched slots of this lane that the window takes (same layout as watch_pend)
            int n = 0;
            auto select = [&](int c_, int first_, uint32_t half_) __attribute__((always_inline)) {
                // (wave-uniform by construction; said again so that the compiler keeps them in scalar registers)
                const int c = __builtin_amdgcn_readfirstlane(c_), first = __builtin_amdgcn_readfirstlane(first_);
                const uint32_t half = (uint32_t)__builtin_amdgcn_readfirstlane((int)half_);
                n = 0;
                sel_bits = 0u;
#pragma unroll
                for (int p = 0; p < kF2RPC; ++p) taken[p] = 0u;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                    for (int p = 0; p < kF2RPC; ++p) {
                        if ((p == first) == (pass == 0)) {
#pragma unroll
                            for (int hh = 0; hh < kF2RH; ++hh) {
                                const int h = p == first ? (hh + (int)half) & (kF2RH - 1) : hh;  // wave-uniform
                                bool match = false;  // (a compile-time h inside: the cluster words are indexed statically)
#pragma unroll
                                for (int hc = 0; hc < kF2RH; ++hc)
                                    if (hc == h) match = ((watch_pend >> (kF2RH * p + hc)) & 1u) != 0u && cl_of(p, hc) == c;
                                const uint64_t msk = __ballot(match);
                                if (msk != 0 && n < kF2Cols) {
                                    const uint32_t k = (uint32_t)__popcll(msk), tk = k < (uint32_t)(kF2Cols - n) ? k : (uint32_t)(kF2Cols - n);
                                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(msk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)msk, 0u));
                                    if (match && rank < tk) {
                                        s_cols[n + (int)rank] = ((uint32_t)p << kF2RingLog) | slot_of(h);
                                        sel_bits |= 1u << (kF2RH * p + h);
                                    }
                                    n = __builtin_amdgcn_readfirstlane(n + (int)tk);
                                    taken[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(taken[p] + tk));
                                }
                            }
                        }
                    }
                }
            };
            // Which sub-module: the one this consumer ran last (its weights are in a slot) while plenty of its samples wait and nobody
            // needs service -- a finished tile's ring is also served with that sub-module first, as long as the ring has any;
            // otherwise the sub-module of the OLDEST waiting sample of the ring that needs service (or of the fullest ring).
            int c_star = -2;
            if (!service_stall) {
                // candidates: this consumer's last sub-module, then whatever the workgroup's weight slots hold (a run of those costs no refill)
#pragma unroll
                for (int k = -1; k < kF2NS; ++k) {
                    if (c_star != -2 || k >= F.weight_slots) break;
                    const int cand = k < 0 ? lds_cluster : __builtin_amdgcn_readfirstlane((int)ld_relaxed(s_wc + 2 + 4 * k));
                    if (cand < 0 || (k >= 0 && cand == lds_cluster)) continue;
                    select(cand, service, 0u);
                    uint32_t of_service = 0u;
#pragma unroll
                    for (int p = 0; p < kF2RPC; ++p)
                        if (p == service) of_service = taken[p];
                    if (service < 0 ? n >= F.switch_min : of_service > 0u) c_star = cand;
                }
            }
            if (c_star == -2) {
                int target = service;
                if (target < 0) {
                    uint32_t best = 0u;
#pragma unroll
                    for (int p = 0; p < kF2RPC; ++p) {
                        if (mine[p] - evald[p] > best) {
                            best = mine[p] - evald[p];
                            target = p;
                        }
                    }
                }
                uint32_t first_half = 0u;  // of the target ring: the 64-slot part that holds its oldest waiting sample
#pragma unroll
                for (int p = 0; p < kF2RPC; ++p) {
                    if (p == target) {
                        // every waiting slot of ring p lies within a ring's length below scan[p]: the oldest is the one furthest below
                        uint32_t far = 0u, far_h = 0u;
#pragma unroll
                        for (int h = 0; h < kF2RH; ++h) {
                            const uint32_t d = (watch_pend >> (kF2RH * p + h)) & 1u ? ((scan[p] - 1u - slot_of(h)) & (kF2Ring - 1)) + 1u : 0u;
                            if (d > far) {
                                far = d;
                                far_h = (uint32_t)h;
                            }
                        }
                        uint32_t top = far;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const uint32_t other = (uint32_t)__shfl_xor((int)top, o);
                            top = other > top ? other : top;
                        }
                        const int who = (int)__builtin_ctzll(__ballot(far == top && far != 0u));  // the lane that watches it
                        first_half = (uint32_t)__builtin_amdgcn_readlane((int)far_h, who);
#pragma unroll
                        for (int h = 0; h < kF2RH; ++h)
                            if ((uint32_t)h == first_half) c_star = __builtin_amdgcn_readlane(cl_of(p, h), who);
                    }
                }
                select(c_star, target, first_half);
            }
            s_watch[kF2RPC * WCL * 64] = watch_pend & ~sel_bits;
            __builtin_amdgcn_wave_barrier();
            const bool col_on = lane < n;
            ++n_runs;
            n_cols += (unsigned long long)n;
            const bool valid_cluster = c_star >= 0 && c_star < S.n_clusters;

            // ---- per column: transmittance factor and colour denominators of its sample (rt_core.cuh:356-392), SH basis of the
            //      owner; 32 columns at a time (the output tile holds 32), two lanes per column: lane share 0 takes the opacity and
            //      the first channel, share 1 the other two; the four floats replace the sample in its ring entry
            auto evaluate = [&](int half, auto valid_tag) __attribute__((always_inline)) {
                constexpr bool kValid = decltype(valid_tag)::value;
                const int le = opaque(lane);
                const int c32 = le & 31, share = le >> 5, jc = half * 32 + c32;
                const bool on = jc < n;
                if (on) {
                    const uint32_t where_j = s_cols[jc];
                    const int ring_j = (int)(where_j >> kF2RingLog);
                    const uint32_t e_j = where_j & (kF2Ring - 1);
                    const uint32_t meta_j = ring_meta(ring_j)[e_j];
                    const int owner_j = (rbase + ring_j) * 64 + (int)(meta_j & 63u);
#ifdef MNV_F2_CHECK_RINGS
                    if (share == 0 && (meta_j & kF2Ready) && F.diag) atomicAdd(F.diag + 29, 1ull << 48);  // evaluated twice
#endif
                    float *dst = reinterpret_cast<float *>(ring_data(ring_j) + e_j);
                    const float dz_j = dst[3];  // read by both lane shares before either writes its results ...
                    __builtin_amdgcn_wave_barrier();  // ... (the other share of this column writes dst[3]: the read must not sink into a branch)
                    const bool last = (meta_j & 64u) != 0;

                    if constexpr (BASIS >= 0) {
                        // One instruction stream for both lane shares (a branch on the share would run both bodies, each on half the
                        // lanes): every lane evaluates two channel sums at ITS offsets and two exponentials of ITS arguments.
                        //   share 0: X = sigma * dz (feature 3), Y = channel 0   -> {last ? 0 : exp(-X), 1 + exp(-Y)}
                        //   share 1: X = channel 1,              Y = channel 2   -> {1 + exp(-X),        1 + exp(-Y)}
                        float basis[NB];
#pragma unroll
                        for (int k = 0; k < NB; ++k) basis[k] = s_ray[k * RB + owner_j];
                        constexpr int stride = BASIS > 0 ? BASIS : 0;
                        const float *col_out = s_out + c32;
                        const int offA = share ? stride : 0, offB = share ? 2 * stride : 0;
                        auto svA = [&](int f) -> float { return kValid ? col_out[(offA + f) * 32] : 0.f; };
                        auto svB = [&](int f) -> float { return kValid ? col_out[(offB + f) * 32] : 0.f; };
                        const float dA = sh_channel<BASIS>(basis, svA, 0), dB = sh_channel<BASIS>(basis, svB, 0);
                        const float sig = kValid ? col_out[3 * 32] : 0.f;
                        const float X = share ? dA : sig * dz_j, Y = share ? dB : dA;
#ifdef MNV_F2_LOG
                        if (F.diag && F.diag[30]) {
                            const uint32_t key = ring_meta(ring_j)[2 * kF2Ring + e_j];  // pixel * 64 + sample
                            float *rec = reinterpret_cast<float *>(F.diag[30]) + ((size_t)key * 2 + share) * 24;
#pragma unroll
                            for (int f = 0; f < 9; ++f) {
                                rec[f] = svA(f);
                                rec[9 + f] = svB(f);
                            }
                            rec[18] = dA; rec[19] = dB; rec[20] = X; rec[21] = __uint_as_float((uint32_t)jc | ((uint32_t)n << 8) | ((uint32_t)half << 16) | ((uint32_t)(n_runs & 0xfff) << 20));
                            rec[22] = dz_j; rec[23] = sig;
                        }
#endif
                        const float eX = exact_expf_select(-X, s_exp), eY = exact_expf_select(-Y, s_exp);
                        const float out0 = share ? 1.f + eX : (last ? 0.f : eX), out1 = 1.f + eY;
                        *reinterpret_cast<float2 *>(dst + 2 * share) = make_float2(out0, out1);
                    } else {
                        auto sv = [&](int f) -> float { return kValid ? s_out[f * 32 + c32] : 0.f; };  // no sub-module: zeros (mlp_histogram)
                        if (share == 0) {
                            const float att = exact_expf_select(-sv(3) * dz_j, s_exp);
                            dst[0] = last ? 0.f : att;
                            dst[1] = sv(0);
                        } else {
                            dst[2] = sv(1);
                            dst[3] = sv(2);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the next half (or the next window's encode) rewrites the tile
            };

            if (valid_cluster) {
                const unsigned long long t_e0 = F.diag ? wall_clock64() : 0;
                // ---- the weights of the sub-module: kF2NS slots in LDS shared by the workgroup's consumers (any number may read a slot at
                //      a time); a sub-module that is in no slot replaces the least recently used slot nobody reads (16 KB from L2)
                int slot = held_slot;
                if (held_slot < 0 || c_star != lds_cluster) {
                    if (held_slot >= 0 && lane == 0) __hip_atomic_fetch_sub(s_wc + 2 + 4 * held_slot + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    uint32_t res = 0u;  // slot | 8: fill it, | 16: somebody else is filling it, | 32: gave up (watchdog)
                    if (lane == 0) {
                        uint32_t lock_spins = 0u;
                        for (;;) {
                            bool locked = false;
                            for (;;) {  // the lock: acquire, so that nothing below is read before the lock is held
                                uint32_t expect = 0u;
                                if (__hip_atomic_compare_exchange_strong(&s_wc[0], &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                    locked = true;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(0);
                                if (++lock_spins > 64u * kF2SpinLimit) break;  // watchdog, as for every other wait of this kernel
                            }
                            if (!locked) {
                                res = 32u;
                                break;
                            }
                            int hit = -1, victim = -1;
                            uint32_t oldest = 0xffffffffu;
#pragma unroll
                            for (int k = 0; k < kF2NS; ++k) {
                                if (k >= F.weight_slots) break;
                                const uint32_t *w4 = s_wc + 2 + 4 * k;
                                const uint32_t wcl = __hip_atomic_load(w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), rd = __hip_atomic_load(w4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                                               stp = __hip_atomic_load(w4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if ((int)wcl == c_star) hit = k;
                                else if (rd == 0u && stp < oldest) {
                                    oldest = stp;
                                    victim = k;
                                }
                            }
                            const uint32_t now = __hip_atomic_load(&s_wc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
                            if (hit >= 0 || victim >= 0) {
                                const int k = hit >= 0 ? hit : victim;
                                uint32_t *w4 = s_wc + 2 + 4 * k;
                                __hip_atomic_store(&s_wc[1], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(w4 + 3, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(w4 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (hit >= 0) {
                                    res = (uint32_t)k | (__hip_atomic_load(w4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 16u : 0u);
                                } else {
                                    __hip_atomic_store(w4, (uint32_t)c_star, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_store(w4 + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    res = (uint32_t)k | 8u;
                                }
                            }
                            __hip_atomic_store(&s_wc[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (hit >= 0 || victim >= 0) break;
                            __builtin_amdgcn_s_sleep(1);  // every slot is being read and none holds this sub-module: a reader will let go
                            if (++lock_spins > 64u * kF2SpinLimit) {
                                res = 32u;
                                break;
                            }
                        }
                    }
                    res = (uint32_t)__builtin_amdgcn_readfirstlane((int)res);
                    if (res & 32u) {  // no slot within the watchdog's patience: leave (counted below), the frame is reported as faulty
                        spins = 64u * kF2SpinLimit + 1u;
                        held_slot = -1;
                        break;
                    }
                    slot = (int)(res & 7u);
                    uint32_t *w4 = s_wc + 2 + 4 * slot;
                    if (res & 8u) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(F.frags + (size_t)c_star * S.frag_halfs);
                        uint4 *dst = reinterpret_cast<uint4 *>(s_mem + Lo.frags + slot * Lo.frag_words);
                        const float *bsrc = F.biases + (size_t)c_star * S.bias_floats;
                        float *bdst = reinterpret_cast<float *>(s_mem + Lo.bias + slot * Lo.bias_words);
                        const int nfrag = S.frag_halfs / 512;  // whole fragments: 64 lanes x 16 bytes each
                        float bv[4];  // the biases travel with the first batch of fragments (bias_floats <= 256 here: launch_accel)
#pragma unroll
                        for (int u = 0; u < 4; ++u) bv[u] = lane + 64 * u < S.bias_floats ? bsrc[lane + 64 * u] : 0.f;
                        for (int u0 = 0; u0 < nfrag; u0 += 16) {  // every guard below is wave-uniform: the batch stays in registers, its loads in flight together
                            uint4 v[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) v[u] = src[(u0 + u < nfrag ? u0 + u : nfrag - 1) * 64 + lane];  // unconditional (a clamped index): registers, not a stack array
#pragma unroll
                            for (int u = 0; u < 16; ++u)
                                if (u0 + u < nfrag) dst[(u0 + u) * 64 + lane] = v[u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (lane + 64 * u < S.bias_floats) bdst[lane + 64 * u] = bv[u];
                        st_release(w4 + 1, 0u);  // filled
                        ++n_reloads;
                        if (F.diag) t_reload += wall_clock64() - t_e0;
                    } else if (res & 16u) {
                        uint32_t fill_spins = 0u;
                        while (__builtin_amdgcn_readfirstlane((int)ld_relaxed(w4 + 1)) != 0 && ++fill_spins <= 64u * kF2SpinLimit) __builtin_amdgcn_s_sleep(0);
                        if (fill_spins > 64u * kF2SpinLimit) {
                            spins = 64u * kF2SpinLimit + 1u;
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    lds_cluster = c_star;
                    held_slot = slot;
                }
                const half8 *s_frag = reinterpret_cast<const half8 *>(s_mem + Lo.frags + slot * Lo.frag_words);
                const float *s_bias = reinterpret_cast<const float *>(s_mem + Lo.bias + slot * Lo.bias_words);
#ifdef MNV_F2_CHECK_WEIGHTS
                auto check_weights = [&](int word) __attribute__((noinline)) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(F.frags + (size_t)c_star * S.frag_halfs);
                    const uint4 *dst = reinterpret_cast<const uint4 *>(s_mem + Lo.frags + slot * Lo.frag_words);
                    bool bad = false;
                    for (int u = 0; u < S.frag_halfs / 512; ++u) {
                        const uint4 a = src[u * 64 + lane], b = dst[u * 64 + lane];
                        bad |= a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
                    }
                    const float *bsrc = F.biases + (size_t)c_star * S.bias_floats;
                    const float *bdst = reinterpret_cast<const float *>(s_mem + Lo.bias + slot * Lo.bias_words);
                    for (int u = lane; u < S.bias_floats; u += 64) bad |= __float_as_uint(bsrc[u]) != __float_as_uint(bdst[u]);
                    if (__ballot(bad) != 0 && lane == 0 && F.diag) atomicAdd(F.diag + word, 1ull);
                    if (lane == 0 && F.diag && (int)ld_relaxed(s_wc + 2 + 4 * slot) != c_star) atomicAdd(F.diag + 28, 1ull);
                };
                check_weights(28);
#endif
                // ---- encode: lane j builds column j of the B operand.  A column of one K tile is 64 bytes in operand order (the 8 halfs
                //      of lane group g at byte 16 g: features 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3), written as four 16-byte
                //      stores.  Position features have compile-time places (p, then per octave three phase-0 and three phase-1/4
                //      triangle waves), guarded per octave; view direction, embedding and padding follow with half-word stores.
                const uint32_t where = col_on ? s_cols[lane] : 0u;
                const float4 smp = col_on ? ring_data((int)(where >> kF2RingLog))[where & (kF2Ring - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
                {
                    float p[3];
                    p[0] = (smp.x - S.center[0]) * S.inv_extent[0];
                    p[1] = (smp.y - S.center[1]) * S.inv_extent[1];
                    p[2] = (smp.z - S.center[2]) * S.inv_extent[2];
                    // position feature f of this column (0 beyond the network's octaves): f < 3 the coordinate itself, then per octave k
                    // three phase-0 and three phase-1/4 triangle waves; f is a compile-time constant, the octave test is wave-uniform
                    auto feat = [&](auto f_tag) __attribute__((always_inline)) -> float {
                        constexpr int f = decltype(f_tag)::value;
                        if constexpr (f < 3) {
                            return p[f];
                        } else {
                            constexpr int k = (f - 3) / 6, r = (f - 3) % 6, i = r % 3;
                            const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
                            return k < S.pos_octaves ? tri_wave(p[i] * scale + (r >= 3 ? 0.25f : 0.f)) : 0.f;
                        }
                    };
                    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                    typedef float float2v __attribute__((ext_vector_type(2)));
                    uint4 *tile_q = reinterpret_cast<uint4 *>(s_tile);
                    auto pair = [&](auto f_tag) __attribute__((always_inline)) -> uint32_t {
                        constexpr int f = decltype(f_tag)::value;
                        const float2v pr = {feat(std::integral_constant<int, f>{}), feat(std::integral_constant<int, f + 1>{})};
                        return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, half2v));
                    };
                    auto group = [&](auto kk_tag, auto gg_tag) __attribute__((always_inline)) {
                        // elements 2 q, 2 q + 1 of lane group gg: features 16 (q >> 1) + 4 gg + 2 (q & 1) + {0, 1}
                        constexpr int kk = decltype(kk_tag)::value, gg = decltype(gg_tag)::value, f0 = 32 * kk + 4 * gg;
                        if (lane < kF2Cols) tile_q[kk * (4 * kF2Cols) + lane * 4 + gg] = make_uint4(pair(std::integral_constant<int, f0>{}), pair(std::integral_constant<int, f0 + 2>{}),
                                                                      pair(std::integral_constant<int, f0 + 16>{}), pair(std::integral_constant<int, f0 + 18>{}));
                    };
                    auto k_tile = [&](auto kk_tag) __attribute__((always_inline)) {
                        group(kk_tag, std::integral_constant<int, 0>{});
                        group(kk_tag, std::integral_constant<int, 1>{});
                        group(kk_tag, std::integral_constant<int, 2>{});
                        group(kk_tag, std::integral_constant<int, 3>{});
                    };
                    k_tile(std::integral_constant<int, 0>{});
                    if constexpr (NKK0 == 2) k_tile(std::integral_constant<int, 1>{});
                    if (S.need_viewdir || S.n_embeddings > 0) {
                        float d[3];
                        const int owner_thread = (rbase + (int)(where >> kF2RingLog)) * 64 + (int)(ring_meta((int)(where >> kF2RingLog))[where & (kF2Ring - 1)] & 63u);
#pragma unroll
                        for (int i = 0; i < 3; ++i) d[i] = S.need_viewdir ? s_ray[(NB + i) * RB + owner_thread] : 0.f;
                        _Float16 *tile_h = reinterpret_cast<_Float16 *>(s_tile);
                        auto put = [&](int f, float v) {  // f is wave-uniform
                            const int r = f & 31;
                            if (lane < kF2Cols) tile_h[(f >> 5) * (32 * kF2Cols) + lane * 32 + ((r & 15) >> 2) * 8 + (r >> 4) * 4 + (r & 3)] = (_Float16)v;
                        };
                        if (S.need_viewdir) {
                            const int base = S.dir_base;  // (every block starts at a multiple of 16 slots: mnv_mlp.h)
#pragma unroll
                            for (int i = 0; i < 3; ++i) put(base + i, d[i]);
                            for (int k = 0; k < S.dir_octaves; ++k) {
                                const float scale = __uint_as_float((uint32_t)(127 + k) << 23);
#pragma unroll
                                for (int i = 0; i < 3; ++i) {
                                    put(base + 3 + 6 * k + i, tri_wave(d[i] * scale + 0.f));
                                    put(base + 3 + 6 * k + 3 + i, tri_wave(d[i] * scale + 0.25f));
                                }
                            }
                        }
                        if (S.n_embeddings > 0) {
                            int idx = (int)(float)F.appearance_embedding;
                            idx = idx < 0 ? 0 : (idx >= S.n_embeddings ? S.n_embeddings - 1 : idx);
                            const uint16_t *emb = F.embeddings + ((size_t)c_star * S.n_embeddings + idx) * S.embedding_dim;
                            for (int j = 0; j < S.embedding_dim; ++j) put(S.emb_base + j, half_bits_to_float(emb[j]));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- the layers: weights = A operand (from LDS), 16 samples of a column tile = B; a layer's C layout is the next layer's B
                //      layout.  Every layer runs over the columns in two halves (column tiles 0-1, then 2-3): the B operands of one half
                //      are 8 to 16 registers instead of 32 and the previous layer's accumulators die half by half, so that the 64
                //      accumulators, the operands and the fragments in flight stay inside the 128-register budget without spills
                //      (the price: a layer's fragments are read from LDS twice).
                f32x4 acc[MT][kF2NT];
                const half8 *w = s_frag;
                const float *b = s_bias;
                auto bias_tile = [&](int mt) __attribute__((always_inline)) -> f32x4 { return *reinterpret_cast<const f32x4 *>(b + 16 * mt + 4 * g); };
                {
                    const uint4 *tile_q = reinterpret_cast<const uint4 *>(s_tile);
#pragma unroll
                    for (int h = 0; h < kF2Halves; ++h) {
                        half8 bf[NKK0][2];
#pragma unroll
                        for (int kk = 0; kk < NKK0; ++kk)
#pragma unroll
                            for (int j = 0; j < 2; ++j) bf[kk][j] = __builtin_bit_cast(half8, tile_q[kk * (4 * kF2Cols) + ((2 * h + j) * 16 + col) * 4 + g]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
#if MNV_F2_L0_BLOCKS
                            // A scheduling hint, nothing else: without it the compiler hoists the 12 operand loads of all four 16-row blocks above
                            // the first MFMA and merges the two column halves -- 64 accumulators AND 48 operand registers live at once, and the
                            // addresses of the evaluation go to scratch (38 instead of 15 scratch reloads in the kernel; 1.30 instead of 1.25 ms).
                            // (Round 3 knew this barrier as the thing that hid the wrong denominator; that was the packed add, header comment.)
                            asm volatile("" ::: "memory");
#endif
                            const f32x4 bv = bias_tile(mt);
#pragma unroll
                            for (int kk = 0; kk < NKK0; ++kk) {
                                const half8 a = w[(mt * NKK0 + kk) * 64 + lane];
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[mt][2 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[kk][j], kk == 0 ? bv : acc[mt][2 * h + j], 0, 0, 0);
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the outputs
                if (F.diag) t_enc += wall_clock64() - t_e0;
                w += MT * NKK0 * 64;
                b += 16 * MT;
                for (int layer = 1; layer <= S.hidden_layers; ++layer) {
                    const int n_mt = layer < S.hidden_layers ? MT : S.mt_out;
#pragma unroll
                    for (int h = 0; h < kF2Halves; ++h) {
                        half8 bf[MT / 2][2];
#pragma unroll
                        for (int kk = 0; kk < MT / 2; ++kk)
#pragma unroll
                            for (int j = 0; j < 2; ++j) bf[kk][j] = relu_pack(acc[2 * kk][2 * h + j], acc[2 * kk + 1][2 * h + j]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (mt < n_mt) {
                                const half8 a0 = w[(mt * 2 + 0) * 64 + lane], a1 = w[(mt * 2 + 1) * 64 + lane];
                                const f32x4 bv = bias_tile(mt);
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[mt][2 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf[0][j], bv, 0, 0, 0);
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[mt][2 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf[1][j], acc[mt][2 * h + j], 0, 0, 0);
                            }
                        }
                    }
                    w += n_mt * (MT / 2) * 64;
                    b += 16 * n_mt;
                }
                // done reading the slot.  With a slot per consumer the reference is kept until this consumer changes sub-module or idles (one
                // that asks for a slot holds none, so a slot nobody reads always exists); with fewer slots it is dropped after every run.
#ifdef MNV_F2_CHECK_WEIGHTS
                check_weights(28);
#endif
                if (F.weight_slots < kF2NC) {
                    if (lane == 0) __hip_atomic_fetch_sub(s_wc + 2 + 4 * slot + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    held_slot = -1;
                }
                if (F.diag) t_lay += wall_clock64() - t_e0;
                const unsigned long long t_c0 = F.diag ? wall_clock64() : 0;
#pragma unroll
                for (int half = 0; half < kF2Halves; ++half) {
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
#ifdef MNV_F2_CHECK_TILE
                    auto readback = [&](int word) __attribute__((always_inline)) {
                        bool bad = false;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (mt < S.mt_out) {
#pragma unroll
                                for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float back = *reinterpret_cast<volatile float *>(&s_out[(16 * mt + 4 * g + r) * 32 + nn * 16 + col]);
                                        bad |= __float_as_uint(back) != __float_as_uint(acc[mt][2 * half + nn][r]);
                                    }
                            }
                        }
                        if (__ballot(bad) != 0 && lane == 0 && F.diag) atomicAdd(F.diag + word, 1ull);
                    };
                    readback(29);
                    __builtin_amdgcn_wave_barrier();
#endif
                    evaluate(half, std::true_type{});
#ifdef MNV_F2_CHECK_TILE
                    readback(30);
                    __builtin_amdgcn_wave_barrier();
#endif
                }
                if (F.diag) t_eval += wall_clock64() - t_c0;
            } else {
                evaluate(0, std::false_type{});
                if constexpr (kF2Halves == 2) evaluate(1, std::false_type{});
            }
            // ---- publish: mark the entries, then the counters (the owners composite them; their slots come free once they have)
            if (col_on) {
                const uint32_t where = s_cols[lane];
                __hip_atomic_fetch_or(ring_meta((int)(where >> kF2RingLog)) + (where & (kF2Ring - 1)), kF2Ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // after the results
            }
#pragma unroll
            for (int p = 0; p < kF2RPC; ++p) {
                if (taken[p] != 0u) {
                    evald[p] += taken[p];
                    if constexpr (kF2SH == 1) {
                        st_release(s_rctrl + 8 * p + 2, evald[p]);
                    } else {  // several consumers evaluate entries of this ring: the count of evaluated entries is a sum
                        if (lane == 0) __hip_atomic_fetch_add(s_rctrl + 8 * p + 2, taken[p], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (F.diag) t_busy += wall_clock64() - t_w0;
        }
        if (spins > 64u * kF2SpinLimit && lane == 0) atomicAdd(F.fault, 1u);
        if (F.diag && lane == 0) {
            if (spins > 64u * kF2SpinLimit) atomicAdd(F.diag + F2Diag::kWatchdog, 1ull);
            atomicAdd(F.diag + F2Diag::kConsSimd + (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3), 1ull);
            atomicAdd(F.diag + F2Diag::kRuns, n_runs);
            atomicAdd(F.diag + F2Diag::kWindows, n_runs);
            atomicAdd(F.diag + F2Diag::kReloads, n_reloads);
            atomicAdd(F.diag + F2Diag::kGlobalRuns, t_reload);  // (the slot of the retired L2 path: ticks spent refilling the weights)
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
