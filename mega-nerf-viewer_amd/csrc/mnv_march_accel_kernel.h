// mnv_march_accel_kernel.h -- the tuned N3Tree march for gfx950 (MI355X) on the packed layout.
//
// Replaces, for the RGBA output of one frame or tile, the reference's
//   render_voxels_kernel          src/cuda/renderer_kernel.cu:243-292
//   render_voxels_trace_ray       include/cuda/rt_core.cuh:162-332
//   query_single_from_root        include/cuda/rt_core.cuh:117-159
// with bit-identical pixels (tests/test_parity_gpu.py).  Design, in CDNA4 terms:
//   * persistent workgroups (a few per CU); each 64-lane wavefront walks a packet of rays,
//     one ray per lane, and refills finished lanes from per-XCD ray queues (__ballot +
//     mbcnt rank + one wave-level atomic), so that early-terminated rays do not leave
//     lanes idle while the slowest ray of an 8x8 tile finishes;
//   * the top `grid_level` levels of the octree are a dense grid staged in LDS (up to
//     128 KiB of the CU's 160 KiB): a step in coarse empty space costs one ds_read and no
//     global load, and deep descents start at level grid_level + 1;
//   * below it one load from a dense brick-ordered level-L2 grid (L2 <= 9), then -- tracker / sample frames -- one 32-bit node
//     word per level: the child link or the leaf's depth and sigma, so the dependent sigma load of the reference disappears;
//   * plain frames (BRICK = true) do not read node words at all when the tree allows it: the grid's cell word carries the last
//     level inline (which of the chunk's eight leaves are non-empty), two more levels come from one 8-byte entry of a 64-byte
//     brick record, and a non-empty leaf's sigma is read WITH its colour row (the reference's row: coefficients, then sigma);
//   * colour rows are fetched (16-B vector loads from a 64-B padded row) only for dense samples;
//   * rays are queued in 8x8-pixel tile order and the tile range is split into 8
//     contiguous bands, one per XCD (workgroup b runs on XCD b % 8), so that each XCD's
//     4 MiB L2 holds the sub-trees of its own screen region; empty queues steal.
// MFMA is not used: the inner step is pointer chasing plus a <= 75-term dot in a fixed
// summation order (DESIGN.md "Why no MFMA").
#pragma once

#include "mnv_accel_launch.h"

#pragma clang fp contract(off)

namespace mnv {

// Attribution of the L2 misses by array (tools/traffic_by_array.sh builds variants of the library with -DMNV_SHADOW_MASK=<bits>): every load
// of the chosen array is repeated at the same index of a copy at other addresses -- results stay right, the miss counters grow by about
// that array's share.  8 grid2 (its shadow is grid2_vox), 16 nodes, 32 rows, 64 brick records.  0 in every shipped build.
#ifndef MNV_SHADOW_MASK
#define MNV_SHADOW_MASK 0
#endif
constexpr int kShadow = MNV_SHADOW_MASK;


// One step of the march on integer cell coordinates.  pos in [0, 1-1e-6] is scaled by 2^Lq
// (Lq = deepest voxel depth of the tree, <= 23: the product is exact and < 2^24) and truncated;
// bit (Lq - d) of each coordinate is the child index at depth d, and the cell numbers of the two
// lookup grids are plain shifts.  The in-leaf coordinates are fract(pos * 2^depth), which equals the
// reference's iterated x*2 - floor(x*2) bit for bit (all three operations are exact in binary32).
template <int BASIS, int BLOCK, int MODE /* 0 plain, 1 statistics, 2 refinement trackers, 3 trackers + emitted samples instead of colour, 4 plain with fast colour math, 5 depth image (render_depth) */,
          bool BRICK = false /* the levels below the second lookup grid come from inline cell words (A.grid2i) and brick records (A.recs) instead of node loads */>
// A/B knobs (tools/build_variant.sh): explicit register budgets on top of the launch bounds
#if defined(MNV_NUM_VGPR) && defined(MNV_NUM_SGPR)
#define MNV_EXTRA_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(MNV_NUM_VGPR), amdgpu_num_sgpr(MNV_NUM_SGPR)))
#elif defined(MNV_NUM_VGPR)
#define MNV_EXTRA_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(MNV_NUM_VGPR)))
#else
#define MNV_EXTRA_KERNEL_ATTR
#endif
__global__ __launch_bounds__(BLOCK, (MODE == 2 || MODE == 3) ? MNV_TRACK_WAVES : MNV_MIN_WAVES) MNV_EXTRA_KERNEL_ATTR void march_accel_kernel(const AccelLaunch K) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint64_t *s_exp = reinterpret_cast<uint64_t *>(s_mem);  // 32 x 8 B
    constexpr int NB = BASIS > 0 ? BASIS : 1;
    // Uniform switches cost scalar registers in the hot loop (the kernel runs at the 80-SGPR limit of 8 workgroups per CU, and
    // what does not fit is parked in VGPR lanes and read back with VALU instructions): the colour kernels (MODE 0 / 4) carry
    // neither the depth-image switch nor the diagnostics word.
    auto depth_mode = [&]() -> bool {
        if constexpr (MODE == 0 || MODE == 4) return false;
        else if constexpr (MODE == 5) return true;
        else return K.P.render_depth != 0;
    };
    auto ablate = [&](int bit) -> bool {
        if constexpr (MODE == 1) return (K.ablate & bit) != 0;
        else return false;
    };
    // per-lane ray constants that only the dense-sample / finish code needs live in LDS, not in VGPRs:
    // [k][thread] for k < NB: SH basis; then delta_scale and the output pixel index
    constexpr int MAPW = BASIS >= 16 ? BLOCK : 0;        // cooperative colour pass (SH16 / SH25 only): dense-sample rank -> lane, per wavefront
    uint32_t *s_map = s_mem + 64;
    float *s_ray = reinterpret_cast<float *>(s_mem + 64 + MAPW);
    constexpr int RAY_ROWS = NB + 2 + ((MODE == 2 || MODE == 3) ? 1 : 0);  // tracker / sample frames also keep a ray's t_min (the re-walk below)
    uint32_t *s_grid = s_mem + 64 + MAPW + RAY_ROWS * BLOCK;  // (2^lds_level)^3 words
    constexpr int CHAN_BYTES = chan_bytes_for(BASIS);
    constexpr int ROW_BYTES = row_bytes_pow2(BASIS);
    const FrameParams &P = K.P;
    const AccelView &A = K.A;

    const int LL = K.lds_level;
    const int cells = 1 << (3 * LL);
    if (threadIdx.x < 32) s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
    if (LL == A.grid_level) {
        for (int i = threadIdx.x; i < cells; i += BLOCK) s_grid[i] = A.grid[i];
    } else {
        // coarser LDS grid: walk the top LL levels as the builder does
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
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int Lq = A.max_depth;
    const float qscale = __uint_as_float((uint32_t)(127 + Lq) << 23);  // 2^Lq
    const int sh1 = Lq - LL;                                            // q >> sh1 = LDS cell
    const int L2 = A.grid2_level;                                       // 0: no second grid
    const int sh2 = Lq - L2;
    const int shg = Lq - A.grid_level;
    float *my_ray = s_ray + threadIdx.x;             // [k * BLOCK]
    float *wave_ray = s_ray + (threadIdx.x & ~63);  // [k * BLOCK + lane]

    const CamBlock *__restrict__ Cp = K.cams;  // camera of the frame this wavefront's rays belong to (wave-uniform pointer: scalar loads)
    float cen0 = Cp->cen[0], cen1 = Cp->cen[1], cen2 = Cp->cen[2];
    // per-lane ray state
    float t = 0.f, T = 1.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    float dir0 = 0.f, dir1 = 0.f, dir2 = 0.f, inv0 = 0.f, inv1 = 0.f, inv2 = 0.f, tmax = 0.f;
    bool alive = false;
    // MODE 2: per-ray tracker state
    float max_weight = -1.f, max_sample_weight = -1.f, sp_prio = 0.f, sa_prio = 0.f;
    int32_t sp_vox = -1, sa_vox = -1;
    int32_t ns = 0;  // MODE 3: samples emitted by this ray so far
    // The trackers' fallbacks -- the LAST leaf without a dense sample that may be split (depth < max_depth), resp. whose sample count is below
    // the limit, each kept as long as no dense leaf has qualified (rt_core.cuh:308-321 overwrites them at every such step) -- are resolved when
    // the pixel is written: the march remembers the t of the last such step (sp_t, sa_t) instead of naming the voxel of every leaf a ray steps
    // through (a second load from the 512-MiB voxel grid per step above the second grid) and gathering sample_counts[] for it.  One look-up per
    // ray then names the voxel (the same arithmetic at the same t finds the same leaf); if THAT leaf's count is saturated, the ray's steps are
    // walked again for the last one that is not.  0.10 of 0.57 ms of a 1080p tracker frame.
    [[maybe_unused]] float sp_t = -1.f, sa_t = -1.f;
    // the leaf at parameter tw of this lane's ray: its word (depth, sigma), its voxel, and the step the march takes from there
    [[maybe_unused]] auto leaf_at = [&](float tw, uint32_t &word, uint32_t &v, float &dt) {
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
        if (src == 0) v = A.grid_vox[((((q[0] >> shg) << A.grid_level) + (q[1] >> shg)) << A.grid_level) + (q[2] >> shg)];
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
    [[maybe_unused]] auto resolve_fallbacks = [&]() {
        if constexpr (MODE == 2 || MODE == 3) {
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
                } else {
                    // the ray's own step sequence once more (same arithmetic: the same t values), looking at the leaves without a dense sample only
                    float tw = my_ray[(NB + 2) * BLOCK];
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
    };
    auto write_trackers = [&](uint32_t p) {
        resolve_fallbacks();
        if constexpr (MODE == 3) K.num_samples[p] = (int16_t)ns;
        if constexpr (MODE == 2 || MODE == 3) {
            if (K.split_track) {
                K.split_track[(int64_t)p * 3 + 0] = sp_prio;
                K.split_track[(int64_t)p * 3 + 1] = sp_vox < 0 ? -1.f : (float)(sp_vox >> 3);
                K.split_track[(int64_t)p * 3 + 2] = sp_vox < 0 ? -1.f : (float)(sp_vox & 7);
            }
            if (K.sample_track) {
                K.sample_track[(int64_t)p * 3 + 0] = sa_prio;
                K.sample_track[(int64_t)p * 3 + 1] = sa_vox < 0 ? -1.f : (float)(sa_vox >> 3);
                K.sample_track[(int64_t)p * 3 + 2] = sa_vox < 0 ? -1.f : (float)(sa_vox & 7);
            }
        }
    };

    // A finished ray only records HOW it ended (fin: 1 loop exit, 2 early stop, 3 missed the bounding box); its lane idles until the
    // wavefront refills anyway, so the end-of-ray arithmetic and the pixel / tracker stores run once per tile for all its lanes
    // instead of once per iteration in which some ray ends (one or two lanes at a time: ~6 % of the kernel's VALU issue).
    int fin = 0;
    auto flush_finished = [&]() {
        if (fin != 0) {
            float a;
            if (fin == 3) {
                o0 = o1 = o2 = 0.f;
                a = depth_mode() ? 1.f : 0.f;
            } else if (fin == 1) {  // rt_core.cuh:325-330
                a = 1.f - T;
                if (depth_mode()) {
                    o0 = o1 = o2 = fminf(o0 * 0.3f, 1.0f);
                    a = 1.f;
                }
            } else {  // rt_core.cuh:295-307
                if (depth_mode()) o0 = o1 = o2 = fminf(o0 * 0.3f, 1.0f);
                const float sc = 1.f / (1.f - T);
                o0 *= sc;
                o1 *= sc;
                o2 *= sc;
                a = 1.f;
            }
            const uint32_t p = __float_as_uint(my_ray[(NB + 1) * BLOCK]);
            if constexpr (MODE != 3) composite_and_write(P, (int64_t)p, o0, o1, o2, a);
            write_trackers(p);
            fin = 0;
        }
    };

    // ray queues: queue q holds band q (a contiguous run of 8x8 tiles) of EVERY frame of the batch, frame-major, behind one
    // head; a wavefront drains its home queue (workgroup b -> XCD b % 8), then steals round robin.  A batch is refilled a
    // whole tile at a time (refill_min = 64), so a grab never straddles two frames and the camera stays wave-uniform; the
    // tail of one frame overlaps the head of the next, and an exhausted queue is polled once per wavefront, not once per frame.
    const uint32_t home = blockIdx.x % kNumQueues;
    uint32_t qsel = 0;     // queues tried so far (wave-uniform)
    bool drained = false;  // every queue is empty (wave-uniform)
    uint32_t frame = 0;    // frame of the rays this wavefront holds (wave-uniform)
    uint32_t pix_base = 0;

    // MODE 1 + MNV_TIMELINE: when each tile was grabbed and finished and by which wavefront (tools/timeline.py)
    uint32_t tl_rec = ~0u, tl_iters = 0;
    auto tl_close = [&](unsigned long long now) {
        if constexpr (MODE == 1) {
            if (K.timeline && tl_rec != ~0u && lane == 0) {
                K.timeline[(size_t)tl_rec * 4 + 1] = now;
                K.timeline[(size_t)tl_rec * 4 + 3] = tl_iters;
            }
        }
    };
    if constexpr (MODE == 1) {
        if (K.timeline && lane == 0) K.timeline[(size_t)K.timeline_tiles * 4 + (size_t)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 2] = wall_clock64();
    }

    // MODE 1 + MNV_FOOTPRINT: the 128-byte line a load touches, by array (region) and byte offset into it
    auto touch = [&](int region, uint64_t byte_off) {
        if constexpr (MODE == 1) {
            if (K.line_bits) {
                const uint32_t line = K.line_base[region] + (uint32_t)(byte_off >> 7);
                atomicOr(&K.line_bits[line >> 5], 1u << (line & 31u));
            }
        }
    };
    auto stat = [&](int slot, bool pred) {
        if constexpr (MODE == 1) {
            if (!K.count_stats) return;
            const uint64_t m = __ballot(pred);
            if (m && lane == (int)__builtin_ctzll(m)) {
                atomicAdd(&K.stats[slot], 1ull);
                atomicAdd(&K.stats[slot + 1], (unsigned long long)__popcll(m));
            }
        }
    };

    // MNV_STATS=2 (diagnostics instantiation only): where the cycles of a wavefront's step go.  Every stamp waits for all outstanding
    // memory operations first, so the phases do not overlap as they may in the product kernel: this is the DEPENDENT chain, what
    // bounds a launch that drains.  stats[16..]: lookup (LDS grid -> grid2 -> node words), step arithmetic + opacity, row wait,
    // colour arithmetic, the wavefront's whole time, wave-steps (tools/step_phases.py; LAB_NOTEBOOK.md round 3).
    unsigned long long ph_lookup = 0, ph_step = 0, ph_row = 0, ph_colour = 0, ph_steps = 0, ph_mark = 0;
    auto phase_clock = [&]() -> unsigned long long {
        if constexpr (MODE == 1) {
            if (K.count_stats == 2) {
                __builtin_amdgcn_s_waitcnt(0);
                return (unsigned long long)__builtin_readcyclecounter();
            }
        }
        return 0ull;
    };
    const unsigned long long ph_begin = phase_clock();

    for (;;) {
        const uint64_t idle = __ballot(!alive);
        const int n_idle = __popcll(idle);
        stat(0, true);
        if constexpr (MODE == 1) ++tl_iters;
        // Tile-sized refills keep a wavefront's rays coherent (sweep in DESIGN.md).
        if (!drained && n_idle >= K.refill_min) {
            flush_finished();  // the idle lanes' pixels, before they take new rays
            if (qsel >= kNumQueues) {
                drained = true;
            } else {
            // ---- refill idle lanes from the ray queues
            const uint32_t q = (home + qsel) % kNumQueues;
            const uint32_t begin = K.band_begin[q] * 64u, span = (K.band_begin[q + 1] - K.band_begin[q]) * 64u;  // rays of the band, per frame
            const uint32_t grab = (uint32_t)n_idle;
            uint32_t off = 0;
            if (lane == 0) off = atomicAdd(&K.queue[q * 16], grab);
            off = __builtin_amdgcn_readfirstlane(off);
            if (span == 0 || (uint64_t)off >= (uint64_t)span * K.n_frames) {
                ++qsel;
                continue;
            }
            uint32_t f = 0;
            if (K.n_frames > 1) {
                f = off / span;
                off -= f * span;
                if (f != frame) {
                    frame = f;
                    Cp = K.cams + f;
                    cen0 = Cp->cen[0];
                    cen1 = Cp->cen[1];
                    cen2 = Cp->cen[2];
                    pix_base = f * K.frame_stride_px;
                }
            }
            const uint32_t base = begin + off, end = begin + span;
            if constexpr (MODE == 1) {
                if (K.timeline) {
                    const unsigned long long now = wall_clock64();
                    tl_close(now);
                    tl_rec = f * K.n_tiles + (base >> 6);
                    tl_iters = 0;
                    if (lane == 0) {
                        K.timeline[(size_t)tl_rec * 4 + 0] = now;
                        K.timeline[(size_t)tl_rec * 4 + 2] = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
                    }
                }
            }
            if (!alive) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t id = base + rank;
                if (rank < grab && id < end) {
                    int bx, by;
                    stat(2, true);
                    uint32_t pix;
                    if (ray_pixel(K, id, bx, by, pix)) {
                        pix += pix_base;
                        if constexpr (MODE == 2 || MODE == 3) {
                            max_weight = max_sample_weight = -1.f;
                            sp_prio = (float)(K.max_depth + 1);
                            sa_prio = (float)(K.max_sample_count + 1);
                            sp_vox = sa_vox = -1;
                            sp_t = sa_t = -1.f;
                        }
                        if constexpr (MODE == 3) ns = K.num_samples[pix];
                        RaySetup<NB> r;
                        setup_ray<(BASIS > 0 ? BASIS : 0)>(P, *Cp, P.x0 + bx, P.y0 + by, r, frame_tmax(P, pix));
                        if constexpr (BASIS == 0)
                            r.basis[0] = (0 < P.basis_min || 0 > P.basis_max) ? 0.f : (float)0.28209479177387814;
                        o0 = o1 = o2 = 0.f;
                        if (r.in_bbox) {
                            alive = true;
                            t = r.tmin;
                            T = 1.f;
                            tmax = r.tmax;
                            dir0 = r.dir[0]; dir1 = r.dir[1]; dir2 = r.dir[2];
                            inv0 = r.invdir[0]; inv1 = r.invdir[1]; inv2 = r.invdir[2];
                            if constexpr (MODE == 3) {
                                float true_dir[3], vdir[3];
                                world_ray_dirs(P, *Cp, P.x0 + bx, P.y0 + by, true_dir, vdir);
#pragma unroll
                                for (int k = 0; k < 3; ++k) {
                                    my_ray[k * BLOCK] = true_dir[k];
                                    my_ray[(3 + k) * BLOCK] = vdir[k];
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < NB; ++k) my_ray[k * BLOCK] = r.basis[k];
                            }
                            my_ray[NB * BLOCK] = r.delta_scale;
                            my_ray[(NB + 1) * BLOCK] = __uint_as_float(pix);
                            if constexpr (MODE == 2 || MODE == 3) my_ray[(NB + 2) * BLOCK] = r.tmin;
                        } else {
                            my_ray[(NB + 1) * BLOCK] = __uint_as_float(pix);
                            fin = 3;  // the ray misses the bounding box: background pixel, written with the tile's others
                        }
                    }
                }
            }
            }
        }
        if (__ballot(alive) == 0) {
            if constexpr (MODE == 1) {
                if (K.timeline && tl_rec != ~0u) {  // the tile is done; what follows is queue polling
                    tl_close(wall_clock64());
                    tl_rec = ~0u;
                }
            }
            if (drained) {
                flush_finished();
                break;
            }
            continue;
        }
        // ---- one march step (rt_core.cuh:220-323) for every live lane
        bool dense = false;
        [[maybe_unused]] bool cand = false;  // BRICK: the leaf came out of a brick record with code 3 (sigma not known yet)
        [[maybe_unused]] float sigma_held = 0.f;  // BRICK: sigma of a dense leaf that did not come out of a record
        float delta_t = 0.f, weight = 0.f, att = 1.f;
        uint32_t vox = 0;
        [[maybe_unused]] bool track_leaf = false;  // MODE 2 / 3: this lane stepped into a leaf the trackers look at (vox is its voxel)
        [[maybe_unused]] int track_depth = 0;
        // rt_core.cuh:237-252 (dense leaf: best weight so far) and :308-321 (first leaf before any dense one); `dense` and `weight` final
        [[maybe_unused]] auto track_update = [&]() {
            if constexpr (MODE == 2 || MODE == 3) {
                if (track_leaf) {
                    if (dense) {  // (its voxel is known: need_vox)
                        if (track_depth < K.max_depth && weight > max_weight) {
                            sp_vox = (int32_t)vox;
                            sp_prio = (float)track_depth;
                            max_weight = weight;
                        }
                        if (K.sample_counts && weight > max_sample_weight) {
                            const int16_t sc = K.sample_counts[vox];
                            if (sc < K.max_sample_count) {
                                sa_vox = (int32_t)vox;
                                sa_prio = (float)sc;
                                max_sample_weight = weight;
                            }
                        }
                    } else {  // (t is still this step's t: the update runs before t += delta_t)
                        if (max_weight == -1.f && track_depth < K.max_depth) sp_t = t;
                        if (max_sample_weight == -1.f) sa_t = t;
                    }
                }
            }
        };
        ph_mark = phase_clock();
        if (alive) {
            if (!(t < tmax)) {
                // loop exit, rt_core.cuh:325-330: the pixel is finished in flush_finished()
                fin = 1;
                alive = false;
            } else {
                stat(4, true);
                float pos[3];
                uint32_t q[3];
                pos[0] = cen0 + t * dir0;
                pos[1] = cen1 + t * dir1;
                pos[2] = cen2 + t * dir2;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pos[i] = __builtin_amdgcn_fmed3f(pos[i], 0.f, 1.f - 1e-6f);  // == max(min(x, hi), 0) for every non-NaN x
                    q[i] = (uint32_t)(pos[i] * qscale);
                }
                // top of the tree: LDS grid at level LL
                uint32_t word = s_grid[((((q[0] >> sh1) << LL) | (q[1] >> sh1)) << LL) | (q[2] >> sh1)];
                int src = 0;  // where the leaf word came from: 0 LDS grid, 1 grid2, 2 node array
                if (!(word & kLeafBit)) {
                    int sh = sh1;  // q >> sh is the cell at the depth `word` describes
                    if (L2 > LL) {
                        // middle of the tree: one load from the brick-ordered level-L2 grid
                        // brick number from the high bits of the three cell coordinates, cell-in-brick from their two low bits
                        const int LB = L2 - 2;
                        uint32_t g = q[0] >> (sh2 + 2);
                        g = (g << LB) | (q[1] >> (sh2 + 2));
                        g = (g << LB) | (q[2] >> (sh2 + 2));
                        g = (g << 2) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh2, 2u);
                        g = (g << 2) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh2, 2u);
                        vox = (g << 2) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh2, 2u);
                        if constexpr (BRICK) word = A.grid2i[vox];
                        else word = A.grid2[vox];
                        touch(0, (uint64_t)vox * 4);
                        src = 1;
                        sh = sh2;
                        if constexpr ((kShadow & 8) != 0) {  // shadow load: the same cell of grid2_vox (same size, same order, other addresses)
                            const uint32_t w2 = A.grid2_vox[vox];
                            asm volatile("" ::"v"(w2));
                        }
                        if constexpr (BRICK) {
                            // the two levels below the grid cell from the 64-byte record of the chunk it names: ONE 8-byte load -- entry s1 = {child
                            // chunk of voxel s1, 2-bit codes of its eight sub-cells}; empty leaves of either level end here
                            if (!(word & kLeafBit)) {
                                uint32_t s1 = __builtin_amdgcn_ubfe(q[0], (uint32_t)(sh2 - 1), 1u);
                                s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)(sh2 - 1), 1u);
                                s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)(sh2 - 1), 1u);
                                if (word & kInlineBit) {
                                    // the cell's eight children are leaves and the word says which of them are empty: no load at all for an
                                    // empty one, and a non-empty one is a candidate whose sigma arrives with its colour row
                                    vox = (((word & ((1u << kInlineMaskShift) - 1u)) + A.inline_base) << 3) | s1;
                                    cand = ((word >> (kInlineMaskShift + s1)) & 1u) != 0u;
                                    word = kLeafBit | ((uint32_t)(L2 + 1) << 16);
                                    src = 2;
                                }
                            }
                            if (!(word & kLeafBit) && A.recs != nullptr) {
                                stat(6, true);
                                uint32_t s1 = __builtin_amdgcn_ubfe(q[0], (uint32_t)(sh2 - 1), 1u), s2 = __builtin_amdgcn_ubfe(q[0], (uint32_t)(sh2 - 2), 1u);
                                s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)(sh2 - 1), 1u);
                                s2 = (s2 << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)(sh2 - 2), 1u);
                                s1 = (s1 << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)(sh2 - 1), 1u);
                                s2 = (s2 << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)(sh2 - 2), 1u);
                                const uint2 e = A.recs[(int64_t)word * 8 + s1];
                                touch(1, ((uint64_t)word * 8 + s1) * 8);
                                if constexpr ((kShadow & 64) != 0) {
                                    const uint2 e2 = K.shadow_recs[(int64_t)word * 8 + s1];
                                    asm volatile("" ::"v"(e2.x));
                                }
                                const uint32_t code = __builtin_amdgcn_ubfe(e.y, s2 << 1, 2u);
                                // 0: walk on from the chunk (word stays); 1 / 2: an empty leaf of depth L2 + 1 / L2 + 2; 3: a leaf of depth L2 + 2
                                // with sigma != 0 -- its voxel through the entry's child word, its sigma with its colour row (cand)
                                // (the voxel of a code-1 leaf is voxel s1 of the cell's chunk; tracker frames name it)
                                vox = code == 1u ? ((word << 3) | s1) : ((e.x << 3) | s2);
                                cand = code == 3u;
                                word = code != 0u ? (kLeafBit | ((uint32_t)(L2 + 2 - (code == 1u ? 1 : 0)) << 16)) : word;
                                src = 2;
                            }
                        }
                    }
                    while (!(word & kLeafBit)) {
                        stat(6, true);
                        --sh;
                        uint32_t v = (word << 1) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh, 1u);
                        v = (v << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh, 1u);
                        vox = (v << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh, 1u);
                        word = A.nodes[vox];
                        touch(2, (uint64_t)vox * 4);
                        src = 2;
                        if constexpr ((kShadow & 16) != 0) {
                            const uint32_t w2 = K.shadow_nodes[vox];
                            asm volatile("" ::"v"(w2));
                        }
                    }
                }
                if constexpr (BRICK && MODE == 3) {
                    // the sample march reads no colour row: a non-empty leaf found through an inline word / a record fetches its node word
                    // (depth and sigma) -- the one load an empty leaf of the last levels no longer costs
                    if (cand) {
                        word = A.nodes[vox];
                        cand = false;
                    }
                }
                if constexpr (MODE == 1) {
                    if (K.count_stats == 2) {
                        const unsigned long long now = phase_clock();
                        ph_lookup += now - ph_mark;
                        ph_mark = now;
                    }
                }
                const int depth = (int)((word >> 16) & 0x7fu);
                const float sc = __uint_as_float((uint32_t)(127 + depth) << 23);        // 2^depth
                const float inv_cube = __uint_as_float((uint32_t)(127 - depth) << 23);  // 2^-depth
                // _dda_unit on the in-leaf coordinates, rt_core.cuh:88-100
                float tu = 1e4f;
                const float invd[3] = {inv0, inv1, inv2};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float x = __builtin_amdgcn_fractf(pos[i] * sc);
                    const float t1 = -x * invd[i];
                    const float t2 = t1 + invd[i];
                    tu = fminf(tu, fmaxf(t1, t2));
                }
                delta_t = tu * inv_cube + P.step_size;
                const float sigma = half_bits_to_float((uint16_t)word);
                const bool is_dense = sigma > P.sigma_thresh && !ablate(2);
                bool need_vox = is_dense;
                if constexpr (MODE == 2 || MODE == 3) need_vox = is_dense || cand || K.visited != nullptr;  // (leaves without a dense sample: by their t, see sp_t / sa_t)
                if (need_vox) {
                    // voxel index of a leaf that was answered by one of the lookup grids
                    if (src == 0) {
                        const uint32_t gc = ((((q[0] >> shg) << A.grid_level) + (q[1] >> shg)) << A.grid_level) + (q[2] >> shg);
                        vox = A.grid_vox[gc];
                        touch(5, (uint64_t)gc * 4);
                    } else if (src == 1) {
                        touch(4, (uint64_t)vox * 4);
                        vox = A.grid2_vox[vox];
                    }
                }
                if constexpr (MODE == 2 || MODE == 3) {
                    // the mark only ever goes 0 -> 1: load + conditional plain store (mnv_march_ref_layout.hip does the same per level)
                    if (K.visited && __hip_atomic_load(&K.visited[vox >> 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) K.visited[vox >> 3] = 1;
                }
                if constexpr (BRICK && MODE != 3) {
                    // the opacity of every dense sample is worked out in the colour block (settle): a candidate's sigma arrives there with its row
                    dense = (is_dense || cand) && !ablate(2);
                    sigma_held = sigma;
                } else if (is_dense) {
                    // opacity of a dense sample, rt_core.cuh:233-235
                    dense = true;
                    att = exact_expf(-delta_t * my_ray[NB * BLOCK] * sigma, s_exp);
                    weight = T * (1.f - att);
                }
                if constexpr (MODE == 2 || MODE == 3) {
                    track_leaf = true;
                    track_depth = depth;
                    // (BRICK tracker frames: whether the leaf is dense, and its weight, are known behind the colour block)
                    if constexpr (!(BRICK && MODE == 2)) track_update();
                }
                if constexpr (MODE == 3) {
                    // rt_core.cuh:508-549: one row per dense step while there is room
                    if (is_dense && ns < K.max_guided_samples) {
                        const uint32_t p = __float_as_uint(my_ray[(NB + 1) * BLOCK]);
                        float *row = K.samples + ((int64_t)p * K.max_guided_samples + ns) * K.samples_dim;
                        const float tz0 = t * dir0 / P.scale[0], tz1 = t * dir1 / P.scale[1], tz2 = t * dir2 / P.scale[2];
                        const float z = sqrtf(tz0 * tz0 + tz1 * tz1 + tz2 * tz2);
                        const float *m = Cp->c2w;
                        const float wx = m[9] + my_ray[0 * BLOCK] * z, wy = m[10] + my_ray[1 * BLOCK] * z, wz = m[11] + my_ray[2 * BLOCK] * z;
                        row[0] = z;
                        row[1] = wx;
                        row[2] = wy;
                        row[3] = wz;
                        if (K.need_viewdir) {
                            row[4] = my_ray[3 * BLOCK];
                            row[5] = my_ray[4 * BLOCK];
                            row[6] = my_ray[5 * BLOCK];
                            if (K.appearance_embedding != -1) row[7] = (float)K.appearance_embedding;
                        } else if (K.appearance_embedding != -1) {
                            row[4] = (float)K.appearance_embedding;
                        }
                        const int g1 = (int)fmaxf(fminf((wy - K.min_position[1]) / K.range[1] * (float)K.grid_dim[0], (float)K.grid_dim[0] - 1.0f), 0.0f);
                        const int g2 = (int)fmaxf(fminf((wz - K.min_position[2]) / K.range[2] * (float)K.grid_dim[1], (float)K.grid_dim[1] - 1.0f), 0.0f);
                        K.cluster_indices[(int64_t)p * K.max_guided_samples + ns] = (int16_t)(g1 * K.grid_dim[1] + g2);
                        ns += 1;
                    }
                }
            }
        }
        if constexpr (MODE == 1) {
            if (K.count_stats == 2) {
                const unsigned long long now = phase_clock();
                ph_step += now - ph_mark;
                ph_mark = now;
                ++ph_steps;
            }
        }
        // ---- colour of the dense samples of this iteration (rt_core.cuh:254-291)
        const uint64_t dense_mask = __ballot(dense);
        if (dense_mask != 0) {
            stat(8, dense);
            // BRICK: whether a sample is dense at all (rt_core.cuh:233) and its opacity (:234-235) are settled here, in one place -- a
            // candidate (leaf with sigma != 0 out of a brick record) learns its sigma from its row, the half behind the three channel blocks
            [[maybe_unused]] auto settle = [&](uint32_t row_sigma_bits) {
                const float sg = cand ? half_bits_to_float((uint16_t)row_sigma_bits) : sigma_held;
                if (sg > P.sigma_thresh) {
                    att = exact_expf(-delta_t * my_ray[NB * BLOCK] * sg, s_exp);
                    weight = T * (1.f - att);
                } else {
                    dense = false;
                }
            };
            [[maybe_unused]] auto settle_from_memory = [&]() {  // frames that read no colour row: the sigma half alone
                if constexpr (BRICK) {
                    if (dense && cand) touch(3, (uint64_t)vox * A.row_bytes + A.sigma_off);
                    if (dense) settle(cand ? (uint32_t)*reinterpret_cast<const uint16_t *>(A.rows + (int64_t)vox * A.row_bytes + A.sigma_off) : 0u);
                }
            };
            if constexpr (MODE == 3) {
                // no colour: the networks supply it (render_nerf_results)
            } else if (depth_mode()) {
                settle_from_memory();
                if (dense) o0 += weight * t;
            } else if (ablate(1)) {
                settle_from_memory();
            } else if constexpr (BASIS >= 16) {
                // SH16 / SH25 (a row is 96 / 150 bytes: per-lane rows would need 24 / 39 registers and spill -- 6234 against 8687 Mrays/s for
                // SH16, 2453 against 4062 for SH25): the wavefront evaluates the samples cooperatively, one lane per (sample, channel):
                // 21 samples x 3 channels per pass.  Each task lane pulls the sample's weight, voxel and
                // SH basis from the owning lane (ds_bpermute), loads its channel's coefficients, and
                // returns weight / (1 + exp(-dot)) to the owner, which accumulates in sample order.
                const int n_dense = __popcll(dense_mask);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dense_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dense_mask, 0u));
                uint32_t *map = s_map + (threadIdx.x & ~63);
                if (dense) map[rank] = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                const int my_s = lane / 3, my_c = lane - 3 * my_s;
                for (int base = 0; base < n_dense; base += 21) {
                    const int smp = base + my_s;
                    const bool task = my_s < 21 && smp < n_dense;
                    stat(10, task);  // colour passes and their busy task lanes
                    const int owner = task ? (int)map[smp] : lane;
                    const float w = lane_read(weight, owner);
                    uint32_t vx = lane_read(vox, owner);
                    if (ablate(4)) vx &= 0xffffu;  // diagnostics: rows served from cache (wrong colours)
                    float b[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) b[k] = wave_ray[k * BLOCK + owner];  // the owner's SH basis, from LDS
                    float v = 0.f;
                    if (task) {
                        constexpr int NW = CHAN_BYTES / 4;
                        ChanWords<NW> cw;
                        cw = *reinterpret_cast<const ChanWords<NW> *>(A.rows + (int64_t)vx * ROW_BYTES + my_c * CHAN_BYTES);
                        auto coef = [&](int k) -> float {
                            const uint32_t wd = cw.w[k >> 1];
                            return half_bits_to_float((uint16_t)((k & 1) ? (wd >> 16) : (wd & 0xffffu)));
                        };
                        const float tmp = sh_channel<BASIS>(b, coef, 0);
                        if constexpr (MODE == 4) {
                            // colour-only arithmetic: it feeds no branch (opacity, transmittance and the step sequence stay exact),
                            // so hardware exp2 / rcp (about 1 ulp each) move a colour by ~1e-7 and nothing else
                            const float e = __builtin_amdgcn_exp2f(tmp * -1.44269504088896341f);
                            v = w * __builtin_amdgcn_rcpf(1.f + e);
                        } else {
                            v = w / (1.f + exact_expf(-tmp, s_exp));
                        }
                    }
                    const int rl = rank - base;
                    const bool mine = dense && rl >= 0 && rl < 21;
                    const int from = mine ? 3 * rl : lane;
                    const float v0 = lane_read(v, from), v1 = lane_read(v, mine ? from + 1 : lane), v2 = lane_read(v, mine ? from + 2 : lane);
                    if (mine) {
                        o0 += v0;
                        o1 += v1;
                        o2 += v2;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else if constexpr (BASIS >= 1) {
                // SH: every dense lane reads its sample's row (three channel blocks of one 64-byte row: one line fill) and evaluates the
                // three channels itself, so all rows of an iteration are requested at once.  (Until round 2 the wavefront shared the work,
                // one lane per (sample, channel), 21 samples per pass: more lanes busy per VALU instruction, but an iteration with 22+
                // dense lanes waited for two or three passes' row misses one after the other -- LAB_NOTEBOOK.md.)
                if (dense) {
                    stat(10, true);
                    uint32_t vx = vox;
                    if (ablate(4)) vx &= 0xffffu;  // diagnostics: rows served from cache (wrong colours)
                    const uint8_t *row = A.rows + (int64_t)vx * ROW_BYTES;
                    touch(3, (uint64_t)vx * ROW_BYTES);
                    if constexpr ((kShadow & 32) != 0) {  // shadow load: one dword of the same row of the copy (one more line fill)
                        const uint32_t w2 = *reinterpret_cast<const uint32_t *>(K.shadow_rows + (int64_t)vx * ROW_BYTES);
                        asm volatile("" ::"v"(w2));
                    }
                    // the whole row -- 3 * BASIS coefficient halfs, then sigma -- with 16-byte loads (SH9: four; until round 5 three channel
                    // blocks of 20 bytes each took six load instructions)
                    constexpr int RW = ROW_BYTES / 4;
                    uint32_t rw[RW];
                    if constexpr (RW >= 4) {
#pragma unroll
                        for (int u = 0; u < RW / 4; ++u) {
                            const uint4 x = reinterpret_cast<const uint4 *>(row)[u];
                            rw[4 * u] = x.x;
                            rw[4 * u + 1] = x.y;
                            rw[4 * u + 2] = x.z;
                            rw[4 * u + 3] = x.w;
                        }
                    } else {
                        const uint2 x = *reinterpret_cast<const uint2 *>(row);
                        rw[0] = x.x;
                        rw[1] = x.y;
                    }
                    if constexpr (MODE == 1) {
                        if (K.count_stats == 2) {
                            const unsigned long long now = phase_clock();   // the rows have arrived
                            ph_row += now - ph_mark;
                            ph_mark = now;
                        }
                    }
                    auto half_at = [&](int hidx) -> uint32_t { return (hidx & 1) ? (rw[hidx >> 1] >> 16) : (rw[hidx >> 1] & 0xffffu); };
                    if constexpr (BRICK) settle(half_at(3 * BASIS));
                    float b[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) b[k] = my_ray[k * BLOCK];
                    auto chan = [&](int c) -> float {
                        auto coef = [&](int k) -> float { return half_bits_to_float((uint16_t)half_at(k)); };
                        const float tmp = sh_channel<BASIS>(b, coef, c * BASIS);
                        if constexpr (MODE == 4) {
                            // colour-only arithmetic: it feeds no branch (opacity, transmittance and the step sequence stay exact),
                            // so hardware exp2 / rcp (about 1 ulp each) move a colour by ~1e-7 and nothing else
                            const float e = __builtin_amdgcn_exp2f(tmp * -1.44269504088896341f);
                            return weight * __builtin_amdgcn_rcpf(1.f + e);
                        } else {
                            return weight / (1.f + exact_expf(-tmp, s_exp));
                        }
                    };
                    if (dense) {  // (a candidate may have turned out not to be dense)
                        o0 += chan(0);
                        o1 += chan(1);
                        o2 += chan(2);
                    }
                }
            } else {
                // RGBA rows (rt_core.cuh:285-290): three halfs per voxel (the fourth half of the row is the voxel's sigma), per-lane
                if (dense) {
                    const uint2 qd = *reinterpret_cast<const uint2 *>(A.rows + (int64_t)vox * ROW_BYTES);
                    if constexpr (BRICK) settle(qd.y >> 16);
                    if (dense) {
                        o0 += half_bits_to_float((uint16_t)(qd.x & 0xffffu)) * weight;
                        o1 += half_bits_to_float((uint16_t)(qd.x >> 16)) * weight;
                        o2 += half_bits_to_float((uint16_t)(qd.y & 0xffffu)) * weight;
                    }
                }
            }
            if constexpr (BRICK && MODE == 2) track_update();  // (lanes of an iteration without any dense sample: below)
            if (dense) {
                T *= att;  // rt_core.cuh:293-307
                if (T < P.stop_thresh) {
                    fin = 2;  // early stop: renormalised and written in flush_finished()
                    alive = false;
                }
            }
        }
        if constexpr (BRICK && MODE == 2) {
            if (dense_mask == 0) track_update();
        }
        t += delta_t;  // 0 for lanes that did not step
        if constexpr (MODE == 1) {
            if (K.count_stats == 2) ph_colour += phase_clock() - ph_mark;  // (iterations without a dense lane: the ballot and the transmittance update)
        }
    }
    if constexpr (MODE == 1) {
        if (K.count_stats == 2 && lane == 0) {
            atomicAdd(&K.stats[16], ph_lookup);
            atomicAdd(&K.stats[17], ph_step);
            atomicAdd(&K.stats[18], ph_row);
            atomicAdd(&K.stats[19], ph_colour);
            atomicAdd(&K.stats[20], phase_clock() - ph_begin);
            atomicAdd(&K.stats[21], ph_steps);
        }
    }
    if constexpr (MODE == 1) {
        if (K.timeline && lane == 0) K.timeline[(size_t)K.timeline_tiles * 4 + (size_t)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 2 + 1] = wall_clock64();
    }
}

}  // namespace mnv
