// mnv_march_ref_layout.hip -- mnv_render_voxels: the march on the reference's own
// array layout (data [cap][8][data_dim] f16, child [cap][8] i32 relative).
//
// This is the direct replacement of render_voxels_kernel
// (reference src/cuda/renderer_kernel.cu:243-292 -> include/cuda/rt_core.cuh:162-332)
// for callers that hand over reference-layout device arrays, including the
// refinement trackers (rt_core.cuh:237-252,308-321) and visit marks (:132-134).
// One lane per ray, 8x8-pixel tile per wavefront, 2 wavefronts per workgroup.  Every step restarts the descent from the
// root as the reference does; unless visit marks are wanted, the first levels of that descent come from tables derived from
// `child` (x*2, floorf and x - floorf(x) are exact, so entering the descent at level G + 1 with fract(pos * 2^G) gives the
// same leaf and the same in-leaf coordinates bit for bit):
//  * launches of 65536 rays and more (mnv_set_ref_table_min_rays): ref_top_build_kernel, enqueued in front of the march on the caller's stream, fills a
//    dense level-G table (G = 7: 2 M cells of 8 bytes, brick-ordered; G = 6 for small trees) in stream-ordered scratch memory
//    -- cell -> the leaf of depth <= G that covers it together with that leaf's sigma, or the chunk that holds its depth-(G+1)
//    voxels -- and the 512-cell level-3 table every workgroup copies into LDS.  A step in a leaf of depth <= 3 costs no global
//    load before the colour row, one in a leaf of depth <= G one, a depth-10 leaf three child words + sigma instead of seven.
//    The table is rebuilt by EVERY call (about 10 us): the caller may have changed the tree in between, as the reference's
//    refinement does, and is not asked to say so;
//  * smaller launches (and when the scratch allocation fails): each workgroup derives the level-3 table itself.
// The tuned path (packed layout, LDS top grid, persistent waves) is
// mnv_march_accel_kernel.h; both produce bit-identical pixels.
#include <atomic>

#include "mnv_device.h"
#include "mnv_internal.h"
#include "mnv_knobs.h"

#pragma clang fp contract(off)

namespace mnv {

constexpr uint32_t kTopLeaf = 0x80000000u;  // table word x: leaf at depth (x >> 28) & 7, voxel index in the low 28 bits; y: the leaf's sigma (f16 bits)
constexpr uint32_t kNoSigma = 0xffffffffu;  // y of a table word whose sigma was not fetched

// Dense level-G table for one launch.  Workgroup b owns the level-(G-3) cell b = (x << 2L | y << L | z) and writes the 512 level-G
// cells below it to table[b * 512 + (lx << 6 | ly << 3 | lz)]; the first workgroup under every level-3 cell also writes top3.
// Cells under a leaf of depth <= 3 are never looked up (the march finds that leaf in top3) and stay unwritten.
__global__ __launch_bounds__(256) void ref_top_build_kernel(const int32_t *__restrict__ child, const uint16_t *__restrict__ data, int data_dim,
                                                            uint2 *__restrict__ top3, uint2 *__restrict__ table, int G) {
    const int L = G - 3, n = 1 << L, b = blockIdx.x;
    const int cx = b >> (2 * L), cy = (b >> L) & (n - 1), cz = b & (n - 1);
    int32_t chunk = 0;
    uint2 word = make_uint2(0u, kNoSigma);
    bool leaf = false;
    for (int l = 1; l <= L; ++l) {
        const int s3 = L - l;
        const int cidx = ((cx >> s3) & 1) << 2 | ((cy >> s3) & 1) << 1 | ((cz >> s3) & 1);
        const int32_t skip = child[(int64_t)chunk * 8 + cidx];
        if (skip == 0) {
            const int32_t vox = chunk * 8 + cidx;
            word = make_uint2(kTopLeaf | ((uint32_t)l << 28) | (uint32_t)vox, (uint32_t)data[(int64_t)vox * data_dim + data_dim - 1]);
            leaf = true;
        } else {
            chunk += skip;
            word.x = (uint32_t)chunk;
        }
        if (l == 3 || (leaf && l < 3)) {
            const int low = (1 << (L - 3)) - 1;
            if (threadIdx.x == 0 && ((cx | cy | cz) & low) == 0) top3[((cx >> (L - 3)) * 8 + (cy >> (L - 3))) * 8 + (cz >> (L - 3))] = word;
            if (leaf) return;
        }
        if (leaf) break;
    }
    for (int i = threadIdx.x; i < 512; i += 256) {
        uint2 w = word;
        if (!leaf) {
            int32_t c = chunk;
            for (int l = 1; l <= 3; ++l) {
                const int s3 = 3 - l;
                const int cidx = (((i >> 6) >> s3) & 1) << 2 | ((((i >> 3) & 7) >> s3) & 1) << 1 | (((i & 7) >> s3) & 1);
                const int32_t skip = child[(int64_t)c * 8 + cidx];
                if (skip == 0) {
                    const int32_t vox = c * 8 + cidx;
                    w = make_uint2(kTopLeaf | ((uint32_t)(L + l) << 28) | (uint32_t)vox, (uint32_t)data[(int64_t)vox * data_dim + data_dim - 1]);
                    break;
                }
                c += skip;
                w.x = (uint32_t)c;
            }
        }
        table[(int64_t)b * 512 + i] = w;
    }
}

#ifndef MNV_REF_WG_WAVES
#define MNV_REF_WG_WAVES 2   // wavefronts (8x8-pixel tiles) per workgroup: 1, 2 or 4 (cfg2, two frames in flight: 5015 / 5062 / 4936 Mrays/s)
#endif
constexpr int kRefWaves = MNV_REF_WG_WAVES, kRefThreads = 64 * kRefWaves;
constexpr int kRefBlockW = kRefWaves >= 2 ? 16 : 8, kRefBlockH = kRefWaves == 4 ? 16 : 8;

// GEN: the walk for a branching factor N != 2 (rt_core.cuh:137-143 multiplies by tree.N): no lookup tables, N^3 voxels per chunk, cube size
// N^depth by repeated multiplication (exact as long as the power is representable, like the oracle's).  N == 2 never takes it.
template <int BASIS /* -1 RGBA, 0 DC-only with runtime stride, 1/4/9/16/25 */, bool GEN = false>
__global__ __launch_bounds__(kRefThreads) void march_ref_layout_kernel(const MarchParams P, const uint2 *__restrict__ gtop3, const uint2 *__restrict__ gtab, const int G) {
    __shared__ uint64_t s_exp[32];
    constexpr int TL = 3, TG = 1 << TL;  // level and cells per axis of the table in LDS
    __shared__ uint2 s_top[TG * TG * TG];
    load_exp_table(s_exp);
    // level-3 cell -> the leaf of depth <= 3 that covers it, or the chunk holding its depth-4 voxels
    const bool use_top = !GEN && !P.track_visit && P.capacity < (1 << 25);
    const int N = GEN ? P.N : 2, N3 = GEN ? P.N * P.N * P.N : 8;
    const float fN = GEN ? (float)P.N : 2.f;
    if (use_top) {
        for (int i = threadIdx.x; i < TG * TG * TG; i += kRefThreads) {
            if (gtab) {
                s_top[i] = gtop3[i];
                continue;
            }
            int32_t chunk = 0;
            uint32_t word = 0;
            for (int l = 1; l <= TL; ++l) {
                const int s3 = TL - l;
                const int cidx = (((i >> (2 * TL)) >> s3) & 1) << 2 | ((((i >> TL) & (TG - 1)) >> s3) & 1) << 1 | (((i & (TG - 1)) >> s3) & 1);
                const int32_t skip = P.child[(int64_t)chunk * 8 + cidx];
                if (skip == 0) {
                    word = kTopLeaf | ((uint32_t)l << 28) | (uint32_t)(chunk * 8 + cidx);
                    break;
                }
                chunk += skip;
                word = (uint32_t)chunk;
            }
            s_top[i] = make_uint2(word, kNoSigma);
        }
        __syncthreads();
    }
    const float gscale = __uint_as_float((uint32_t)(127 + G) << 23);  // 2^G
    const int GL = G - 3;

    // 16x16 pixel block per workgroup, 8x8 per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x * kRefBlockW + (wave & 1) * 8 + (lane & 7);
    const int by = blockIdx.y * kRefBlockH + (wave >> 1) * 8 + (lane >> 3);
    if (bx >= P.tw || by >= P.th) return;
    const int ix = P.x0 + bx, iy = P.y0 + by;
    const int64_t p = (int64_t)by * P.tw + bx;

    constexpr int NB = BASIS > 0 ? BASIS : 1;
    RaySetup<NB> r;
    setup_ray<(BASIS > 0 ? BASIS : 0)>(P, P.cam, ix, iy, r, frame_tmax(P, p));
    if constexpr (BASIS == 0) {  // DC only: basis[0] subject to minmax mask
        r.basis[0] = (0 < P.basis_min || 0 > P.basis_max) ? 0.f : (float)0.28209479177387814;
    }

    float out0 = 0.f, out1 = 0.f, out2 = 0.f, out3 = 0.f;
    float sp_prio = (float)(P.max_depth + 1), sp_chunk = -1.f, sp_child = -1.f;
    float sa_prio = (float)(P.max_sample_count + 1), sa_chunk = -1.f, sa_child = -1.f;
    const bool track = (P.split_track != nullptr) || (P.sample_track != nullptr);

    if (!r.in_bbox) {
        if (P.render_depth) out3 = 1.f;
    } else {
        float T = 1.f, t = r.tmin;
        float max_weight = -1.f, max_sample_weight = -1.f;
        bool stopped = false;
        while (t < r.tmax) {
            float pos[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pos[i] = P.cam.cen[i] + t * r.dir[i];
                pos[i] = fmaxf(fminf(pos[i], 1.f - 1e-6f), 0.f);
            }
            int32_t chunk = 0, cidx;
            int depth = 1;
            bool at_leaf = false;
            uint32_t sigma_bits = kNoSigma;
            if (use_top) {
                uint2 word = s_top[((int)(pos[0] * (float)TG) * TG + (int)(pos[1] * (float)TG)) * TG + (int)(pos[2] * (float)TG)];
                float sc = (float)TG;
                depth = TL + 1;
                if (gtab && !(word.x & kTopLeaf)) {
                    const int gx = (int)(pos[0] * gscale), gy = (int)(pos[1] * gscale), gz = (int)(pos[2] * gscale);
                    const uint32_t cell = ((uint32_t)(gx >> 3) << (2 * GL)) | ((uint32_t)(gy >> 3) << GL) | (uint32_t)(gz >> 3);
                    word = gtab[(size_t)cell * 512u + (uint32_t)((gx & 7) << 6 | (gy & 7) << 3 | (gz & 7))];
                    sc = gscale;
                    depth = G + 1;
                }
                if (word.x & kTopLeaf) {
                    depth = (int)((word.x >> 28) & 7u);
                    chunk = (int32_t)((word.x & 0x0fffffffu) >> 3);
                    cidx = (int32_t)(word.x & 7u);
                    sigma_bits = word.y;
                    sc = __uint_as_float((uint32_t)(127 + depth) << 23);
                    at_leaf = true;
                } else {
                    chunk = (int32_t)word.x;
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) pos[i] = __builtin_amdgcn_fractf(pos[i] * sc);
            }
            if (!at_leaf)
            for (;;) {
                // rt_core.cuh:132-134 marks with atomicCAS(&visited[chunk], 0, 1); the mark only ever goes 0 -> 1, so a
                // load and a conditional plain store leave the same array -- without every ray serialising on the
                // root's word (measured: 974 ms -> about the unmarked frame time on the cfg2 frame)
                if (P.track_visit && __hip_atomic_load(&P.visited[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) P.visited[chunk] = 1;
                cidx = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pos[i] *= fN;
                    const float f = floorf(pos[i]);
                    cidx = cidx * N + (int)f;
                    pos[i] -= f;
                }
                const int32_t skip = P.child[(int64_t)chunk * N3 + cidx];
                if (skip == 0) break;
                ++depth;
                chunk += skip;
            }
            float tu = 1e4f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t1 = -pos[i] * r.invdir[i];
                const float t2 = t1 + r.invdir[i];
                tu = fminf(tu, fmaxf(t1, t2));
            }
            // powf(2, depth) is exactly 2^depth; dividing by a power of two is exact scaling
            float cube = __uint_as_float((uint32_t)(127 + depth) << 23);
            if constexpr (GEN) {
                cube = 1.f;
                for (int i = 0; i < depth; ++i) cube *= fN;
            }
            const float delta_t = tu / cube + P.step_size;
            const uint16_t *row = P.data + ((int64_t)chunk * N3 + cidx) * P.data_dim;
            const float sigma = half_bits_to_float(sigma_bits != kNoSigma ? (uint16_t)sigma_bits : row[P.data_dim - 1]);

            if (sigma > P.sigma_thresh) {
                const float att = exact_expf(-delta_t * r.delta_scale * sigma, s_exp);
                const float weight = T * (1.f - att);
                if (track) {
                    if (weight > max_weight && depth < P.max_depth) {
                        sp_chunk = (float)chunk;
                        sp_child = (float)cidx;
                        sp_prio = (float)depth;
                        max_weight = weight;
                    }
                    if (P.sample_counts && weight > max_sample_weight) {  // (the count is gathered only where it can decide something)
                        const int16_t sc = P.sample_counts[(int64_t)chunk * N3 + cidx];
                        if (sc < P.max_sample_count) {
                            sa_chunk = (float)chunk;
                            sa_child = (float)cidx;
                            sa_prio = (float)sc;
                            max_sample_weight = weight;
                        }
                    }
                }
                if (P.render_depth) {
                    out0 += weight * t;
                } else if constexpr (BASIS >= 0) {
                    auto coef = [&](int k) { return half_bits_to_float(row[k]); };
                    const int stride = BASIS > 0 ? BASIS : P.basis_dim;
                    const float c0 = sh_channel<BASIS>(r.basis, coef, 0);
                    const float c1 = sh_channel<BASIS>(r.basis, coef, stride);
                    const float c2 = sh_channel<BASIS>(r.basis, coef, 2 * stride);
                    out0 += weight / (1.f + exact_expf(-c0, s_exp));
                    out1 += weight / (1.f + exact_expf(-c1, s_exp));
                    out2 += weight / (1.f + exact_expf(-c2, s_exp));
                } else {
                    out0 += half_bits_to_float(row[0]) * weight;
                    out1 += half_bits_to_float(row[1]) * weight;
                    out2 += half_bits_to_float(row[2]) * weight;
                }
                T *= att;
                if (T < P.stop_thresh) {
                    if (P.render_depth) out0 = out1 = out2 = fminf(out0 * 0.3f, 1.0f);
                    const float s = 1.f / (1.f - T);
                    out0 *= s;
                    out1 *= s;
                    out2 *= s;
                    out3 = 1.f;
                    stopped = true;
                    break;
                }
            } else if (track) {
                if (max_weight == -1.f && depth < P.max_depth) {
                    sp_chunk = (float)chunk;
                    sp_child = (float)cidx;
                    sp_prio = (float)depth;
                }
                if (P.sample_counts && max_sample_weight == -1.f) {
                    const int16_t sc = P.sample_counts[(int64_t)chunk * N3 + cidx];
                    if (sc < P.max_sample_count) {
                        sa_chunk = (float)chunk;
                        sa_child = (float)cidx;
                        sa_prio = (float)sc;
                    }
                }
            }
            t += delta_t;
        }
        if (!stopped) {
            if (P.render_depth) {
                out0 = out1 = out2 = fminf(out0 * 0.3f, 1.0f);
                out3 = 1.f;
            } else {
                out3 = 1.f - T;
            }
        }
    }
    composite_and_write(P, p, out0, out1, out2, out3);
    // rows are pre-filled with -1 by the caller (cuda_renderer.cpp:97-98); the reference
    // overwrites chunk/child only when a candidate was found, which leaves that -1
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

int launch_ref_layout(const MarchParams &P, hipStream_t stream) {
    if (P.tw <= 0 || P.th <= 0) return 0;
    dim3 grid((P.tw + kRefBlockW - 1) / kRefBlockW, (P.th + kRefBlockH - 1) / kRefBlockH), block(kRefThreads);
    int b = P.format == 1 ? P.basis_dim : -1;
    if (P.format == 1 && b < 0) b = -1;  // SH without digits behaves like the RGBA branch (:285)
    const int64_t min_rays = ref_table_min_rays();  // 65536; the test-hook build can move it (mnv_knobs.h)
    const bool gen = P.N != 2;
    const bool big = !gen && min_rays >= 0 && (int64_t)P.tw * P.th >= min_rays;
    uint2 *scratch = nullptr;
    int G = 7;
    // the per-launch lookup table (see the head of this file): stream-ordered scratch, released behind the march
    if (big && !P.track_visit && P.capacity < (1 << 25)) {
        G = P.capacity < (1 << 15) ? 6 : 7;
        const size_t cells = ((size_t)1 << (3 * G)) + 512;
        if (hipMallocAsync(reinterpret_cast<void **>(&scratch), cells * sizeof(uint2), stream) != hipSuccess) {
            (void)hipGetLastError();
            scratch = nullptr;
        } else {
            hipLaunchKernelGGL(ref_top_build_kernel, dim3(1u << (3 * (G - 3))), dim3(256), 0, stream, P.child, P.data, P.data_dim, scratch, scratch + 512, G);
        }
    }
    const uint2 *gtop3 = scratch, *gtab = scratch ? scratch + 512 : nullptr;
#define MNV_LAUNCH(B)                                                                                                   \
    do {                                                                                                                \
        if (gen) hipLaunchKernelGGL((march_ref_layout_kernel<B, true>), grid, block, 0, stream, P, gtop3, gtab, G);     \
        else hipLaunchKernelGGL((march_ref_layout_kernel<B, false>), grid, block, 0, stream, P, gtop3, gtab, G);        \
    } while (0)
    switch (b) {
        case -1: MNV_LAUNCH(-1); break;
        case 4: MNV_LAUNCH(4); break;
        case 9: MNV_LAUNCH(9); break;
        case 16: MNV_LAUNCH(16); break;
        case 25: MNV_LAUNCH(25); break;
        default: MNV_LAUNCH(0); break;  // any other basis_dim >= 0: DC term only (rt_core.cuh:262-281)
    }
#undef MNV_LAUNCH
    const int rc = (int)hipGetLastError();
    if (scratch) (void)hipFreeAsync(scratch, stream);
    return rc;
}

}  // namespace mnv

