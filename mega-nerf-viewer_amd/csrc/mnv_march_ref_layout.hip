// mnv_march_ref_layout.hip -- mnv_render_voxels: the march on the reference's own
// array layout (data [cap][8][data_dim] f16, child [cap][8] i32 relative).
//
// This is the direct replacement of render_voxels_kernel
// (reference src/cuda/renderer_kernel.cu:243-292 -> include/cuda/rt_core.cuh:162-332)
// for callers that hand over reference-layout device arrays, including the
// refinement trackers (rt_core.cuh:237-252,308-321) and visit marks (:132-134).
// One lane per ray, 8x8-pixel tile per wavefront, 4 wavefronts per workgroup.  Every step restarts the descent from the
// root as the reference does; unless visit marks are wanted, the first three levels of that descent come from a 512-cell
// table each workgroup derives from `child` when it starts (x*2, floorf and x - floorf(x) are exact, so entering the
// descent at level 4 with fract(pos * 8) gives the same leaf and the same in-leaf coordinates bit for bit).
// The tuned path (packed layout, LDS top grid, persistent waves) is
// mnv_march_accel.hip; both produce bit-identical pixels.
#include "mnv_device.h"
#include "mnv_internal.h"

#pragma clang fp contract(off)

namespace mnv {

#ifndef MNV_REF_TOP_LEVEL
#define MNV_REF_TOP_LEVEL 3
#endif
constexpr uint32_t kTopLeaf = 0x80000000u;  // top-table word: leaf at depth (word >> 28) & 7, voxel index in the low 28 bits

template <int BASIS /* -1 RGBA, 0 DC-only with runtime stride, 1/4/9/16/25 */>
__global__ __launch_bounds__(256) void march_ref_layout_kernel(const MarchParams P) {
    __shared__ uint64_t s_exp[32];
    constexpr int TL = MNV_REF_TOP_LEVEL, TG = 1 << TL;  // table level and cells per axis
    __shared__ uint32_t s_top[TG * TG * TG];
    load_exp_table(s_exp);
    // level-3 cell -> the leaf of depth <= 3 that covers it, or the chunk holding its depth-4 voxels
    const bool use_top = !P.track_visit && P.capacity < (1 << 25);
    if (use_top) {
        for (int i = threadIdx.x; i < TG * TG * TG; i += 256) {
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
            s_top[i] = word;
        }
        __syncthreads();
    }

    // 16x16 pixel block per workgroup, 8x8 per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int by = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (bx >= P.tw || by >= P.th) return;
    const int ix = P.x0 + bx, iy = P.y0 + by;
    const int64_t p = (int64_t)by * P.tw + bx;

    constexpr int NB = BASIS > 0 ? BASIS : 1;
    RaySetup<NB> r;
    setup_ray<(BASIS > 0 ? BASIS : 0)>(P, P.cam, ix, iy, r);
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
            if (use_top) {
                const uint32_t word = s_top[((int)(pos[0] * (float)TG) * TG + (int)(pos[1] * (float)TG)) * TG + (int)(pos[2] * (float)TG)];
                if (word & kTopLeaf) {
                    depth = (int)((word >> 28) & 7u);
                    chunk = (int32_t)((word & 0x0fffffffu) >> 3);
                    cidx = (int32_t)(word & 7u);
                    const float sc = __uint_as_float((uint32_t)(127 + depth) << 23);
#pragma unroll
                    for (int i = 0; i < 3; ++i) pos[i] = __builtin_amdgcn_fractf(pos[i] * sc);
                    at_leaf = true;
                } else {
                    depth = TL + 1;
                    chunk = (int32_t)word;
#pragma unroll
                    for (int i = 0; i < 3; ++i) pos[i] = __builtin_amdgcn_fractf(pos[i] * (float)TG);
                }
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
                    pos[i] *= 2.f;
                    const float f = floorf(pos[i]);
                    cidx = cidx * 2 + (int)f;
                    pos[i] -= f;
                }
                const int32_t skip = P.child[(int64_t)chunk * 8 + cidx];
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
            const float cube = __uint_as_float((uint32_t)(127 + depth) << 23);
            const float delta_t = tu / cube + P.step_size;
            const uint16_t *row = P.data + ((int64_t)chunk * 8 + cidx) * P.data_dim;
            const float sigma = half_bits_to_float(row[P.data_dim - 1]);

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
                    if (P.sample_counts) {
                        const int16_t sc = P.sample_counts[(int64_t)chunk * 8 + cidx];
                        if (weight > max_sample_weight && sc < P.max_sample_count) {
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
                if (P.sample_counts) {
                    const int16_t sc = P.sample_counts[(int64_t)chunk * 8 + cidx];
                    if (max_sample_weight == -1.f && sc < P.max_sample_count) {
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
    dim3 grid((P.tw + 15) / 16, (P.th + 15) / 16), block(256);
    int b = P.format == 1 ? P.basis_dim : -1;
    if (P.format == 1 && b < 0) b = -1;  // SH without digits behaves like the RGBA branch (:285)
#define MNV_LAUNCH(B) hipLaunchKernelGGL(march_ref_layout_kernel<B>, grid, block, 0, stream, P)
    switch (b) {
        case -1: MNV_LAUNCH(-1); break;
        case 4: MNV_LAUNCH(4); break;
        case 9: MNV_LAUNCH(9); break;
        case 16: MNV_LAUNCH(16); break;
        case 25: MNV_LAUNCH(25); break;
        default: MNV_LAUNCH(0); break;  // any other basis_dim >= 0: DC term only (rt_core.cuh:262-281)
    }
#undef MNV_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace mnv
