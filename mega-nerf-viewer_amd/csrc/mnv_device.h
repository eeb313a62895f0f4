// mnv_device.h -- device-side building blocks shared by the march kernels.
//
// Arithmetic contract (DESIGN.md "Arithmetic spec"): every value that feeds a
// branch of the march (cell classification, t < tmax, sigma > sigma_thresh,
// light_intensity < stop_thresh) is computed with IEEE-754 binary32/binary64
// +,-,*,/,sqrt in the reference's source evaluation order with NO fused
// multiply-add contraction, and expf is the table-driven binary64 algorithm
// of glibc 2.35 (restated independently in oracle/mnv_oracle.c).  This file
// must be compiled with -ffp-contract=off; the pragma below is a second lock.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mnv {

// Per-frame part of the kernel argument block: the by-value CameraSpec / RenderOptions of the
// reference launch (renderer_kernel.cu:431-436) plus the tree's offset/scale and the outputs.
// One camera (data_spec.hpp:9-23 CameraSpec with the c2w by value) + the tree-space ray origin
// cen = offset + scale * c2w[9..11] (renderer_kernel.cu:272-275), which is the same for every ray of
// the frame and therefore computed once on the host.  80 bytes, 16-byte aligned.
struct __attribute__((aligned(16))) CamBlock {
    float fx, fy, cx, cy;
    float c2w[12];
    float cen[3];
    float pad;
};

struct FrameParams {
    CamBlock cam;
    // tile of the image rendered by this launch
    int32_t x0, y0, tw, th;
    float offset[3], scale[3];
    // march-relevant RenderOptions (render_options.hpp:9-56)
    float step_size, sigma_thresh, stop_thresh, background_brightness;
    float render_bbox[6];
    int32_t basis_min, basis_max;
    int32_t render_depth;
    // rodrigues(opt.rot_dirs) constants, precomputed on the host with libm
    // (renderer_kernel.cu:43-51): identity when rot_enabled == 0
    int32_t rot_enabled;
    float rot_k[3], rot_cos, rot_sin;
    // outputs
    float *rgba;
    uint8_t *rgba8;
    // what the reference's kernels read per pixel when they are called with offscreen == false (mnv_frame_inputs; both indexed like
    // the outputs, both may be NULL = the offscreen branch): the depth attachment (renderer_kernel.cu:277-280) and the pixel
    // that is already in the image (renderer_kernel.cu:230-234,260-264)
    const float *tmax_px;
    const uint8_t *rgba8_init;
};

// t_max of pixel p: renderer_kernel.cu:277-280
__device__ __forceinline__ float frame_tmax(const FrameParams &P, int64_t p) { return P.tmax_px ? P.tmax_px[p] : 1e9f; }

// Full argument block of the reference-layout kernel: + TreeSpec (data_spec.hpp:25-50) and
// the refinement trackers.
struct MarchParams : FrameParams {
    const uint16_t *data;
    const int32_t *child;
    const int16_t *sample_counts;
    int32_t data_dim, basis_dim, format, capacity;
    int32_t max_depth, max_sample_count;
    float *split_track;
    float *sample_track;
    int32_t *visited;
    int32_t track_visit;
    int32_t N;  // branching factor per axis (TreeSpec::N): 2 for every PlenOctree; other values take the reference-layout kernel's general walk
};

// glibc 2.35 expf table: tab[i] = asuint64(2^(i/32)) - (i << 47)
__device__ static const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

// Copy the table into LDS (32 x 8 B = one conflict-free ds_read_b64 bank row).
__device__ __forceinline__ void load_exp_table(uint64_t *lds_tab) {
    if (threadIdx.x < 32) lds_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
}

// expf, bit-compatible with glibc 2.35 (non-FMA variant); `tab` lives in LDS.
__device__ __forceinline__ float exact_expf(float x, const uint64_t *tab) {
    const uint32_t ix = __float_as_uint(x);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    constexpr double N = 32.0;
    constexpr double InvLn2N = 0x1.71547652b82fep+0 * N;
    constexpr double SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    constexpr double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    constexpr double C2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = tab[ki & 31u];
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// expf as exact_expf, without branches: the main path runs on a clamped argument (nothing overflows on the way) and the special cases
// (|x| >= 88) are selected afterwards, in exact_expf's order.  Same bits for every input; several of these in a row have no control
// flow between them, so the compiler interleaves their binary64 chains.
__device__ __forceinline__ float exact_expf_select(float x, const uint64_t *tab) {
    constexpr double N = 32.0;
    constexpr double InvLn2N = 0x1.71547652b82fep+0 * N;
    constexpr double SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    constexpr double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    constexpr double C2 = 0x1.62e42ff0c52d6p-1 / N;
    const float xc = __builtin_amdgcn_fmed3f(x, -128.f, 128.f);
    const double xd = (double)xc;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = tab[ki & 31u];
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    float res = (float)y;
    const uint32_t ix = __float_as_uint(x);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    const bool big = abstop >= 0x42bu;
    res = (big && x < -0x1.9fe368p6f) ? 0.0f : res;
    res = (big && x > 0x1.62e42ep6f) ? __uint_as_float(0x7f800000u) : res;
    res = (big && abstop >= 0x7f8u) ? x + x : res;
    res = ix == 0xff800000u ? 0.0f : res;
    return res;
}

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}

// rt_core.cuh:12-68, literal C++ arithmetic conversions (double literals
// promote only the sub-expression they appear in).
template <int BASIS>
__device__ __forceinline__ void sh_basis(const float vdir[3], float *out) {
    out[0] = (float)0.28209479177387814;
    if constexpr (BASIS >= 4) {
        const float x = vdir[0], y = vdir[1], z = vdir[2];
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        if constexpr (BASIS >= 25) {
            out[16] = (float)(2.5033429417967046 * xy * (xx - yy));
            out[17] = (float)(-1.7701307697799304 * yz * (3 * xx - yy));
            out[18] = (float)(0.9461746957575601 * xy * (7 * zz - 1.f));
            out[19] = (float)(-0.6690465435572892 * yz * (7 * zz - 3.f));
            out[20] = (float)(0.10578554691520431 * (zz * (35 * zz - 30) + 3));
            out[21] = (float)(-0.6690465435572892 * xz * (7 * zz - 3));
            out[22] = (float)(0.47308734787878004 * (xx - yy) * (7 * zz - 1.f));
            out[23] = (float)(-1.7701307697799304 * xz * (xx - 3 * yy));
            out[24] = (float)(0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)));
        }
        if constexpr (BASIS >= 16) {
            out[9] = (float)(-0.5900435899266435 * y * (3 * xx - yy));
            out[10] = (float)(2.890611442640554 * xy * z);
            out[11] = (float)(-0.4570457994644658 * y * (4 * zz - xx - yy));
            out[12] = (float)(0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy));
            out[13] = (float)(-0.4570457994644658 * x * (4 * zz - xx - yy));
            out[14] = (float)(1.445305721320277 * z * (xx - yy));
            out[15] = (float)(-0.5900435899266435 * x * (xx - 3 * yy));
        }
        if constexpr (BASIS >= 9) {
            out[4] = (float)(1.0925484305920792 * xy);
            out[5] = (float)(-1.0925484305920792 * yz);
            out[6] = (float)(0.31539156525252005 * (2.0 * zz - xx - yy));
            out[7] = (float)(-1.0925484305920792 * xz);
            out[8] = (float)(0.5462742152960396 * (xx - yy));
        }
        out[1] = (float)(-0.4886025119029199 * y);
        out[2] = (float)(0.4886025119029199 * z);
        out[3] = (float)(-0.4886025119029199 * x);
    }
}

// World-space unit ray direction and the (rodrigues-rotated) view direction of pixel (ix, iy): what the sample-emitting march
// multiplies z with and writes into the view-direction columns (renderer_kernel.cu:348-351 -> :30-38, :40-61).
__device__ __forceinline__ void world_ray_dirs(const FrameParams &P, const CamBlock &C, int ix, int iy, float true_dir[3], float vdir[3]) {
    const float *m = C.c2w;
    const float xyz0 = (ix + 0.5f - C.cx) / C.fx, xyz1 = -(iy + 0.5f - C.cy) / C.fy, xyz2 = -1.0f;
    true_dir[0] = m[0] * xyz0 + m[3] * xyz1 + m[6] * xyz2;
    true_dir[1] = m[1] * xyz0 + m[4] * xyz1 + m[7] * xyz2;
    true_dir[2] = m[2] * xyz0 + m[5] * xyz1 + m[8] * xyz2;
    const float inv = 1.f / sqrtf(true_dir[0] * true_dir[0] + true_dir[1] * true_dir[1] + true_dir[2] * true_dir[2]);
    for (int i = 0; i < 3; ++i) {
        true_dir[i] *= inv;
        vdir[i] = true_dir[i];
    }
    if (P.rot_enabled) {
        const float *k = P.rot_k;
        float cross[3];
        cross[0] = k[1] * vdir[2] - k[2] * vdir[1];
        cross[1] = k[2] * vdir[0] - k[0] * vdir[2];
        cross[2] = k[0] * vdir[1] - k[1] * vdir[0];
        const float dot = k[0] * vdir[0] + k[1] * vdir[1] + k[2] * vdir[2];
        for (int i = 0; i < 3; ++i)
            vdir[i] = (float)((double)(vdir[i] * P.rot_cos + cross[i] * P.rot_sin) + (double)(k[i] * dot) * (1.0 - (double)P.rot_cos));
    }
}

// Per-ray constants produced by ray generation + march set-up.
template <int NB>
struct RaySetup {
    float dir[3];     // tree-space unit direction (after _get_delta_scale)
    float invdir[3];  // rt_core.cuh:189
    float basis[NB];  // SH basis of the (rotated) view direction, minmax-masked
    float delta_scale;
    float tmin, tmax;
    bool in_bbox;
};

// renderer_kernel.cu:30-38 (screen2worlddir), :272-275, :282-283 (rodrigues),
// rt_core.cuh:182-209.  BASIS: number of SH basis functions kept in registers
// (1 for DC-only / RGBA).
template <int BASIS>
__device__ __forceinline__ void setup_ray(const FrameParams &P, const CamBlock &C, int ix, int iy,
                                          RaySetup<(BASIS > 0 ? BASIS : 1)> &r, float t_max = 1e9f) {
    const float xyz0 = (ix + 0.5f - C.cx) / C.fx;
    const float xyz1 = -(iy + 0.5f - C.cy) / C.fy;
    const float xyz2 = -1.0f;
    const float *m = C.c2w;
    float dir[3];
    dir[0] = m[0] * xyz0 + m[3] * xyz1 + m[6] * xyz2;
    dir[1] = m[1] * xyz0 + m[4] * xyz1 + m[7] * xyz2;
    dir[2] = m[2] * xyz0 + m[5] * xyz1 + m[8] * xyz2;
    const float invnorm = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= invnorm;
    dir[1] *= invnorm;
    dir[2] *= invnorm;

    float vdir[3] = {dir[0], dir[1], dir[2]};
    if (P.rot_enabled) {  // renderer_kernel.cu:52-60
        const float *k = P.rot_k;
        float cross[3];
        cross[0] = k[1] * vdir[2] - k[2] * vdir[1];
        cross[1] = k[2] * vdir[0] - k[0] * vdir[2];
        cross[2] = k[0] * vdir[1] - k[1] * vdir[0];
        const float dot = k[0] * vdir[0] + k[1] * vdir[1] + k[2] * vdir[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            vdir[i] = (float)((double)(vdir[i] * P.rot_cos + cross[i] * P.rot_sin) +
                              (double)(k[i] * dot) * (1.0 - (double)P.rot_cos));
        }
    }

    // _get_delta_scale, rt_core.cuh:102-115
    dir[0] *= P.scale[0];
    dir[1] *= P.scale[1];
    dir[2] *= P.scale[2];
    const float delta_scale = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    r.delta_scale = delta_scale;
    const float tmax_bg = t_max / delta_scale;  // :183 (t_max: 1e9f offscreen, the pixel's depth otherwise, renderer_kernel.cu:277-280)

    float tmin = 0.0f, tmax = 1e4f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        r.dir[i] = dir[i];
        r.invdir[i] = (float)(1.0 / ((double)dir[i] + 1e-9));  // :189
        const double inv = (double)r.invdir[i];
        const float t1 = (float)(((double)P.render_bbox[i] + 1e-6 - (double)C.cen[i]) * inv);
        const float t2 = (float)(((double)P.render_bbox[i + 3] - 1e-6 - (double)C.cen[i]) * inv);
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    tmax = fminf(tmax, tmax_bg);
    r.tmin = tmin;
    r.tmax = tmax;
    r.in_bbox = !(tmax < 0 || tmin > tmax);

    constexpr int NB = (BASIS > 0 ? BASIS : 1);
    if constexpr (BASIS > 0) {
        sh_basis<BASIS>(vdir, r.basis);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (i < P.basis_min || i > P.basis_max) r.basis[i] = 0.f;  // :203-209
    } else {
        r.basis[0] = 0.f;
    }
}

// Colour of one dense sample from a row of binary16 coefficients (rt_core.cuh:257-291).
// `coef(k)` returns coefficient k of the row as float.  Summation order: DC, then the
// groups 16-24, 9-15, 4-8, 1-3, each summed left to right before being added.
template <int BASIS, typename F>
__device__ __forceinline__ float sh_channel(const float *b, F coef, int off) {
    float tmp = b[0] * coef(off);
    if constexpr (BASIS >= 25) {
        tmp += b[16] * coef(off + 16) + b[17] * coef(off + 17) + b[18] * coef(off + 18) +
               b[19] * coef(off + 19) + b[20] * coef(off + 20) + b[21] * coef(off + 21) +
               b[22] * coef(off + 22) + b[23] * coef(off + 23) + b[24] * coef(off + 24);
    }
    if constexpr (BASIS >= 16) {
        tmp += b[9] * coef(off + 9) + b[10] * coef(off + 10) + b[11] * coef(off + 11) +
               b[12] * coef(off + 12) + b[13] * coef(off + 13) + b[14] * coef(off + 14) +
               b[15] * coef(off + 15);
    }
    if constexpr (BASIS >= 9) {
        tmp += b[4] * coef(off + 4) + b[5] * coef(off + 5) + b[6] * coef(off + 6) + b[7] * coef(off + 7) +
               b[8] * coef(off + 8);
    }
    if constexpr (BASIS >= 4) {
        tmp += b[1] * coef(off + 1) + b[2] * coef(off + 2) + b[3] * coef(off + 3);
    }
    return tmp;
}

// renderer_kernel.cu:237: uint8_t(v * 255) (truncating; CUDA's conversion saturates).
__device__ __forceinline__ uint32_t pack_u8(float v) {
    const float s = v * 255.f;
    if (!(s > 0.f)) return 0u;
    if (s >= 255.f) return 255u;
    return (uint32_t)s;
}

// renderer_kernel.cu:215-241, both branches, + the two output formats.  rgba8_init may be the same buffer as rgba8 (the reference
// reads and writes one surface): a pixel is read and written by the same lane.
__device__ __forceinline__ void composite_and_write(const FrameParams &P, int64_t p, float o0, float o1,
                                                    float o2, float o3) {
    const float nalpha = 1.f - o3;
    if (P.rgba8_init) {  // :230-234 (offscreen == false): over the pixel already there
        const uint32_t px = reinterpret_cast<const uint32_t *>(P.rgba8_init)[p];
        o0 += (float)(int)(px & 0xffu) / 255.f * nalpha;
        o1 += (float)(int)((px >> 8) & 0xffu) / 255.f * nalpha;
        o2 += (float)(int)((px >> 16) & 0xffu) / 255.f * nalpha;
    } else {  // :225-229
        const float remain = P.background_brightness * nalpha;
        o0 += remain;
        o1 += remain;
        o2 += remain;
    }
    if (P.rgba) reinterpret_cast<float4 *>(P.rgba)[p] = make_float4(o0, o1, o2, o3);
    if (P.rgba8)
        reinterpret_cast<uint32_t *>(P.rgba8)[p] =
            pack_u8(o0) | (pack_u8(o1) << 8) | (pack_u8(o2) << 16) | (255u << 24);
}

}  // namespace mnv
