/*
 * mnv_oracle.c -- CPU parity oracle (plain C) for the N3Tree ray-march path.
 * TEST INFRASTRUCTURE ONLY -- see mnv_oracle.h for scope, citations, the
 * arithmetic specification and the pinning status.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off is part of the specification, not an optimisation choice:
 * every control-flow-relevant value (cell classification, t < tmax, the
 * light_intensity < stop_thresh early stop) must be reproduced bit for bit.
 */
#include "mnv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* CUDA min/max on floats (rt_core.cuh uses the unqualified device overloads). */
static uint64_t *g_depth_hist = NULL; /* analysis hook, see orc_set_depth_histogram */
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

float orc_half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        /* subnormal half: normalise */
        int e = -1;
        do { man <<= 1; ++e; } while (!(man & 0x400u));
        man &= 0x3ffu;
        return u2f(sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13));
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 127 - 15) << 23) | (man << 13));
}

uint16_t orc_float_to_half(float f) {
    const uint32_t x = f2u(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u | ((ax >> 13) & 0x3ffu) : 0));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return sign;                        /* <= 2^-25 -> 0 (ties to even) */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | (base + q)); /* carry into the exponent is correct by construction */
}

/* ---------------------------------------------------------------- orc_expf */
/*
 * glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c (the upstream is ARM
 * optimized-routines math/expf.c, EXP2F_TABLE_BITS = 5, non-TOINT_INTRINSICS
 * path): everything in binary64, one rounding to binary32 at the end.
 * tab[i] = asuint64(2^(i/32)) - (i << 47); regenerated from first principles
 * (tests/test_oracle_math.py re-derives it) and checked against libm's expf.
 */
static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

float orc_expf(float x) {
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / N;

    const uint32_t ix = f2u(x);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) { /* |x| >= 88 or nan */
        if (ix == 0xff800000u) return 0.0f;          /* -inf */
        if (abstop >= 0x7f8u) return x + x;           /* +inf, nan */
        if (x > 0x1.62e42ep6f) return INFINITY;       /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;          /* underflow */
        /* the remaining [-103.97, -88) u (88, 88.72] range takes the main path */
    }
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = d2u(kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = EXP2F_TAB[ki % 32u];
    t += ki << (52 - 5);
    const double s = u2d(t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

/* -------------------------------------------------------------- SH basis */
/* rt_core.cuh:12-68.  Double literals promote only the sub-expression they
 * appear in; xx..xz are scalar_t (float) products.  Entries that the reference
 * leaves uninitialised are set to 0 here (they are never read: the colour
 * switch at rt_core.cuh:263-281 uses the same fallthrough structure). */
void orc_sh_basis(int basis_dim, const float vdir[3], float out[ORC_BASIS_MAX]) {
    for (int i = 0; i < ORC_BASIS_MAX; ++i) out[i] = 0.f;
    out[0] = (float)0.28209479177387814;
    const float x = vdir[0], y = vdir[1], z = vdir[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    switch (basis_dim) {
        case 25:
            out[16] = (float)(2.5033429417967046 * xy * (xx - yy));
            out[17] = (float)(-1.7701307697799304 * yz * (3 * xx - yy));
            out[18] = (float)(0.9461746957575601 * xy * (7 * zz - 1.f));
            out[19] = (float)(-0.6690465435572892 * yz * (7 * zz - 3.f));
            out[20] = (float)(0.10578554691520431 * (zz * (35 * zz - 30) + 3));
            out[21] = (float)(-0.6690465435572892 * xz * (7 * zz - 3));
            out[22] = (float)(0.47308734787878004 * (xx - yy) * (7 * zz - 1.f));
            out[23] = (float)(-1.7701307697799304 * xz * (xx - 3 * yy));
            out[24] = (float)(0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)));
            /* fallthrough */
        case 16:
            out[9] = (float)(-0.5900435899266435 * y * (3 * xx - yy));
            out[10] = (float)(2.890611442640554 * xy * z);
            out[11] = (float)(-0.4570457994644658 * y * (4 * zz - xx - yy));
            out[12] = (float)(0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy));
            out[13] = (float)(-0.4570457994644658 * x * (4 * zz - xx - yy));
            out[14] = (float)(1.445305721320277 * z * (xx - yy));
            out[15] = (float)(-0.5900435899266435 * x * (xx - 3 * yy));
            /* fallthrough */
        case 9:
            out[4] = (float)(1.0925484305920792 * xy);
            out[5] = (float)(-1.0925484305920792 * yz);
            out[6] = (float)(0.31539156525252005 * (2.0 * zz - xx - yy));
            out[7] = (float)(-1.0925484305920792 * xz);
            out[8] = (float)(0.5462742152960396 * (xx - yy));
            /* fallthrough */
        case 4:
            out[1] = (float)(-0.4886025119029199 * y);
            out[2] = (float)(0.4886025119029199 * z);
            out[3] = (float)(-0.4886025119029199 * x);
            break;
        default:
            break;
    }
}

/* ------------------------------------------------------------ camera pose */
/* camera.cpp:54-82 with glm semantics (3rdparty/glm/glm/detail/func_geometric.inl:
 * normalize(v) = v * (1/sqrt(dot(v,v))), dot = x*x + y*y + z*z left to right,
 * cross as at :74-77). */
static void glm_normalize3(const float v[3], float o[3]) {
    const float d = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float inv = 1.0f / sqrtf(d);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}
static void glm_cross3(const float x[3], const float y[3], float o[3]) {
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}
void orc_camera_pose(const float center[3], const float v_back_in[3], const float v_world_up[3],
                     float c2w[12]) {
    float back[3], right[3], up[3], tmp[3];
    glm_normalize3(v_back_in, back);
    glm_cross3(v_world_up, back, tmp);
    glm_normalize3(tmp, right);
    glm_cross3(back, right, up);
    for (int i = 0; i < 3; ++i) {
        c2w[0 + i] = right[i];
        c2w[3 + i] = up[i];
        c2w[6 + i] = back[i];
        c2w[9 + i] = center[i];
    }
}

void orc_default_options(orc_options *o) {
    memset(o, 0, sizeof(*o));
    o->step_size = 1e-4f;
    o->sigma_thresh = 1e-2f;
    o->stop_thresh = 1e-2f;
    o->background_brightness = 1.f;
    o->render_bbox[0] = o->render_bbox[1] = o->render_bbox[2] = 0.f;
    o->render_bbox[3] = o->render_bbox[4] = o->render_bbox[5] = 1.f;
    o->basis_minmax[0] = 0;
    o->basis_minmax[1] = ORC_BASIS_MAX - 1;
    o->show_grid = false;
    o->grid_max_depth = 4;
    o->render_depth = false;
    o->use_splitting = false;
    o->use_guided_sampling = false;
    o->max_depth = 16;
    o->samples_per_corner = 8;
    o->split_batch_size = 4192;
    o->nerf_batch_size = 1024;
    o->max_sample_count = 256;
    o->need_viewdir = false;
    o->appearance_embedding = -1;
    o->max_guided_samples = 128;
}

uint64_t orc_algorithmic_bytes(const orc_counters *c, int32_t format, int32_t basis_dim) {
    /* colour halfs read per dense sample: 3*basis_dim in SH mode, 3 in RGBA mode */
    const uint64_t per_hit = (format == 1 && basis_dim >= 0) ? 6ull * (uint64_t)basis_dim : 6ull;
    return 16ull * c->rays + 4ull * c->levels + 2ull * c->steps + per_hit * c->hits;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ----------------------------------------------------------- per-ray march */

typedef struct {
    uint64_t steps, levels, hits;
    int in_bbox, early_stop;
} ray_stats;

/* renderer_kernel.cu:40-61 */
static void rodrigues(const float aa[3], float dir[3]) {
    float angle = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if ((double)angle < 1e-6) return;
    float k[3];
    for (int i = 0; i < 3; ++i) k[i] = aa[i] / angle;
    const float cos_angle = cosf(angle), sin_angle = sinf(angle);
    float cross[3];
    cross[0] = k[1] * dir[2] - k[2] * dir[1];
    cross[1] = k[2] * dir[0] - k[0] * dir[2];
    cross[2] = k[0] * dir[1] - k[1] * dir[0];
    const float dot = k[0] * dir[0] + k[1] * dir[1] + k[2] * dir[2];
    for (int i = 0; i < 3; ++i) {
        dir[i] = (float)((double)(dir[i] * cos_angle + cross[i] * sin_angle) +
                         (double)(k[i] * dot) * (1.0 - (double)cos_angle));
    }
}

/* rt_core.cuh:162-332.  `dir` is modified (as in the reference). */
static void trace_ray(const orc_tree *tree, int32_t *visited, float dir[3], const float vdir[3],
                      const float cen[3], const orc_options *opt, float tmax_bg, float out[4],
                      float *split_chunk, float *split_child, float *split_prio,
                      float *sample_chunk, float *sample_child, float *sample_prio,
                      int track_visit, ray_stats *st) {
    const int N = tree->N;
    const int N3 = N * N * N;
    const int data_dim = tree->data_dim;
    const int basis_dim = tree->basis_dim;

    *split_prio = (float)(opt->max_depth + 1);          /* :179 */
    *sample_prio = (float)(opt->max_sample_count + 1);  /* :180 */

    /* _get_delta_scale :102-115 */
    dir[0] *= tree->scale[0];
    dir[1] *= tree->scale[1];
    dir[2] *= tree->scale[2];
    const float delta_scale = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    tmax_bg /= delta_scale; /* :183 */

    float invdir[3];
    for (int i = 0; i < 3; ++i) invdir[i] = (float)(1.0 / ((double)dir[i] + 1e-9)); /* :189 */

    /* _dda_world :70-86 */
    float tmin = 0.0f, tmax = 1e4f;
    for (int i = 0; i < 3; ++i) {
        const float t1 = (float)(((double)opt->render_bbox[i] + 1e-6 - (double)cen[i]) * (double)invdir[i]);
        const float t2 = (float)(((double)opt->render_bbox[i + 3] - 1e-6 - (double)cen[i]) * (double)invdir[i]);
        tmin = fmaxf_(tmin, fminf_(t1, t2));
        tmax = fminf_(tmax, fmaxf_(t1, t2));
    }
    tmax = fminf_(tmax, tmax_bg); /* :192 */

    if (tmax < 0 || tmin > tmax) { /* :194-197 */
        if (opt->render_depth) out[3] = 1.f;
        return;
    }
    st->in_bbox = 1;

    float basis_fn[ORC_BASIS_MAX];
    if (tree->format == 1) orc_sh_basis(basis_dim, vdir, basis_fn); /* :201 */
    else for (int i = 0; i < ORC_BASIS_MAX; ++i) basis_fn[i] = 0.f;
    for (int i = 0; i < opt->basis_minmax[0] && i < ORC_BASIS_MAX; ++i) basis_fn[i] = 0.f; /* :203-205 */
    for (int i = opt->basis_minmax[1] + 1; i < ORC_BASIS_MAX; ++i)                       /* :207-209 */
        if (i >= 0) basis_fn[i] = 0.f;

    float light_intensity = 1.f;
    float t = tmin;
    float max_weight = -1, max_sample_weight = -1;

    while (t < tmax) { /* :220 */
        float pos[3];
        pos[0] = cen[0] + t * dir[0];
        pos[1] = cen[1] + t * dir[1];
        pos[2] = cen[2] + t * dir[2];

        /* query_single_from_root :117-159 */
        pos[0] = fmaxf_(fminf_(pos[0], 1.f - 1e-6f), 0.f);
        pos[1] = fmaxf_(fminf_(pos[1], 1.f - 1e-6f), 0.f);
        pos[2] = fmaxf_(fminf_(pos[2], 1.f - 1e-6f), 0.f);
        int32_t chunk = 0, child_idx = 0;
        int depth = 1;
        for (;;) {
            if (track_visit && visited) {
                if (visited[chunk] == 0) visited[chunk] = 1; /* atomicCAS(&visited[i],0,1) :133 */
            }
            int cur = 0;
            for (int i = 0; i < 3; ++i) {
                pos[i] *= (float)N;
                const float idx_dimi = floorf(pos[i]);
                cur = (int)((float)(cur * N) + idx_dimi);
                pos[i] -= idx_dimi;
            }
            const int32_t skip = tree->child[(int64_t)chunk * N3 + cur];
            ++st->levels;
            if (skip == 0) { child_idx = cur; break; }
            depth += 1;
            chunk += skip;
        }
        /* powf(N, depth) :226 -- exact for the representable powers used here */
        float cube_size = 1.f;
        for (int i = 0; i < depth; ++i) cube_size *= (float)N;

        /* _dda_unit :88-100 */
        float tu = 1e4f;
        for (int i = 0; i < 3; ++i) {
            const float t1 = -pos[i] * invdir[i];
            const float t2 = t1 + invdir[i];
            tu = fminf_(tu, fmaxf_(t1, t2));
        }
        const float t_subcube = tu / cube_size;          /* :229 */
        const float delta_t = t_subcube + opt->step_size; /* :230 */
        const uint16_t *row = tree->data + ((int64_t)chunk * N3 + child_idx) * data_dim;
        const float sigma = orc_half_to_float(row[data_dim - 1]); /* :231 */
        ++st->steps;
        if (g_depth_hist) { /* analysis hook (orc_set_depth_histogram): steps by leaf depth, empty | dense */
            const int dense_step = sigma > opt->sigma_thresh;
#ifdef _OPENMP
#pragma omp atomic
#endif
            g_depth_hist[dense_step * 32 + (depth < 31 ? depth : 31)] += 1;
        }

        if (sigma > opt->sigma_thresh) { /* :233 */
            ++st->hits;
            const float att = orc_expf(-delta_t * delta_scale * sigma);
            const float weight = light_intensity * (1.f - att);

            if (weight > max_weight && depth < opt->max_depth) { /* :237-243 */
                *split_chunk = (float)chunk;
                *split_child = (float)child_idx;
                *split_prio = (float)depth;
                max_weight = weight;
            }
            if (tree->sample_counts) { /* :245-252 */
                const int16_t sc = tree->sample_counts[(int64_t)chunk * N3 + child_idx];
                if (weight > max_sample_weight && sc < opt->max_sample_count) {
                    *sample_chunk = (float)chunk;
                    *sample_child = (float)child_idx;
                    *sample_prio = (float)sc;
                    max_sample_weight = weight;
                }
            }

            if (opt->render_depth) { /* :254-256 */
                out[0] += weight * t;
            } else if (basis_dim >= 0) { /* :257-284 */
                int off = 0;
#define MB(k) (basis_fn[k] * orc_half_to_float(row[off + (k)]))
                for (int c = 0; c < 3; ++c) {
                    float tmp = basis_fn[0] * orc_half_to_float(row[off]);
                    switch (basis_dim) {
                        case 25:
                            tmp += MB(16) + MB(17) + MB(18) + MB(19) + MB(20) + MB(21) + MB(22) + MB(23) + MB(24);
                            /* fallthrough */
                        case 16:
                            tmp += MB(9) + MB(10) + MB(11) + MB(12) + MB(13) + MB(14) + MB(15);
                            /* fallthrough */
                        case 9:
                            tmp += MB(4) + MB(5) + MB(6) + MB(7) + MB(8);
                            /* fallthrough */
                        case 4:
                            tmp += MB(1) + MB(2) + MB(3);
                    }
                    out[c] += weight / (1.f + orc_expf(-tmp));
                    off += basis_dim;
                }
#undef MB
            } else { /* :285-290 */
                for (int j = 0; j < 3; ++j) out[j] += orc_half_to_float(row[j]) * weight;
            }

            light_intensity *= att; /* :293 */

            if (light_intensity < opt->stop_thresh) { /* :295-307 */
                if (opt->render_depth) out[0] = out[1] = out[2] = fminf_(out[0] * 0.3f, 1.0f);
                const float scale = 1.f / (1.f - light_intensity);
                out[0] *= scale;
                out[1] *= scale;
                out[2] *= scale;
                out[3] = 1.f;
                st->early_stop = 1;
                return;
            }
        } else { /* :308-321 */
            if (max_weight == -1 && depth < opt->max_depth) {
                *split_chunk = (float)chunk;
                *split_child = (float)child_idx;
                *split_prio = (float)depth;
            }
            if (tree->sample_counts) {
                const int16_t sc = tree->sample_counts[(int64_t)chunk * N3 + child_idx];
                if (max_sample_weight == -1 && sc < opt->max_sample_count) {
                    *sample_chunk = (float)chunk;
                    *sample_child = (float)child_idx;
                    *sample_prio = (float)sc;
                }
            }
        }
        t += delta_t; /* :323 */
    }
    if (opt->render_depth) { /* :325-330 */
        out[0] = out[1] = out[2] = fminf_(out[0] * 0.3f, 1.0f);
        out[3] = 1.f;
    } else {
        out[3] = 1.f - light_intensity;
    }
}

/* analysis hook, not part of any parity check: hist[0][d] += 1 per march step that lands in an empty leaf of depth d (sigma <= sigma_thresh),
 * hist[1][d] per dense step; NULL switches it off */
void orc_set_depth_histogram(uint64_t *hist_2x32) { g_depth_hist = hist_2x32; }

/* CPU-baseline fairness (bench.py's cpu_baseline leg, tools/cpu_ladder.py): a copy of the tree's arrays whose pages are first touched by
 * the threads that will march them -- schedule(static) over 2 MiB blocks, so the pages end up spread over the NUMA nodes of a big host
 * instead of on the node of the one thread that built the tree.  Free with orc_tree_free_copy. */
int orc_tree_copy_first_touch(const orc_tree *src, orc_tree *dst, int n_threads) {
    if (!src || !dst || !src->data || !src->child) return -1;
    const int64_t n3 = (int64_t)src->N * src->N * src->N;
    const size_t data_bytes = (size_t)src->capacity * n3 * src->data_dim * 2, child_bytes = (size_t)src->capacity * n3 * 4;
    *dst = *src;
    uint8_t *d = (uint8_t *)malloc(data_bytes), *c = (uint8_t *)malloc(child_bytes); /* large: mmap'd, untouched until written */
    if (!d || !c) {
        free(d);
        free(c);
        return -2;
    }
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    (void)n_threads;
    const size_t blk = (size_t)2 << 20;
    const int64_t nd = (int64_t)((data_bytes + blk - 1) / blk), nc = (int64_t)((child_bytes + blk - 1) / blk);
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t i = 0; i < nd; ++i) {
        const size_t o = (size_t)i * blk, n = o + blk <= data_bytes ? blk : data_bytes - o;
        memcpy(d + o, (const uint8_t *)src->data + o, n);
    }
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t i = 0; i < nc; ++i) {
        const size_t o = (size_t)i * blk, n = o + blk <= child_bytes ? blk : child_bytes - o;
        memcpy(c + o, (const uint8_t *)src->child + o, n);
    }
    dst->data = (const uint16_t *)d;
    dst->child = (const int32_t *)c;
    return 0;
}

void orc_tree_free_copy(orc_tree *t) {
    if (!t) return;
    free((void *)t->data);
    free((void *)t->child);
    t->data = NULL;
    t->child = NULL;
}

static inline uint8_t pack_u8(float v) {
    /* renderer_kernel.cu:237 uint8_t(v * 255): truncation; the CUDA float->u8
     * conversion saturates, restated here as a clamp to [0, 255]. */
    const float s = v * 255.f;
    if (!(s > 0.f)) return 0;
    if (s >= 255.f) return 255;
    return (uint8_t)s;
}

int orc_render_voxels(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                      int32_t x0, int32_t y0, int32_t w, int32_t h,
                      float *rgba, uint8_t *rgba8, float *split_track, float *sample_track,
                      int32_t *visited, int track_visit, int32_t *steps_out, orc_counters *ctr,
                      int n_threads) {
    return orc_render_voxels_ex(tree, cam, opt, x0, y0, w, h, NULL, NULL, rgba, rgba8, split_track, sample_track, visited, track_visit,
                                steps_out, ctr, n_threads);
}

int orc_render_voxels_ex(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                         int32_t x0, int32_t y0, int32_t w, int32_t h,
                         const float *tmax_px, const uint8_t *rgba8_init,
                         float *rgba, uint8_t *rgba8, float *split_track, float *sample_track,
                         int32_t *visited, int track_visit, int32_t *steps_out, orc_counters *ctr,
                         int n_threads) {
    if (!tree || !cam || !opt || w < 0 || h < 0) return -1;
    if (tree->N > 0 && (!tree->data || !tree->child)) return -1;
    uint64_t c_rays = 0, c_inb = 0, c_hit = 0, c_steps = 0, c_levels = 0, c_hits = 0, c_stop = 0, c_max = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (track_visit) n_threads = 1; /* serial visited marks, no races */
#else
    n_threads = 1;
#endif
    (void)n_threads;

    /* Work items are 16x16-pixel tiles handed out one at a time: rays differ in cost by two orders of magnitude (sky against a
       grazing ray through the shell), and rows of a 1080p frame are only 1080 / 4 = 270 items for 128 threads.  8160 tiles keep
       every core busy to the end; neighbouring rays share the sub-tree in a core's cache. */
    const int32_t tiles_x = (w + 15) / 16, tiles_y = (h + 15) / 16;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) \
    reduction(+ : c_rays, c_inb, c_hit, c_steps, c_levels, c_hits, c_stop) reduction(max : c_max)
    for (int32_t tile = 0; tile < tiles_x * tiles_y; ++tile) {
      const int32_t ty0 = (tile / tiles_x) * 16, tx0 = (tile % tiles_x) * 16;
      for (int32_t ty = ty0; ty < ty0 + 16 && ty < h; ++ty) {
        for (int32_t tx = tx0; tx < tx0 + 16 && tx < w; ++tx) {
            const int ix = x0 + tx, iy = y0 + ty;
            const int64_t p = (int64_t)ty * w + tx;
            float dir[3], cen[3], out[4] = {0.f, 0.f, 0.f, 0.f};
            float trk[6] = {-1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
            ray_stats st;
            memset(&st, 0, sizeof(st));
            if (tree->N > 0) { /* renderer_kernel.cu:266-269 */
                /* screen2worlddir :30-38 */
                const float xyz[3] = {(ix + 0.5f - cam->cx) / cam->fx, -(iy + 0.5f - cam->cy) / cam->fy, -1.0f};
                const float *m = cam->c2w;
                dir[0] = m[0] * xyz[0] + m[3] * xyz[1] + m[6] * xyz[2];
                dir[1] = m[1] * xyz[0] + m[4] * xyz[1] + m[7] * xyz[2];
                dir[2] = m[2] * xyz[0] + m[5] * xyz[1] + m[8] * xyz[2];
                const float invnorm = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                dir[0] *= invnorm;
                dir[1] *= invnorm;
                dir[2] *= invnorm;
                for (int i = 0; i < 3; ++i) cen[i] = tree->offset[i] + tree->scale[i] * m[9 + i]; /* :272-275 */
                float vdir[3] = {dir[0], dir[1], dir[2]};
                rodrigues(opt->rot_dirs, vdir); /* :282-283 */
                /* renderer_kernel.cu:277-280: t_max = 1e9f offscreen, else the pixel of the depth attachment */
                const float t_max = tmax_px ? tmax_px[p] : 1e9f;
                trace_ray(tree, visited, dir, vdir, cen, opt, t_max, out, &trk[1], &trk[2], &trk[0],
                          &trk[4], &trk[5], &trk[3], track_visit, &st);
            }
            /* composite_and_write :224-234 */
            const float nalpha = 1.f - out[3];
            if (rgba8_init) { /* offscreen == false, :230-234: over the pixel already in the image (uint8 / 255.f * nalpha) */
                const uint8_t init[3] = {rgba8_init[p * 4 + 0], rgba8_init[p * 4 + 1], rgba8_init[p * 4 + 2]}; /* (may alias rgba8) */
                out[0] += init[0] / 255.f * nalpha;
                out[1] += init[1] / 255.f * nalpha;
                out[2] += init[2] / 255.f * nalpha;
            } else { /* offscreen branch :225-229 */
                const float remain = opt->background_brightness * nalpha;
                out[0] += remain;
                out[1] += remain;
                out[2] += remain;
            }
            if (rgba) memcpy(rgba + p * 4, out, sizeof(out));
            if (rgba8) {
                rgba8[p * 4 + 0] = pack_u8(out[0]);
                rgba8[p * 4 + 1] = pack_u8(out[1]);
                rgba8[p * 4 + 2] = pack_u8(out[2]);
                rgba8[p * 4 + 3] = 255;
            }
            if (split_track) memcpy(split_track + p * 3, trk, 3 * sizeof(float));
            if (sample_track) memcpy(sample_track + p * 3, trk + 3, 3 * sizeof(float));
            if (steps_out) steps_out[p] = (int32_t)st.steps;
            c_rays += 1;
            c_inb += (uint64_t)st.in_bbox;
            c_hit += st.hits > 0;
            c_steps += st.steps;
            c_levels += st.levels;
            c_hits += st.hits;
            c_stop += (uint64_t)st.early_stop;
            if (st.steps > c_max) c_max = st.steps;
        }
      }
    }
    if (ctr) {
        ctr->rays += c_rays;
        ctr->rays_in_bbox += c_inb;
        ctr->rays_hit += c_hit;
        ctr->steps += c_steps;
        ctr->levels += c_levels;
        ctr->hits += c_hits;
        ctr->early_stops += c_stop;
        if (c_max > ctr->max_steps) ctr->max_steps = c_max;
    }
    return 0;
}

/* ------------------------------------------------------------ guided sampling (config 5) */

static void ray_gen(const orc_camera *cam, const orc_options *opt, int ix, int iy, float dir[3], float cen[3],
                    float vdir[3]) {
    /* screen2worlddir renderer_kernel.cu:30-38 + rodrigues :40-61 */
    const float xyz[3] = {(ix + 0.5f - cam->cx) / cam->fx, -(iy + 0.5f - cam->cy) / cam->fy, -1.0f};
    const float *m = cam->c2w;
    dir[0] = m[0] * xyz[0] + m[3] * xyz[1] + m[6] * xyz[2];
    dir[1] = m[1] * xyz[0] + m[4] * xyz[1] + m[7] * xyz[2];
    dir[2] = m[2] * xyz[0] + m[5] * xyz[1] + m[8] * xyz[2];
    const float invnorm = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= invnorm;
    dir[1] *= invnorm;
    dir[2] *= invnorm;
    for (int i = 0; i < 3; ++i) {
        cen[i] = m[9 + i];
        vdir[i] = dir[i];
    }
    rodrigues(opt->rot_dirs, vdir);
}

int orc_get_samples_from_voxels(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                                float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                int16_t *num_samples, float *samples, int32_t samples_dim,
                                int16_t *cluster_indices, const orc_cluster_grid *grid, int n_threads) {
    return orc_get_samples_from_voxels_ex(tree, cam, opt, NULL, split_track, sample_track, visited, track_visit, num_samples, samples, samples_dim,
                                          cluster_indices, grid, n_threads);
}

/* ... with the depth attachment of offscreen == false: t_max of every pixel (renderer_kernel.cu:354-357), NULL = 1e9f */
int orc_get_samples_from_voxels_ex(const orc_tree *tree, const orc_camera *cam, const orc_options *opt, const float *tmax_px,
                                   float *split_track, float *sample_track, int32_t *visited, int track_visit,
                                   int16_t *num_samples, float *samples, int32_t samples_dim,
                                   int16_t *cluster_indices, const orc_cluster_grid *grid, int n_threads) {
    if (!tree || !cam || !opt || !num_samples || !samples || !cluster_indices || !grid || tree->N <= 0) return -1;
    const int need = 4 + (opt->need_viewdir ? 3 : 0) + (opt->appearance_embedding != -1 ? 1 : 0);
    if (samples_dim < need) return -1;
    const int N = tree->N, N3 = N * N * N, data_dim = tree->data_dim;
    const int W = cam->width, H = cam->height, MG = opt->max_guided_samples;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (track_visit) n_threads = 1;
#else
    n_threads = 1;
#endif
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (int iy = 0; iy < H; ++iy) {
        for (int ix = 0; ix < W; ++ix) {
            const int64_t idx = (int64_t)iy * W + ix;
            float true_dir[3], true_cen[3], vdir[3];
            ray_gen(cam, opt, ix, iy, true_dir, true_cen, vdir);
            float trk[6] = {-1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
            trk[0] = (float)(opt->max_depth + 1);        /* rt_core.cuh:440 */
            trk[3] = (float)(opt->max_sample_count + 1); /* :441 */
            float cen[3];
            for (int i = 0; i < 3; ++i) cen[i] = tree->offset[i] + tree->scale[i] * true_cen[i]; /* :443-447 */
            float dir[3] = {true_dir[0], true_dir[1], true_dir[2]};
            dir[0] *= tree->scale[0];
            dir[1] *= tree->scale[1];
            dir[2] *= tree->scale[2];
            const float delta_scale = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] *= delta_scale;
            dir[1] *= delta_scale;
            dir[2] *= delta_scale;
            float tmax_bg = tmax_px ? tmax_px[idx] : 1e9f; /* renderer_kernel.cu:354-357 */
            tmax_bg /= delta_scale;
            float invdir[3];
            for (int i = 0; i < 3; ++i) invdir[i] = (float)(1.0 / ((double)dir[i] + 1e-9));
            float tmin = 0.0f, tmax = 1e4f;
            for (int i = 0; i < 3; ++i) {
                const float t1 = (float)(((double)opt->render_bbox[i] + 1e-6 - (double)cen[i]) * (double)invdir[i]);
                const float t2 = (float)(((double)opt->render_bbox[i + 3] - 1e-6 - (double)cen[i]) * (double)invdir[i]);
                tmin = fmaxf_(tmin, fminf_(t1, t2));
                tmax = fminf_(tmax, fmaxf_(t1, t2));
            }
            tmax = fminf_(tmax, tmax_bg);
            int16_t ns = num_samples[idx];
            if (!(tmax < 0 || tmin > tmax)) {
                float light_intensity = 1.f, t = tmin, max_weight = -1, max_sample_weight = -1;
                while (t < tmax) {
                    float pos[3];
                    for (int i = 0; i < 3; ++i) {
                        pos[i] = cen[i] + t * dir[i];
                        pos[i] = fmaxf_(fminf_(pos[i], 1.f - 1e-6f), 0.f);
                    }
                    int32_t chunk = 0, child_idx = 0;
                    int depth = 1;
                    for (;;) {
                        if (track_visit && visited && visited[chunk] == 0) visited[chunk] = 1;
                        int cur = 0;
                        for (int i = 0; i < 3; ++i) {
                            pos[i] *= (float)N;
                            const float f = floorf(pos[i]);
                            cur = (int)((float)(cur * N) + f);
                            pos[i] -= f;
                        }
                        const int32_t skip = tree->child[(int64_t)chunk * N3 + cur];
                        if (skip == 0) { child_idx = cur; break; }
                        depth += 1;
                        chunk += skip;
                    }
                    float cube_size = 1.f;
                    for (int i = 0; i < depth; ++i) cube_size *= (float)N;
                    float tu = 1e4f;
                    for (int i = 0; i < 3; ++i) {
                        const float t1 = -pos[i] * invdir[i];
                        const float t2 = t1 + invdir[i];
                        tu = fminf_(tu, fmaxf_(t1, t2));
                    }
                    const float delta_t = tu / cube_size + opt->step_size;
                    const float sigma = orc_half_to_float(tree->data[((int64_t)chunk * N3 + child_idx) * data_dim + data_dim - 1]);
                    const int16_t sc = tree->sample_counts ? tree->sample_counts[(int64_t)chunk * N3 + child_idx] : 0;
                    if (sigma > opt->sigma_thresh) {
                        const float att = orc_expf(-delta_t * delta_scale * sigma);
                        const float weight = light_intensity * (1.f - att);
                        if (weight > max_weight && depth < opt->max_depth) {
                            trk[1] = (float)chunk; trk[2] = (float)child_idx; trk[0] = (float)depth;
                            max_weight = weight;
                        }
                        if (tree->sample_counts && weight > max_sample_weight && sc < opt->max_sample_count) {
                            trk[4] = (float)chunk; trk[5] = (float)child_idx; trk[3] = (float)sc;
                            max_sample_weight = weight;
                        }
                        if (ns < MG) { /* rt_core.cuh:508-549 */
                            float *row = samples + (idx * MG + ns) * samples_dim;
                            float tz[3];
                            for (int i = 0; i < 3; ++i) tz[i] = t * dir[i] / tree->scale[i];
                            row[0] = sqrtf(tz[0] * tz[0] + tz[1] * tz[1] + tz[2] * tz[2]);
                            row[1] = true_cen[0] + true_dir[0] * row[0];
                            row[2] = true_cen[1] + true_dir[1] * row[0];
                            row[3] = true_cen[2] + true_dir[2] * row[0];
                            if (opt->need_viewdir) {
                                row[4] = vdir[0]; row[5] = vdir[1]; row[6] = vdir[2];
                                if (opt->appearance_embedding != -1) row[7] = (float)opt->appearance_embedding;
                            } else if (opt->appearance_embedding != -1) {
                                row[4] = (float)opt->appearance_embedding;
                            }
                            const int g1 = (int)fmaxf(fminf((row[2] - grid->min_position[1]) / grid->range[1] * (float)grid->grid_dim[0],
                                                            (float)grid->grid_dim[0] - 1.0f), 0.0f);
                            const int g2 = (int)fmaxf(fminf((row[3] - grid->min_position[2]) / grid->range[2] * (float)grid->grid_dim[1],
                                                            (float)grid->grid_dim[1] - 1.0f), 0.0f);
                            cluster_indices[idx * MG + ns] = (int16_t)(g1 * grid->grid_dim[1] + g2);
                            ns += 1;
                        }
                        light_intensity *= att;
                        if (light_intensity < opt->stop_thresh) break;
                    } else {
                        if (max_weight == -1 && depth < opt->max_depth) {
                            trk[1] = (float)chunk; trk[2] = (float)child_idx; trk[0] = (float)depth;
                        }
                        if (tree->sample_counts && max_sample_weight == -1 && sc < opt->max_sample_count) {
                            trk[4] = (float)chunk; trk[5] = (float)child_idx; trk[3] = (float)sc;
                        }
                    }
                    t += delta_t;
                }
            }
            num_samples[idx] = ns;
            if (split_track) memcpy(split_track + idx * 3, trk, 3 * sizeof(float));
            if (sample_track) memcpy(sample_track + idx * 3, trk + 3, 3 * sizeof(float));
        }
    }
    return 0;
}

int orc_render_nerf_results(const orc_tree *tree, const orc_camera *cam, const orc_options *opt,
                            const float *sample_values, int32_t value_stride, const float *z_vals,
                            const int64_t *offsets, float *rgba, uint8_t *rgba8, int n_threads) {
    if (!tree || !cam || !opt || !offsets || value_stride < 4) return -1;
    const int W = cam->width, H = cam->height, basis_dim = tree->basis_dim;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (int iy = 0; iy < H; ++iy) {
        for (int ix = 0; ix < W; ++ix) {
            const int64_t idx = (int64_t)iy * W + ix;
            float dir[3], cen[3], vdir[3], out[4] = {0.f, 0.f, 0.f, 1.0f}; /* renderer_kernel.cu:315-316 */
            ray_gen(cam, opt, ix, iy, dir, cen, vdir);
            const int64_t start = idx == 0 ? 0 : offsets[idx - 1], end = offsets[idx];
            if (start != end) { /* rt_core.cuh:344-346 */
                float basis_fn[ORC_BASIS_MAX];
                if (tree->format == 1) orc_sh_basis(basis_dim, vdir, basis_fn);
                else for (int i = 0; i < ORC_BASIS_MAX; ++i) basis_fn[i] = 0.f;
                for (int i = 0; i < opt->basis_minmax[0] && i < ORC_BASIS_MAX; ++i) basis_fn[i] = 0.f;
                for (int i = opt->basis_minmax[1] + 1; i < ORC_BASIS_MAX; ++i) if (i >= 0) basis_fn[i] = 0.f;
                float ti = 1, weight_component = 0.f, weight;
                for (int64_t i = start; i < end; i++) {
                    const float *sv = sample_values + i * value_stride;
                    if (i < end - 1) {
                        const float delta_i = z_vals[i + 1] - z_vals[i];
                        weight_component = orc_expf(-sv[3] * delta_i);
                        weight = ti * (1.0f - weight_component);
                    } else {
                        weight = ti;
                    }
                    if (opt->render_depth) {
                        out[0] += weight * ti; /* sic, rt_core.cuh:372 */
                    } else if (basis_dim >= 0) {
                        int off = 0;
#define MB(k) (basis_fn[k] * sv[off + (k)])
                        for (int c = 0; c < 3; ++c) {
                            float tmp = basis_fn[0] * sv[off];
                            switch (basis_dim) {
                                case 25: tmp += MB(16) + MB(17) + MB(18) + MB(19) + MB(20) + MB(21) + MB(22) + MB(23) + MB(24); /* fallthrough */
                                case 16: tmp += MB(9) + MB(10) + MB(11) + MB(12) + MB(13) + MB(14) + MB(15); /* fallthrough */
                                case 9: tmp += MB(4) + MB(5) + MB(6) + MB(7) + MB(8); /* fallthrough */
                                case 4: tmp += MB(1) + MB(2) + MB(3);
                            }
                            out[c] += weight / (1.f + orc_expf(-tmp));
                            off += basis_dim;
                        }
#undef MB
                    } else {
                        for (int j = 0; j < 3; ++j) out[j] += weight * sv[j];
                    }
                    ti *= weight_component;
                }
                if (opt->render_depth) out[0] = out[1] = out[2] = fminf_(out[0] * 0.3f, 1.0f);
            }
            const float nalpha = 1.f - out[3];
            const float remain = opt->background_brightness * nalpha;
            out[0] += remain;
            out[1] += remain;
            out[2] += remain;
            if (rgba) memcpy(rgba + idx * 4, out, sizeof(out));
            if (rgba8) {
                rgba8[idx * 4 + 0] = pack_u8(out[0]);
                rgba8[idx * 4 + 1] = pack_u8(out[1]);
                rgba8[idx * 4 + 2] = pack_u8(out[2]);
                rgba8[idx * 4 + 3] = 255;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------ refinement kernels (config 5) */

/* generate_samples_inner, renderer_kernel.cu:88-168 (N == 2) */
static void gen_samples_inner(const int32_t *parent, const float offset[3], const float scale[3], const orc_options *opt,
                              float *samples, int32_t dim, int16_t *clusters, const orc_cluster_grid *grid, int64_t idx,
                              int32_t abs_chunk, int32_t child_idx) {
    int32_t curr[4];
    curr[0] = abs_chunk * 8 + child_idx;
    uint8_t depth = 0;
    float corners[3] = {0, 0, 0};
    for (;;) {
        for (int i = 3; i > 0; --i) {
            curr[i] = curr[0] % 2;
            curr[0] /= 2;
        }
        for (int i = 0; i < 3; ++i) {
            corners[i] += (float)curr[i + 1];
            corners[i] /= 2.f;
        }
        if (curr[0] == 0) break;
        curr[0] = parent[curr[0]];
        depth += 1;
    }
    float length_local = 1.f;
    for (int i = 0; i < depth + 1; ++i) length_local /= 2.f; /* pow(N, -depth - 1) */
    const int spc = opt->samples_per_corner;
    float *row0 = samples + idx * spc * dim;
    for (int i = 0; i < 3; ++i) {
        corners[i] -= offset[i];
        corners[i] /= scale[i];
        for (int j = 0; j < spc; j++) {
            row0[j * dim + i] *= (length_local / scale[i]);
            row0[j * dim + i] += corners[i];
        }
    }
    if (opt->need_viewdir) {
        for (int j = 0; j < spc; j++) {
            row0[j * dim + 3] = 1;
            row0[j * dim + 4] = 0;
            row0[j * dim + 5] = 0;
            if (opt->appearance_embedding != -1) row0[j * dim + 6] = (float)opt->appearance_embedding;
        }
    } else if (opt->appearance_embedding != -1) {
        for (int j = 0; j < spc; j++) row0[j * dim + 3] = (float)opt->appearance_embedding;
    }
    for (int j = 0; j < spc; j++) {
        const int g1 = (int)fmaxf(fminf((row0[j * dim + 1] - grid->min_position[1]) / grid->range[1] * (float)grid->grid_dim[0],
                                        (float)grid->grid_dim[0] - 1.0f), 0.0f);
        const int g2 = (int)fmaxf(fminf((row0[j * dim + 2] - grid->min_position[2]) / grid->range[2] * (float)grid->grid_dim[1],
                                        (float)grid->grid_dim[1] - 1.0f), 0.0f);
        clusters[idx * spc + j] = (int16_t)(g1 * grid->grid_dim[1] + g2);
    }
}

int orc_add_children_and_generate_samples(int32_t *child, int32_t *parent, const float offset[3], const float scale[3],
                                          int32_t capacity, const orc_options *opt, const int32_t *parent_nodes,
                                          int32_t num_parents, float *samples, int32_t samples_dim,
                                          int16_t *cluster_indices, int32_t *visited, const orc_cluster_grid *grid) {
    if (!child || !parent || !opt || !parent_nodes || !samples || !cluster_indices || !visited || !grid) return -1;
    /* renderer_kernel.cu:170-198, one "thread" per (new chunk, child); linking first, as thread child_idx == 0 does */
    for (int32_t rel = 0; rel < num_parents; ++rel) {
        const int32_t abs_chunk = capacity + rel, pc = parent_nodes[rel * 2], pj = parent_nodes[rel * 2 + 1];
        child[(int64_t)pc * 8 + pj] = abs_chunk - pc;
        parent[abs_chunk] = pc * 8 + pj;
        visited[abs_chunk] = visited[pc];
    }
    for (int64_t tid = 0; tid < (int64_t)num_parents * 8; ++tid) {
        const int32_t abs_chunk = capacity + (int32_t)(tid / 8), child_idx = (int32_t)(tid % 8);
        child[(int64_t)abs_chunk * 8 + child_idx] = 0;
        gen_samples_inner(parent, offset, scale, opt, samples, samples_dim, cluster_indices, grid, tid, abs_chunk, child_idx);
    }
    return 0;
}

int orc_generate_samples(const int32_t *parent, const float offset[3], const float scale[3], const orc_options *opt,
                         const int32_t *nodes, int32_t num_items, float *samples, int32_t samples_dim,
                         int16_t *cluster_indices, const orc_cluster_grid *grid) {
    if (!parent || !opt || !nodes || !samples || !cluster_indices || !grid) return -1;
    for (int64_t tid = 0; tid < num_items; ++tid) /* renderer_kernel.cu:200-213 */
        gen_samples_inner(parent, offset, scale, opt, samples, samples_dim, cluster_indices, grid, tid, nodes[tid * 2], nodes[tid * 2 + 1]);
    return 0;
}

int orc_adjust_parents_and_children(int32_t *child, int32_t *parent, int32_t capacity, int32_t first_shift_index,
                                    const uint8_t *to_delete, const int32_t *index_shifts) {
    if (!child || !parent || !to_delete || !index_shifts || first_shift_index < 0) return -1;
    for (int32_t chunk = first_shift_index; chunk < capacity; ++chunk) { /* renderer_kernel.cu:63-86 */
        /* chunk 0 (the root, parent word 0 in svox files): the reference thread adds shift[0] - shift[0] = 0 to
         * child[0][0] -- no effect, so it is skipped (cuda_renderer.cpp:350 does call with first_shift_index 0) */
        if (chunk == 0) continue;
        const int32_t pc = parent[chunk] / 8, pj = parent[chunk] % 8;
        if (to_delete[chunk]) {
            child[(int64_t)pc * 8 + pj] = 0;
        } else {
            const int32_t parent_shift = index_shifts[pc], child_shift = index_shifts[chunk];
            child[(int64_t)pc * 8 + pj] += (parent_shift - child_shift);
            parent[chunk] -= (index_shifts[pc] * 8);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ the build's own sub-module MLP (unpinned) */

static float mlp_tri(float t) {
    const float r = t - floorf(t + 0.5f);
    return 4.f * fabsf(r) - 1.f;
}

static void mlp_encode_block(const float v[3], int octaves, float *out) {
    for (int i = 0; i < 3; ++i) out[i] = v[i];
    for (int k = 0; k < octaves; ++k) {
        const float scale = (float)(1u << k);
        for (int i = 0; i < 3; ++i) {
            out[3 + 6 * k + i] = mlp_tri(v[i] * scale);
            out[3 + 6 * k + 3 + i] = mlp_tri(v[i] * scale + 0.25f);
        }
    }
}

int orc_mlp_forward(const orc_mlp_desc *D, const uint16_t *params, const int16_t *cluster_indices, const float *samples,
                    int32_t samples_stride, int64_t n, float *results, int32_t result_stride) {
    if (!D || !params || !cluster_indices || !samples || !results || D->hidden_width > 256 || D->hidden_layers < 1) return -1;
    const int n_pos = 3 + 6 * D->pos_octaves, n_dir = D->need_viewdir ? 3 + 6 * D->dir_octaves : 0;
    const int emb_dim = D->n_embeddings > 0 ? D->embedding_dim : 0;
    const int in_dim = n_pos + n_dir + emb_dim, W = D->hidden_width;
    const size_t per_cluster = (size_t)W * in_dim + W + (size_t)(D->hidden_layers - 1) * ((size_t)W * W + W) + (size_t)D->out_dim * W +
                               D->out_dim + (size_t)D->n_embeddings * emb_dim;
    if (in_dim > 512) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < n; ++row) {
        float *out = results + row * result_stride;
        const int c = cluster_indices[row];
        if (c < 0 || c >= D->n_clusters) {
            for (int k = 0; k < D->out_dim; ++k) out[k] = 0.f;
            continue;
        }
        const float *x = samples + row * samples_stride;
        const uint16_t *P = params + (size_t)c * per_cluster;
        float enc[512], h[256], h2[256];
        float p[3];
        for (int i = 0; i < 3; ++i) p[i] = (x[i] - D->center[i]) * D->inv_extent[i];
        mlp_encode_block(p, D->pos_octaves, enc);
        if (D->need_viewdir) mlp_encode_block(x + 3, D->dir_octaves, enc + n_pos);
        if (emb_dim > 0) {
            int idx = (int)x[D->need_viewdir ? 6 : 3];
            idx = idx < 0 ? 0 : (idx >= D->n_embeddings ? D->n_embeddings - 1 : idx);
            const uint16_t *table = P + per_cluster - (size_t)D->n_embeddings * emb_dim;
            for (int k = 0; k < emb_dim; ++k) enc[n_pos + n_dir + k] = orc_half_to_float(table[(size_t)idx * emb_dim + k]);
        }
        for (int k = 0; k < in_dim; ++k) enc[k] = orc_half_to_float(orc_float_to_half(enc[k])); /* activations are binary16 */
        const float *cur = enc;
        int cur_dim = in_dim;
        const uint16_t *w = P;
        for (int layer = 0; layer <= D->hidden_layers; ++layer) {
            const int od = layer == D->hidden_layers ? D->out_dim : W;
            float *dst = layer == D->hidden_layers ? out : (cur == h ? h2 : h);
            const uint16_t *bias = w + (size_t)od * cur_dim;
            for (int o = 0; o < od; ++o) {
                float acc = orc_half_to_float(bias[o]);
                for (int k = 0; k < cur_dim; ++k) acc += orc_half_to_float(w[(size_t)o * cur_dim + k]) * cur[k];
                dst[o] = layer == D->hidden_layers ? acc : orc_half_to_float(orc_float_to_half(acc > 0.f ? acc : 0.f));
            }
            w = bias + od;
            cur = dst;
            cur_dim = od;
        }
    }
    return 0;
}
