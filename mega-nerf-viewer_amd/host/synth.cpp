// synth.cpp -- deterministic synthetic N3Trees for tests, goldens and the benchmark
// (SURVEY.md 8(d)).  Integer-hash PRNG (splitmix64) and IEEE-only float arithmetic:
// the same parameters give bit-identical trees on every host, so the container that
// writes the goldens and the GPU box that checks them agree without shipping the data.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../csrc/mnv_knobs.h"
#include "n3tree.hpp"

namespace viewer::synth {

namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ (a * 0xD6E8FEB86659FD93ull)) + b);
}
inline float uniform01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }  // exact
// Irwin-Hall(4) stand-in for a unit normal: exact integer sum, two exact float ops.
inline float approx_normal(uint64_t h) {
    const uint32_t s = (uint32_t)(h & 0xffff) + (uint32_t)((h >> 16) & 0xffff) + (uint32_t)((h >> 32) & 0xffff) + (uint32_t)(h >> 48);
    return ((float)s * (1.0f / 65536.0f) - 2.0f) * 1.7320508f;
}

uint16_t float_to_half(float f) {  // round-to-nearest-even
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) return sign;
    const int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = 13;
    uint32_t base = 0;
    if (e < -14) shift = 13 + (-14 - e);
    else { base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | (base + q));
}

template <typename F>
void parallel_for(size_t n, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = t * per, e = std::min(n, b + per);
        if (b >= e) break;
        th.emplace_back([=] { for (size_t i = b; i < e; ++i) f(i); });
    }
    for (auto &x : th) x.join();
}

// Fill colour coefficients of one dense voxel row.
void fill_row(uint16_t *row, int format, int basis_dim, float coef_sd, uint64_t seed, uint64_t vox) {
    if (format == MNV_FORMAT_SH && basis_dim >= 0) {
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < basis_dim; ++k) {
                int l = 0;
                while ((l + 1) * (l + 1) <= k) ++l;  // SH degree of coefficient k
                float sd = coef_sd;
                for (int i = 0; i < l; ++i) sd *= 0.5f;
                row[c * basis_dim + k] = float_to_half(approx_normal(hash3(seed, vox, 16 + (uint64_t)(c * basis_dim + k))) * sd);
            }
        }
    } else {
        for (int c = 0; c < 3; ++c) row[c] = float_to_half(uniform01(hash3(seed, vox, 16 + (uint64_t)c)));
    }
}

struct Builder {
    std::vector<int32_t> child;
    std::vector<int32_t> parent;
    std::vector<uint32_t> cx, cy, cz;  // integer cell origin of the chunk at its own level
    std::vector<uint8_t> level;        // cells of this chunk have depth `level` (root chunk: 1)
    size_t add_chunk(int32_t par, uint32_t x, uint32_t y, uint32_t z, int lvl) {
        child.insert(child.end(), 8, 0);
        parent.push_back(par);
        cx.push_back(x);
        cy.push_back(y);
        cz.push_back(z);
        level.push_back((uint8_t)lvl);
        return parent.size() - 1;
    }
};

void finish(N3Tree &t, Builder &b, int format, int basis_dim, const float offset[3], const float scale[3],
            std::vector<uint16_t> &data, int data_dim) {
    mnv_tree_view v;
    std::memset(&v, 0, sizeof(v));
    v.N = 2;
    v.data_dim = data_dim;
    v.format = format;
    v.basis_dim = basis_dim;
    v.capacity = (int32_t)b.parent.size();
    for (int i = 0; i < 3; ++i) { v.offset[i] = offset[i]; v.scale[i] = scale[i]; }
    t.adopt(v, std::move(data), std::move(b.child), std::move(b.parent));
}

}  // namespace

void random_tree(const mnv_synth_random_params &p, N3Tree &out) {
    if (p.depth < 1 || p.depth > 20) throw std::runtime_error("synth: depth out of range");
    const bool sh = p.format == MNV_FORMAT_SH && p.basis_dim >= 0;
    const int data_dim = sh ? 3 * p.basis_dim + 1 : 4;
    Builder b;
    b.add_chunk(-1, 0, 0, 0, 1);
    for (size_t c = 0; c < b.parent.size(); ++c) {  // BFS: chunk ids are level-ordered
        const int lvl = b.level[c];
        for (int j = 0; j < 8; ++j) {
            const uint64_t vox = (uint64_t)c * 8 + j;
            if (lvl < p.depth && uniform01(hash3(p.seed, vox, 0)) < p.refine_prob) {
                const uint32_t x = b.cx[c] * 2 + ((j >> 2) & 1), y = b.cy[c] * 2 + ((j >> 1) & 1), z = b.cz[c] * 2 + (j & 1);
                const size_t n = b.add_chunk((int32_t)vox, x, y, z, lvl + 1);
                b.child[vox] = (int32_t)(n - c);
            }
        }
    }
    const size_t nvox = b.child.size();
    std::vector<uint16_t> data(nvox * data_dim, 0);
    parallel_for(nvox, [&](size_t vox) {
        if (b.child[vox] != 0) return;
        uint16_t *row = data.data() + vox * data_dim;
        const bool empty = uniform01(hash3(p.seed, vox, 1)) < p.empty_prob;
        const float sigma = empty ? 0.f : uniform01(hash3(p.seed, vox, 2)) * p.sigma_max;
        row[data_dim - 1] = float_to_half(sigma);
        fill_row(row, p.format, p.basis_dim, p.coef_sd, p.seed, vox);
    });
    finish(out, b, p.format, sh ? p.basis_dim : -1, p.offset, p.scale, data, data_dim);
}

namespace {
struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!mnv::knob_set(mnv::KNOB_SYNTH_TIMING)) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[synth] %s: %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace

void shell_tree(const mnv_synth_shell_params &p, N3Tree &out) {
    Timer tm;
    if (p.depth < 1 || p.depth > 14) throw std::runtime_error("synth: depth out of range");
    if (p.basis_dim < 1) throw std::runtime_error("synth: shell tree needs an SH basis");
    const int data_dim = 3 * p.basis_dim + 1;
    const double r_lo = (double)p.radius - (double)p.half_thickness, r_hi = (double)p.radius + (double)p.half_thickness;
    const double lo2 = r_lo > 0 ? r_lo * r_lo : 0.0, hi2 = r_hi * r_hi;
    // does the cell [x, x+s]^3 (unit-cube coords) overlap the shell lo <= |q - c| <= hi ?
    auto overlaps = [&](uint32_t ix, uint32_t iy, uint32_t iz, int depth) {
        const double s = std::ldexp(1.0, -depth);
        const uint32_t ii[3] = {ix, iy, iz};
        double dmin2 = 0.0, dmax2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double x0 = ii[a] * s - 0.5, x1 = x0 + s;
            const double near = x0 > 0 ? x0 : (x1 < 0 ? -x1 : 0.0);
            const double far = std::fabs(x0) > std::fabs(x1) ? std::fabs(x0) : std::fabs(x1);
            dmin2 += near * near;
            dmax2 += far * far;
        }
        return dmin2 <= hi2 && dmax2 >= lo2;
    };
    Builder b;
    b.add_chunk(-1, 0, 0, 0, 1);
    std::vector<uint8_t> dense;  // per voxel
    for (size_t c = 0; c < b.parent.size(); ++c) {
        const int lvl = b.level[c];
        dense.resize((c + 1) * 8, 0);
        for (int j = 0; j < 8; ++j) {
            const uint64_t vox = (uint64_t)c * 8 + j;
            const uint32_t x = b.cx[c] * 2 + ((j >> 2) & 1), y = b.cy[c] * 2 + ((j >> 1) & 1), z = b.cz[c] * 2 + (j & 1);
            if (!overlaps(x, y, z, lvl)) continue;
            if (lvl < p.depth) {
                const size_t n = b.add_chunk((int32_t)vox, x, y, z, lvl + 1);
                b.child[vox] = (int32_t)(n - c);
            } else {
                dense[vox] = 1;
            }
        }
    }
    tm.lap("structure");
    const size_t nvox = b.child.size();
    dense.resize(nvox, 0);
    std::vector<uint16_t> data(nvox * data_dim, 0);
    tm.lap("alloc");
    parallel_for(nvox, [&](size_t vox) {
        if (!dense[vox]) return;
        uint16_t *row = data.data() + vox * data_dim;
        const float sigma = p.sigma_lo + uniform01(hash3(p.seed, vox, 2)) * (p.sigma_hi - p.sigma_lo);
        row[data_dim - 1] = float_to_half(sigma);
        fill_row(row, MNV_FORMAT_SH, p.basis_dim, 1.0f, p.seed, vox);
    });
    tm.lap("fill");
    finish(out, b, MNV_FORMAT_SH, p.basis_dim, p.offset, p.scale, data, data_dim);
    tm.lap("assign");
}

// "Merged Mega-NeRF" stand-in: an anisotropic volume (tree-x is height) holding a
// terrain-like occupied layer x = h(y, z).  The y-z plane is cut into bricks_y x bricks_z "sub-modules",
// each with its own noise seed, so the surface is discontinuous at brick borders like independently
// trained sub-modules.  h is bilinear value noise on a hashed lattice: float mul/add only.
void terrain_tree(const mnv_synth_terrain_params &p, N3Tree &out) {
    if (p.depth < 1 || p.depth > 14) throw std::runtime_error("synth: depth out of range");
    if (p.basis_dim < 1) throw std::runtime_error("synth: terrain tree needs an SH basis");
    Timer tm;
    const int data_dim = 3 * p.basis_dim + 1;
    const int by = p.bricks_y > 0 ? p.bricks_y : 1, bz = p.bricks_z > 0 ? p.bricks_z : 1;
    const double cells = p.noise_cells > 0 ? p.noise_cells : 8;
    auto lattice = [&](int brick, int iy, int iz) {
        return (double)uniform01(hash3(p.seed ^ 0x7e44a1full, (uint64_t)brick, ((uint64_t)(uint32_t)iy << 32) | (uint32_t)iz));
    };
    // height in [base, base + amplitude] at unit-cube (y, z)
    auto height = [&](double y, double z) {
        int jy = (int)(y * by), jz = (int)(z * bz);
        jy = jy >= by ? by - 1 : jy;
        jz = jz >= bz ? bz - 1 : jz;
        const int brick = jy * bz + jz;
        const double u = y * cells, v = z * cells;
        const int iy = (int)u, iz = (int)v;
        const double fu = u - iy, fv = v - iz;
        const double a = lattice(brick, iy, iz), b = lattice(brick, iy + 1, iz), c = lattice(brick, iy, iz + 1), d = lattice(brick, iy + 1, iz + 1);
        const double n = (a * (1 - fu) + b * fu) * (1 - fv) + (c * (1 - fu) + d * fu) * fv;
        return (double)p.base + (double)p.amplitude * n;
    };
    // a cell overlaps the layer if its x range meets [hmin - thick - slack, hmax + thick + slack], with h sampled on
    // a 3x3 grid of the cell's y-z footprint and a Lipschitz slack for what lies between the samples
    const double lip = (double)p.amplitude * cells * 2.0;
    auto overlaps = [&](uint32_t ix, uint32_t iy, uint32_t iz, int depth) {
        const double s = std::ldexp(1.0, -depth);
        const double x0 = ix * s, y0 = iy * s, z0 = iz * s;
        double hmin = 1e30, hmax = -1e30;
        for (int a = 0; a <= 2; ++a)
            for (int b = 0; b <= 2; ++b) {
                const double yy = y0 + 0.5 * a * s, zz = z0 + 0.5 * b * s;
                const double h = height(yy < 1.0 ? yy : 0.999999, zz < 1.0 ? zz : 0.999999);
                hmin = h < hmin ? h : hmin;
                hmax = h > hmax ? h : hmax;
            }
        const double slack = lip * s * 0.5;
        return x0 <= hmax + p.thickness + slack && x0 + s >= hmin - p.thickness - slack;
    };
    Builder b;
    b.add_chunk(-1, 0, 0, 0, 1);
    std::vector<uint8_t> dense, hit;
    // Chunks are numbered in breadth-first order, so a level is a contiguous range: the overlap tests of a level (nine height samples
    // per voxel, the whole cost of a 10 M-chunk tree) run on all cores, the children are then appended in the serial order -- the
    // tree is the one the chunk-by-chunk loop builds.
    for (size_t lo = 0; lo < b.parent.size();) {
        const size_t hi = b.parent.size();
        hit.assign((hi - lo) * 8, 0);
        parallel_for((hi - lo) * 8, [&](size_t i) {
            const size_t c = lo + i / 8;
            const int j = (int)(i & 7);
            hit[i] = overlaps(b.cx[c] * 2 + ((j >> 2) & 1), b.cy[c] * 2 + ((j >> 1) & 1), b.cz[c] * 2 + (j & 1), b.level[c]) ? 1 : 0;
        });
        dense.resize(hi * 8, 0);
        for (size_t c = lo; c < hi; ++c) {
            const int lvl = b.level[c];
            for (int j = 0; j < 8; ++j) {
                if (!hit[(c - lo) * 8 + j]) continue;
                const uint64_t vox = (uint64_t)c * 8 + j;
                if (lvl < p.depth) {
                    const size_t n = b.add_chunk((int32_t)vox, b.cx[c] * 2 + ((j >> 2) & 1), b.cy[c] * 2 + ((j >> 1) & 1), b.cz[c] * 2 + (j & 1), lvl + 1);
                    b.child[vox] = (int32_t)(n - c);
                } else {
                    dense[vox] = 1;
                }
            }
        }
        lo = hi;
    }
    tm.lap("structure");
    const size_t nvox = b.child.size();
    dense.resize(nvox, 0);
    std::vector<uint16_t> data(nvox * data_dim, 0);
    parallel_for(nvox, [&](size_t vox) {
        if (!dense[vox]) return;
        uint16_t *row = data.data() + vox * data_dim;
        const float sigma = p.sigma_lo + uniform01(hash3(p.seed, vox, 2)) * (p.sigma_hi - p.sigma_lo);
        row[data_dim - 1] = float_to_half(sigma);
        fill_row(row, MNV_FORMAT_SH, p.basis_dim, 1.0f, p.seed, vox);
    });
    tm.lap("fill");
    finish(out, b, MNV_FORMAT_SH, p.basis_dim, p.offset, p.scale, data, data_dim);
}

}  // namespace viewer::synth
