// mnv_march_accel.hip -- the tuned N3Tree march for gfx950 (MI355X) on the packed layout.
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
//   * one 32-bit node word per voxel carries either the child link or the leaf's sigma,
//     so the dependent sigma load of the reference disappears; colour rows are fetched
//     (16-B vector loads from a 64-B padded row) only for dense samples;
//   * rays are queued in 8x8-pixel tile order and the tile range is split into 8
//     contiguous bands, one per XCD (workgroup b runs on XCD b % 8), so that each XCD's
//     4 MiB L2 holds the sub-trees of its own screen region; empty queues steal.
// MFMA is not used: the inner step is pointer chasing plus a <= 75-term dot in a fixed
// summation order (DESIGN.md "Why no MFMA").
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "mnv_accel.h"
#include "mnv_internal.h"

#pragma clang fp contract(off)

namespace mnv {

// ------------------------------------------------------------------ accel build kernels

// Level-synchronous depth propagation: depth[child chunk] = depth[chunk] + 1.
__global__ void accel_depth_pass(const int32_t *child, int32_t *depth, int32_t capacity, int32_t level,
                                 int32_t *changed) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (int64_t)capacity * 8) return;
    const int32_t c = (int32_t)(v >> 3);
    if (depth[c] != level) return;
    const int32_t skip = child[v];
    if (skip != 0) {
        const int64_t t = (int64_t)c + skip;
        if (t >= 0 && t < capacity) {
            depth[t] = level + 1;
            *changed = 1;
        }
    }
}

__global__ void accel_pack_nodes(const int32_t *child, const uint16_t *data, const int32_t *depth,
                                 uint32_t *nodes, int32_t capacity, int32_t data_dim) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (int64_t)capacity * 8) return;
    const int32_t c = (int32_t)(v >> 3);
    const int32_t skip = child[v];
    if (skip != 0) {
        nodes[v] = (uint32_t)(c + skip);
    } else {
        const uint32_t d = (uint32_t)depth[c] & 0x7fu;
        nodes[v] = kLeafBit | (d << 16) | (uint32_t)data[v * data_dim + data_dim - 1];
    }
}

// rows[v] = 3 channel blocks of chan_halfs binary16 each (the basis_dim coefficients of the channel,
// zero padded to a multiple of 8 B); one thread per (voxel, channel)
__global__ void accel_pack_rows(const uint16_t *data, uint16_t *rows, int64_t nvox, int32_t data_dim,
                                int32_t per_chan, int32_t chan_halfs, int32_t row_halfs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox * 3) return;
    const int64_t v = i / 3;
    const int32_t c = (int32_t)(i % 3);
    const uint16_t *src = data + v * data_dim + c * per_chan;
    uint16_t *dst = rows + v * row_halfs + c * chan_halfs;
    for (int32_t k = 0; k < chan_halfs; ++k) dst[k] = k < per_chan ? src[k] : (uint16_t)0;
    if (c == 2)
        for (int32_t k = 3 * chan_halfs; k < row_halfs; ++k) rows[v * row_halfs + k] = 0;
}

// grid[(ix*G + iy)*G + iz] = word of the voxel of depth <= L that covers cell (ix,iy,iz)
__global__ void accel_build_grid(const uint32_t *nodes, uint32_t *grid, uint32_t *grid_vox, int32_t L) {
    const int32_t G = 1 << L;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * G * G) return;
    const int32_t iz = i & (G - 1), iy = (i >> L) & (G - 1), ix = i >> (2 * L);
    uint32_t chunk = 0, word = 0, vox = 0;
    for (int32_t l = 1; l <= L; ++l) {
        const int32_t s = L - l;
        const int32_t cidx = (((ix >> s) & 1) << 2) | (((iy >> s) & 1) << 1) | ((iz >> s) & 1);
        vox = chunk * 8u + (uint32_t)cidx;
        word = nodes[vox];
        if (word & kLeafBit) break;
        chunk = word;
    }
    grid[i] = word;
    grid_vox[i] = vox;  // voxel index of the covering leaf (meaningful when `word` is a leaf)
}

// Brick-ordered index of cell (cx,cy,cz) of the level-L2 grid: 4x4x4-cell bricks (256 B) in
// row-major brick order, so that the cells neighbouring rays touch share cache lines.
__host__ __device__ __forceinline__ uint32_t grid2_index(uint32_t cx, uint32_t cy, uint32_t cz, int L2) {
    const int LB = L2 - 2;
    const uint32_t brick = (((((cx >> 2) << LB) + (cy >> 2))) << LB) + (cz >> 2);
    return (brick << 6) | ((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u);
}

// grid2[grid2_index(c)] = word of the voxel of depth <= L2 covering cell c, grid2_vox = its voxel index
__global__ void accel_build_grid2(const uint32_t *nodes, uint32_t *grid2, uint32_t *grid2_vox, int32_t L2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // natural (x-major) cell number
    const uint32_t G = 1u << L2;
    if (i >= (uint64_t)G * G * G) return;
    const uint32_t iz = i & (G - 1), iy = (i >> L2) & (G - 1), ix = i >> (2 * L2);
    uint32_t chunk = 0, word = 0, vox = 0;
    for (int32_t l = 1; l <= L2; ++l) {
        const int32_t s = L2 - l;
        const uint32_t cidx = (((ix >> s) & 1u) << 2) | (((iy >> s) & 1u) << 1) | ((iz >> s) & 1u);
        vox = chunk * 8u + cidx;
        word = nodes[vox];
        if (word & kLeafBit) break;
        chunk = word;
    }
    const uint32_t o = grid2_index(ix, iy, iz, L2);
    grid2[o] = word;
    grid2_vox[o] = vox;
}

// ---- incremental update after a refinement step (mnv_accel_refresh)

// depth of appended chunks from their parent words (parent[c] = parent_chunk * 8 + slot): a new chunk may hang under another new chunk,
// so every thread walks up until it meets a chunk that existed before (its depth is known) and adds the hops.  One launch, no
// iteration on the host.  flags[1] = deepest depth seen.
__global__ void accel_refresh_depth(const int32_t *parent, int32_t *depth, int32_t first, int32_t capacity, int32_t *flags) {
    const int32_t c = first + (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= capacity) return;
    int32_t cur = c, hops = 0;
    while (cur >= first && hops < 64) {
        const int32_t pc = parent[cur] >> 3;
        if (pc < 0 || pc >= capacity) return;  // not linked (yet): left at depth 0, as before
        cur = pc;
        ++hops;
    }
    const int32_t d = depth[cur] + hops;
    depth[c] = d;
    atomicMax(&flags[1], d);
}

// node words of the appended chunks' voxels and the link word of the voxel each of them hangs under;
// flags[2] = 1 when that voxel was shallow enough to be held by a lookup grid
__device__ __forceinline__ uint32_t patch_items(int32_t d, int32_t L2);

__global__ void accel_refresh_nodes(const int32_t *child, const int32_t *parent, const uint16_t *data, const int32_t *depth, uint32_t *nodes,
                                    int32_t first, int32_t capacity, int32_t data_dim, int32_t grid_depth, int32_t *flags, uint32_t *items, int32_t L2) {
    const int64_t v = (int64_t)first * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (int64_t)capacity * 8) return;
    const int32_t c = (int32_t)(v >> 3);
    const int32_t skip = child[v];
    if (skip != 0) {
        nodes[v] = (uint32_t)(c + skip);
    } else {
        nodes[v] = kLeafBit | (((uint32_t)depth[c] & 0x7fu) << 16) | (uint32_t)data[v * data_dim + data_dim - 1];
    }
    if ((v & 7) == 0) {
        const int32_t pv = parent[c];
        nodes[pv] = (uint32_t)c;
        if (depth[pv >> 3] <= grid_depth) flags[2] = 1;
        atomicMin(&flags[3], depth[pv >> 3]);  // the shallowest voxel that stopped being a leaf
        if (items) items[c - first] = patch_items(depth[pv >> 3], L2);  // (accel_patch_plan turns the counts into first items)
    }
}

// existing leaves whose data row was rewritten (mnv_apply_sample_results): sigma in the node word, colour row
__global__ void accel_refresh_changed(const int32_t *changed_nodes, int32_t n, const int32_t *child, const uint16_t *data, const int32_t *depth,
                                      uint32_t *nodes, uint16_t *rows, int32_t data_dim, int32_t per_chan, int32_t chan_halfs,
                                      int32_t row_halfs, int32_t grid_depth, int32_t *flags, uint32_t *items, int32_t L2) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t c = changed_nodes[2 * i];
    const int64_t v = (int64_t)c * 8 + changed_nodes[2 * i + 1];
    if (child[v] != 0) {
        if (items) items[i] = 0u;
        return;
    }
    nodes[v] = kLeafBit | (((uint32_t)depth[c] & 0x7fu) << 16) | (uint32_t)data[v * data_dim + data_dim - 1];
    for (int ch = 0; ch < 3; ++ch)
        for (int32_t k = 0; k < chan_halfs; ++k) rows[v * row_halfs + ch * chan_halfs + k] = k < per_chan ? data[v * data_dim + ch * per_chan + k] : (uint16_t)0;
    if (depth[c] <= grid_depth) flags[2] = 1;
    atomicMin(&flags[3], depth[c]);
    if (items) items[i] = patch_items(depth[c], L2);
}

// Rewrite the level-L2 lookup cells covered by voxels that stopped being (or changed as) leaves -- voxel b of the list is vox_list[b],
// or the parent voxel of chunk first_chunk + b when vox_list is NULL.  A voxel of depth d covers 8^(L2 - d) cells: one for most of a
// refinement step's voxels, two million for the depth-2 voxels the vote prefers.  A fixed number of slices per voxel either drowns
// the device in empty workgroups or leaves the shallow voxels to a few thousand threads (round 3: 32 slices, 60 us of a configs[4]
// frame), so the work is cut into ITEMS of kPatchCells cells: accel_patch_plan counts every voxel's items and scans the counts, the
// host reads the total with the refresh flags it waits for anyway, and accel_patch_grid2 runs one workgroup per item.
constexpr int kPatchCells = 4096;

__device__ __forceinline__ int64_t patch_voxel(const int32_t *vox_pairs, int32_t first_chunk, const int32_t *parent, int32_t b) {
    return vox_pairs ? (int64_t)vox_pairs[2 * b] * 8 + vox_pairs[2 * b + 1] : (int64_t)parent[first_chunk + b];
}

__device__ __forceinline__ uint32_t patch_items(int32_t d, int32_t L2) {
    return d >= 1 && d <= L2 ? (uint32_t)((((uint64_t)1 << (3 * (L2 - d))) + kPatchCells - 1) / kPatchCells) : 0u;
}

// in: prefix[b] = items of voxel b (written by the refresh kernels); out: prefix[b] = items of voxels 0 .. b-1, prefix[n] = all of
// them, also written to *total.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void accel_patch_plan(int32_t n, uint32_t *prefix, int32_t *total) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_run;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_run = 0u;
    __syncthreads();
    for (int32_t base = 0; base < n; base += 1024) {
        const int32_t b = base + t;
        const uint32_t items = b < n ? prefix[b] : 0u;
        uint32_t incl = items;  // inclusive scan: within the wavefront, then over the 16 wavefronts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t other = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += other;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_run;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (b < n) prefix[b] = before + incl - items;
        __syncthreads();
        if (t == 1023) s_run = before + incl;
        __syncthreads();
    }
    if (t == 0) {
        prefix[n] = s_run;
        *total = (int32_t)s_run;
    }
}

// workgroup i: item i.  The voxel's integer coordinates come from the walk up the parent words.
__global__ __launch_bounds__(256) void accel_patch_grid2(const int32_t *vox_pairs, int32_t first_chunk, int32_t n, const uint32_t *prefix, const int32_t *parent,
                                                         const int32_t *depth, const uint32_t *nodes, uint32_t *grid2, uint32_t *grid2_vox, int32_t L2) {
    __shared__ uint32_t s_box[6];  // x, y, z at the voxel's own level; its depth; the voxel; the item's number among the voxel's items
    if (threadIdx.x == 0) {
        int32_t lo = 0, hi = n;  // the last voxel whose first item is <= this one (voxels without cells have no items and are never met)
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        const int64_t pv = patch_voxel(vox_pairs, first_chunk, parent, lo);
        int32_t cur = (int32_t)(pv >> 3);
        const int32_t d = depth[cur];
        uint32_t x = 0, y = 0, z = 0, slot = (uint32_t)(pv & 7);
        for (int k = 0; k < d; ++k) {
            x |= ((slot >> 2) & 1u) << k;
            y |= ((slot >> 1) & 1u) << k;
            z |= (slot & 1u) << k;
            if (cur == 0) break;
            const int32_t p = parent[cur];
            slot = (uint32_t)(p & 7);
            cur = p >> 3;
        }
        s_box[0] = x;
        s_box[1] = y;
        s_box[2] = z;
        s_box[3] = (uint32_t)d;
        s_box[4] = (uint32_t)pv;
        s_box[5] = blockIdx.x - prefix[lo];
    }
    __syncthreads();
    const int d = (int)s_box[3];
    const int sh = L2 - d;  // the voxel covers (2^sh)^3 cells
    const uint32_t bx = s_box[0] << sh, by = s_box[1] << sh, bz = s_box[2] << sh;
    const uint64_t total = (uint64_t)1 << (3 * sh), first = (uint64_t)s_box[5] * kPatchCells;
    const uint64_t last = first + kPatchCells < total ? first + kPatchCells : total;
    for (uint64_t i = first + threadIdx.x; i < last; i += blockDim.x) {
        const uint32_t ix = bx + (uint32_t)(i >> (2 * sh)), iy = by + (uint32_t)((i >> sh) & ((1u << sh) - 1u)), iz = bz + (uint32_t)(i & ((1u << sh) - 1u));
        // the walk starts at the voxel itself (every cell of its box passes through it), not at the root: a split voxel's cells end one
        // level below it -- two dependent loads instead of L2
        uint32_t vox = s_box[4], word = nodes[vox];
        for (int32_t l = d + 1; l <= L2 && !(word & kLeafBit); ++l) {
            const int32_t s2 = L2 - l;
            const uint32_t cidx = (((ix >> s2) & 1u) << 2) | (((iy >> s2) & 1u) << 1) | ((iz >> s2) & 1u);
            vox = word * 8u + cidx;
            word = nodes[vox];
        }
        const uint32_t o = grid2_index(ix, iy, iz, L2);
        grid2[o] = word;
        grid2_vox[o] = vox;
    }
}

// ---- the packed layout follows a prune (mnv_prune_tree_accel): chunk c survives as c - shifts[c] unless to_delete[c]; a voxel whose
// child chunk is deleted becomes a leaf (the marks are closed under ancestors, so a deleted chunk's whole sub-tree goes with it).
// All three kernels read the OLD numbering: they run before the tree arrays are fixed up and compacted.

__device__ __forceinline__ uint32_t pruned_leaf_word(const int32_t *depth, const uint16_t *data, int32_t data_dim, int32_t chunk, int32_t slot) {
    return kLeafBit | (((uint32_t)depth[chunk] & 0x7fu) << 16) | (uint32_t)data[((int64_t)chunk * 8 + slot) * data_dim + data_dim - 1];
}

__global__ void accel_prune_nodes(const uint32_t *nodes, const int32_t *depth, const uint16_t *data, int32_t data_dim, const uint8_t *to_delete,
                                  const int32_t *shifts, int32_t capacity, uint32_t *nodes_out, int32_t *depth_out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (int64_t)capacity * 8) return;
    const int32_t c = (int32_t)(v >> 3);
    if (to_delete[c]) return;
    uint32_t word = nodes[v];
    if (!(word & kLeafBit)) {
        const int32_t cc = (int32_t)word;
        word = to_delete[cc] ? pruned_leaf_word(depth, data, data_dim, c, (int32_t)(v & 7)) : (uint32_t)(cc - shifts[cc]);
    }
    const int32_t nc = c - shifts[c];
    nodes_out[(int64_t)nc * 8 + (v & 7)] = word;
    if ((v & 7) == 0) depth_out[nc] = depth[c];
}

__global__ void accel_prune_rows(const uint4 *rows, const uint8_t *to_delete, const int32_t *shifts, int32_t capacity, int32_t quads_per_chunk,
                                 uint4 *rows_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)capacity * quads_per_chunk) return;
    const int32_t c = (int32_t)(i / quads_per_chunk);
    if (to_delete[c]) return;
    rows_out[(int64_t)(c - shifts[c]) * quads_per_chunk + (i - (int64_t)c * quads_per_chunk)] = rows[i];
}

__global__ void accel_prune_grid(uint32_t *grid, uint32_t *grid_vox, int64_t cells, const int32_t *parent, const int32_t *depth, const uint16_t *data,
                                 int32_t data_dim, const uint8_t *to_delete, const int32_t *shifts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    uint32_t word = grid[i];
    int32_t c, slot = 0;
    if (word & kLeafBit) {
        const uint32_t vox = grid_vox[i];
        c = (int32_t)(vox >> 3);
        slot = (int32_t)(vox & 7u);
        if (!to_delete[c]) {
            grid_vox[i] = (uint32_t)(c - shifts[c]) * 8u + (uint32_t)slot;
            return;
        }
    } else {
        c = (int32_t)word;  // the chunk of the cell's children
        if (!to_delete[c]) {
            grid[i] = (uint32_t)(c - shifts[c]);
            return;
        }
    }
    // the covering voxel sits in (or points into) a deleted sub-tree: the leaf is now the voxel under which the first deleted chunk hung
    int32_t pc;
    do {
        const int32_t pv = parent[c];
        pc = pv >> 3;
        slot = pv & 7;
        c = pc;
    } while (to_delete[pc]);
    grid[i] = pruned_leaf_word(depth, data, data_dim, pc, slot);
    grid_vox[i] = (uint32_t)(pc - shifts[pc]) * 8u + (uint32_t)slot;
}

// Per-launch parameters that live in device memory: zeroes the ray-queue heads of every frame and
// stores the camera blocks (handed over by value, so no host staging buffer or copy engine is involved).
constexpr int kStageCams = 32;
struct StageCams {
    CamBlock c[kStageCams];
};
__global__ void stage_launch_kernel(uint32_t *heads, int32_t head_words, CamBlock *dst, const StageCams cams, int32_t count) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < head_words) heads[i] = 0u;
    if (i < count * (int32_t)(sizeof(CamBlock) / 4))
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(cams.c)[i];
}

// Visit marks on the packed layout: the march marks the chunk that holds each leaf it steps through; the reference marks every
// chunk of every descent (query_single_from_root, rt_core.cuh:132-134), i.e. those chunks and all their ancestors.  One thread
// per marked chunk walks up the parent words until it meets a chunk that is marked already.
__global__ void close_visit_marks(int32_t *visited, const int32_t *parent, int32_t capacity) {
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= capacity || c == 0 || visited[c] == 0) return;
    int32_t p = parent[c] >> 3;
    while (p >= 0 && p < capacity) {
        if (__hip_atomic_load(&visited[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        visited[p] = 1;
        if (p == 0) break;
        p = parent[p] >> 3;
    }
}

// ------------------------------------------------------------------------ march kernel

struct AccelLaunch {
    FrameParams P;   // tile, options, outputs (P.cam is unused: cameras come from `cams`)
    AccelView A;
    const CamBlock *__restrict__ cams;  // [n_frames] device array (written by stage_launch_kernel)
    uint32_t n_frames;
    uint32_t frame_stride_px;  // pixels between consecutive frames in the output buffers
    uint32_t *queue;           // [kNumQueues] heads, 64 B apart; queue q = band q of every frame, frame-major
    uint32_t tiles_x, n_tiles;
    uint32_t tile_wlog;                   // log2 of the ray-tile width (tile = 2^wlog x 2^(6-wlog) pixels, 8x8 by default)
    uint32_t band_begin[kNumQueues + 1];  // tile ranges per queue
    int32_t lds_level;                    // levels staged in LDS (<= A.grid_level)
    int32_t refill_min;                   // refill a wavefront once this many lanes are idle
    // interleaved macro-tile partition (part_world = 0: plain tile)
    int32_t part_rank, part_world, part_period;  // part_period: mnv_partition.root_period
    uint32_t macro_w, macro_h;            // macro tile size in pixels
    uint32_t macros_x;                    // macro tiles per row of the rectangle
    uint32_t micro_x, micro_per_macro;    // 8x8 micro tiles per macro-tile row / per macro tile
    unsigned long long *stats;            // MODE 1 only: 16 counters
    int32_t count_stats;                  // MODE 1 only: 0 = ablation run without the counters' atomics
    unsigned long long *timeline;         // MODE 1 only (MNV_TIMELINE): per tile {t_grab, t_done, wave, iterations}, then per wave {t_entry, t_exit} (100 MHz ticks)
    uint32_t timeline_tiles;              // tile records (n_tiles * n_frames)
    // MODE 2 only: refinement trackers (rt_core.cuh:179-180,237-252,308-321), indexed like the pixels
    float *split_track, *sample_track;
    const int16_t *sample_counts;         // reference layout [capacity][8], may be NULL
    int32_t max_depth, max_sample_count;
    int32_t *visited;                     // MODE 2 / 3 only: visit marks [capacity]; the march marks the chunk of every leaf it steps through,
                                          // close_visit_marks adds the ancestors (= every chunk of every descent, rt_core.cuh:132-134)
    int32_t ablate;                       // diagnostics only (breaks results): 1 no colour, 2 no dense samples, 4 cached rows
    // MODE 3 only: the sample-emitting march of guided sampling (rt_core.cuh:418-576) -- no colour, rows of
    // (z, world xyz[, view dir][, embedding]) per dense step and the trackers of MODE 2
    int32_t max_guided_samples, samples_dim, need_viewdir, appearance_embedding;
    int16_t *num_samples;
    float *samples;
    int16_t *cluster_indices;
    int32_t grid_dim[2];
    float min_position[3], range[3];
};

// ray id -> pixel of the rectangle (bx, by) and index of the pixel in the output buffer
__device__ __forceinline__ bool ray_pixel(const AccelLaunch &K, uint32_t id, int &bx, int &by, uint32_t &pix) {
    const uint32_t tile = id >> 6, w = id & 63u;
    if (K.part_world < 1) {
        const uint32_t tx = tile % K.tiles_x, ty = tile / K.tiles_x;
        bx = (int)((tx << K.tile_wlog) + (w & ((1u << K.tile_wlog) - 1u)));
        by = (int)((ty << (6 - K.tile_wlog)) + (w >> K.tile_wlog));
        pix = (uint32_t)by * (uint32_t)K.P.tw + (uint32_t)bx;
    } else {
        const uint32_t j = tile / K.micro_per_macro, u = tile % K.micro_per_macro;
        const uint32_t mx = u % K.micro_x, my = u / K.micro_x;
        const uint32_t m = part_tile_of(j, K.part_rank, K.part_world, K.part_period);
        const uint32_t MX = m % K.macros_x, MY = m / K.macros_x;
        const uint32_t lx = mx * 8 + (w & 7u), ly = my * 8 + (w >> 3);
        bx = (int)(MX * K.macro_w + lx);
        by = (int)(MY * K.macro_h + ly);
        pix = (j * K.macro_h + ly) * K.macro_w + lx;
    }
    return bx < K.P.tw && by < K.P.th;
}

__device__ __forceinline__ float lane_read(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// bytes of one colour channel block of a packed row: basis_dim halfs padded to a whole dword
__host__ __device__ constexpr int chan_bytes_for(int basis) { return basis > 0 ? ((2 * basis + 3) / 4) * 4 : 4; }
// bytes of a packed row: three channel blocks rounded up to a power of two (16 ... 256), so that a
// row never straddles a 128-B cache line (SH9: 3 * 20 = 60 -> 64 B)
__host__ __device__ constexpr int row_bytes_pow2(int basis) {
    if (basis <= 0) return 8;
    int r = 16;
    while (r < 3 * chan_bytes_for(basis)) r *= 2;
    return r;
}
// channel block as dwords with 4-byte alignment (the compiler picks the widest legal loads)
template <int N>
struct __attribute__((packed, aligned(4))) ChanWords {
    uint32_t w[N];
};

// One step of the march on integer cell coordinates.  pos in [0, 1-1e-6] is scaled by 2^Lq
// (Lq = deepest voxel depth of the tree, <= 23: the product is exact and < 2^24) and truncated;
// bit (Lq - d) of each coordinate is the child index at depth d, and the cell numbers of the two
// lookup grids are plain shifts.  The in-leaf coordinates are fract(pos * 2^depth), which equals the
// reference's iterated x*2 - floor(x*2) bit for bit (all three operations are exact in binary32).
template <int BASIS, int BLOCK, int MODE /* 0 plain, 1 statistics, 2 refinement trackers, 3 trackers + emitted samples instead of colour, 4 plain with fast colour math, 5 depth image (render_depth) */>
#ifndef MNV_TRACK_WAVES
#define MNV_TRACK_WAVES 6  // tracker / sample modes carry six more live values per ray
#endif
#ifndef MNV_MIN_WAVES
#define MNV_MIN_WAVES 8  // register budget for 8 waves per SIMD: the few spills land in the ray set-up (A/B in DESIGN.md)
#endif
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
    uint32_t *s_grid = s_mem + 64 + MAPW + (NB + 2) * BLOCK;  // (2^lds_level)^3 words
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

    // per-lane ray state
    float t = 0.f, T = 1.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    float dir0 = 0.f, dir1 = 0.f, dir2 = 0.f, inv0 = 0.f, inv1 = 0.f, inv2 = 0.f, tmax = 0.f;
    bool alive = false;
    // MODE 2: per-ray tracker state
    float max_weight = -1.f, max_sample_weight = -1.f, sp_prio = 0.f, sa_prio = 0.f;
    int32_t sp_vox = -1, sa_vox = -1;
    int32_t ns = 0;  // MODE 3: samples emitted by this ray so far
    static_assert(MODE != 3 || NB >= 6, "MODE 3 keeps the world-space ray in the LDS slots of the SH basis");
    auto write_trackers = [&](uint32_t p) {
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
    const CamBlock *__restrict__ Cp = K.cams;  // camera of `frame` (wave-uniform pointer: scalar loads)
    float cen0 = Cp->cen[0], cen1 = Cp->cen[1], cen2 = Cp->cen[2];
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
                        }
                        if constexpr (MODE == 3) ns = K.num_samples[pix];
                        RaySetup<NB> r;
                        setup_ray<(BASIS > 0 ? BASIS : 0)>(P, *Cp, P.x0 + bx, P.y0 + by, r);
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
        float delta_t = 0.f, weight = 0.f, att = 1.f;
        uint32_t vox = 0;
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
                        word = A.grid2[vox];
                        src = 1;
                        sh = sh2;
                    }
                    while (!(word & kLeafBit)) {
                        stat(6, true);
                        --sh;
                        uint32_t v = (word << 1) | __builtin_amdgcn_ubfe(q[0], (uint32_t)sh, 1u);
                        v = (v << 1) | __builtin_amdgcn_ubfe(q[1], (uint32_t)sh, 1u);
                        vox = (v << 1) | __builtin_amdgcn_ubfe(q[2], (uint32_t)sh, 1u);
                        word = A.nodes[vox];
                        src = 2;
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
                if constexpr (MODE == 2 || MODE == 3) need_vox = is_dense || max_weight == -1.f || max_sample_weight == -1.f || K.visited != nullptr;
                if (need_vox) {
                    // voxel index of a leaf that was answered by one of the lookup grids
                    if (src == 0) vox = A.grid_vox[((((q[0] >> shg) << A.grid_level) + (q[1] >> shg)) << A.grid_level) + (q[2] >> shg)];
                    else if (src == 1) vox = A.grid2_vox[vox];
                }
                if constexpr (MODE == 2 || MODE == 3) {
                    // the mark only ever goes 0 -> 1: load + conditional plain store (mnv_march_ref_layout.hip does the same per level)
                    if (K.visited && __hip_atomic_load(&K.visited[vox >> 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) K.visited[vox >> 3] = 1;
                }
                if (is_dense) {
                    // opacity of a dense sample, rt_core.cuh:233-235
                    dense = true;
                    att = exact_expf(-delta_t * my_ray[NB * BLOCK] * sigma, s_exp);
                    weight = T * (1.f - att);
                }
                if constexpr (MODE == 2 || MODE == 3) {
                    // rt_core.cuh:237-252 (dense leaf: best weight so far) and :308-321 (first leaf before any dense one)
                    if (need_vox) {
                        const bool split_ok = depth < K.max_depth && (is_dense ? weight > max_weight : max_weight == -1.f);
                        if (split_ok) {
                            sp_vox = (int32_t)vox;
                            sp_prio = (float)depth;
                            if (is_dense) max_weight = weight;
                        }
                        if (K.sample_counts && (is_dense ? weight > max_sample_weight : max_sample_weight == -1.f)) {
                            const int16_t sc = K.sample_counts[vox];
                            if (sc < K.max_sample_count) {
                                sa_vox = (int32_t)vox;
                                sa_prio = (float)sc;
                                if (is_dense) max_sample_weight = weight;
                            }
                        }
                    }
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
            if constexpr (MODE == 3) {
                // no colour: the networks supply it (render_nerf_results)
            } else if (depth_mode()) {
                if (dense) o0 += weight * t;
            } else if (ablate(1)) {
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
                    constexpr int NW = CHAN_BYTES / 4;
                    uint32_t vx = vox;
                    if (ablate(4)) vx &= 0xffffu;  // diagnostics: rows served from cache (wrong colours)
                    const uint8_t *row = A.rows + (int64_t)vx * ROW_BYTES;
                    const ChanWords<NW> c0 = *reinterpret_cast<const ChanWords<NW> *>(row);
                    const ChanWords<NW> c1 = *reinterpret_cast<const ChanWords<NW> *>(row + CHAN_BYTES);
                    const ChanWords<NW> c2 = *reinterpret_cast<const ChanWords<NW> *>(row + 2 * CHAN_BYTES);
                    if constexpr (MODE == 1) {
                        if (K.count_stats == 2) {
                            const unsigned long long now = phase_clock();   // the rows have arrived
                            ph_row += now - ph_mark;
                            ph_mark = now;
                        }
                    }
                    float b[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) b[k] = my_ray[k * BLOCK];
                    auto chan = [&](const ChanWords<NW> &cw) -> float {
                        auto coef = [&](int k) -> float {
                            const uint32_t wd = cw.w[k >> 1];
                            return half_bits_to_float((uint16_t)((k & 1) ? (wd >> 16) : (wd & 0xffffu)));
                        };
                        const float tmp = sh_channel<BASIS>(b, coef, 0);
                        if constexpr (MODE == 4) {
                            // colour-only arithmetic: it feeds no branch (opacity, transmittance and the step sequence stay exact),
                            // so hardware exp2 / rcp (about 1 ulp each) move a colour by ~1e-7 and nothing else
                            const float e = __builtin_amdgcn_exp2f(tmp * -1.44269504088896341f);
                            return weight * __builtin_amdgcn_rcpf(1.f + e);
                        } else {
                            return weight / (1.f + exact_expf(-tmp, s_exp));
                        }
                    };
                    o0 += chan(c0);
                    o1 += chan(c1);
                    o2 += chan(c2);
                }
            } else {
                // RGBA rows (rt_core.cuh:285-290): three halfs per voxel, per-lane
                if (dense) {
                    const uint2 qd = *reinterpret_cast<const uint2 *>(A.rows + (int64_t)vox * ROW_BYTES);
                    o0 += half_bits_to_float((uint16_t)(qd.x & 0xffffu)) * weight;
                    o1 += half_bits_to_float((uint16_t)(qd.x >> 16)) * weight;
                    o2 += half_bits_to_float((uint16_t)(qd.y & 0xffffu)) * weight;
                }
            }
            if (dense) {
                T *= att;  // rt_core.cuh:293-307
                if (T < P.stop_thresh) {
                    fin = 2;  // early stop: renormalised and written in flush_finished()
                    alive = false;
                }
            }
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

// Rank 0 after the gather: macro tile m of frame f sits at gathered[m % world][f][m / world];
// one thread per pixel, the destination is written row-major (coalesced), the source is read in tile rows.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16 bytes: one float RGBA pixel or four RGBA8 pixels
template <typename PIXEL>
__global__ void assemble_tiles_kernel(const PIXEL *__restrict__ gathered, PIXEL *__restrict__ frames, int32_t width, int32_t height,
                                      int32_t tile_w, int32_t tile_h, int32_t macros_x, int32_t j_max, int32_t world, int32_t period, int32_t n_frames) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_frame = (int64_t)width * height;
    if (idx >= per_frame * n_frames) return;
    const int32_t f = (int32_t)(idx / per_frame);
    const int32_t p = (int32_t)(idx - (int64_t)f * per_frame);
    const int32_t y = p / width, x = p - y * width;
    const int32_t mx = x / tile_w, my = y / tile_h;
    uint32_t r, j;
    part_owner_of((uint32_t)(my * macros_x + mx), world, period, r, j);
    const int64_t src = ((((int64_t)r * n_frames + f) * j_max + j) * tile_h + (y - my * tile_h)) * tile_w + (x - mx * tile_w);
    // streamed once: keep these lines from displacing the march's lookup structures in L2
    __builtin_nontemporal_store(__builtin_nontemporal_load(&gathered[src]), &frames[idx]);
}

}  // namespace mnv

#include "mnv_guided_fused.h"   // guided_fused_kernel: march + per-sample network + composite in one kernel, every wavefront both roles
#include "mnv_guided_fused2.h"  // guided_fused2_kernel: the same frame with producer (march) and consumer (network) wavefronts

namespace mnv {

// ---------------------------------------------------------------------------- host side

static int row_bytes_for(int basis) { return row_bytes_pow2(basis); }

// mnv_set_colour_math: 0 = exact (bit-identical to the oracle, default), 1 = hardware exp2 / rcp in the colour sigmoid
static std::atomic<int> g_fast_colour{0};
// mnv_set_fused_kernel / mnv_set_fused_diag
static std::atomic<int> g_fused_kernel{0};
static std::atomic<unsigned long long *> g_fused_diag{nullptr};

template <int BASIS, int MODE>
static int launch_variant2(const AccelLaunch &K, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    constexpr int BLOCK = 256;
    auto kern = march_accel_kernel<BASIS, BLOCK, MODE>;
    if (lds_bytes > 65536) {  // diagnostics only (MNV_LDS_LEVEL=5); the attribute is per device, so set it on every such launch
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(BLOCK), lds_bytes, stream, K);
    return (int)hipGetLastError();
}

template <int BASIS>
static int launch_variant(const AccelLaunch &K, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    if constexpr (BASIS == 9) {  // MNV_STATS=1 / MNV_ABLATE diagnostics build of the headline variant only
        if (K.stats) return launch_variant2<BASIS, 1>(K, n_blocks, lds_bytes, stream);
    }
    if constexpr (BASIS == 9) {  // the sample-emitting march reads no colour rows: one instantiation serves every row format
        if (K.samples) return launch_variant2<BASIS, 3>(K, n_blocks, lds_bytes, stream);
    }
    if (K.split_track || K.sample_track || K.visited) return launch_variant2<BASIS, 2>(K, n_blocks, lds_bytes, stream);
    if constexpr (BASIS == 9) {  // the depth image reads no colour rows either
        if (K.P.render_depth) return launch_variant2<BASIS, 5>(K, n_blocks, lds_bytes, stream);
    }
    if constexpr (BASIS >= 1) {
        if (g_fast_colour.load(std::memory_order_relaxed)) return launch_variant2<BASIS, 4>(K, n_blocks, lds_bytes, stream);
    }
    return launch_variant2<BASIS, 0>(K, n_blocks, lds_bytes, stream);
}

// world > 1, or a single rank that asks for the macro-tile-major layout by naming a tile size
static bool is_partitioned(mnv_partition part) { return part.world > 1 || (part.world == 1 && part.tile_w > 0); }
static int32_t root_period_of(mnv_partition part) { return part.world > 1 && part.root_period >= 2 ? part.root_period : 0; }

// Called by mnv_prune_tree_accel between its scan and its fix-up / compaction (old numbering everywhere).
int accel_apply_prune(mnv_accel *a, const int32_t *parent, const uint16_t *data, int32_t data_dim, const uint8_t *to_delete, const int32_t *shifts,
                      int32_t old_capacity, int32_t n_deleted, hipStream_t stream) {
    if (!a || old_capacity != a->view.capacity) return set_error(MNV_E_INVALID, "the accel does not describe the tree that is being pruned");
    int rc;
    const int64_t reserved = a->reserved;
    const int row_bytes = a->view.row_bytes;
    if (!a->nodes_spare) {  // second set of the per-voxel arrays: the survivors are written out of place, then the sets swap
        if ((rc = check_hip(hipMalloc((void **)&a->nodes_spare, reserved * 8 * 4), "hipMalloc(nodes spare)"))) return rc;
        if ((rc = check_hip(hipMalloc((void **)&a->rows_spare, reserved * 8 * row_bytes), "hipMalloc(rows spare)"))) return rc;
        if ((rc = check_hip(hipMalloc((void **)&a->depth_spare, reserved * 4), "hipMalloc(depth spare)"))) return rc;
    }
    if ((rc = check_hip(hipMemsetAsync(a->depth_spare, 0, reserved * 4, stream), "memset depth"))) return rc;
    const int64_t nvox = (int64_t)old_capacity * 8;
    hipLaunchKernelGGL(accel_prune_nodes, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, stream, a->nodes, a->depth, data, data_dim, to_delete, shifts,
                       old_capacity, a->nodes_spare, a->depth_spare);
    const int32_t quads = 8 * row_bytes / 16;
    if (8 * row_bytes % 16) return set_error(MNV_E_UNSUPPORTED, "row size");
    const int64_t nq = (int64_t)old_capacity * quads;
    hipLaunchKernelGGL(accel_prune_rows, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(a->rows), to_delete, shifts,
                       old_capacity, quads, reinterpret_cast<uint4 *>(a->rows_spare));
    const int64_t gcells = (int64_t)1 << (3 * a->view.grid_level);
    hipLaunchKernelGGL(accel_prune_grid, dim3((unsigned)((gcells + 255) / 256)), dim3(256), 0, stream, a->grid, a->grid_vox, gcells, parent, a->depth, data,
                       data_dim, to_delete, shifts);
    if (a->view.grid2_level > 0) {
        const int64_t g2 = (int64_t)1 << (3 * a->view.grid2_level);
        hipLaunchKernelGGL(accel_prune_grid, dim3((unsigned)((g2 + 255) / 256)), dim3(256), 0, stream, a->grid2, a->grid2_vox, g2, parent, a->depth, data,
                           data_dim, to_delete, shifts);
    }
    if ((rc = check_hip(hipGetLastError(), "accel prune launch"))) return rc;
    std::swap(a->nodes, a->nodes_spare);
    std::swap(a->rows, a->rows_spare);
    std::swap(a->depth, a->depth_spare);
    a->view.nodes = a->nodes;
    a->view.rows = a->rows;
    a->view.capacity = old_capacity - n_deleted;  // max_depth stays an upper bound (the march only needs pos * 2^max_depth < 2^24)
    return MNV_OK;
}

int32_t partition_local_tiles(mnv_rect tile, mnv_partition part) {
    if (tile.w <= 0 || tile.h <= 0 || !is_partitioned(part)) return is_partitioned(part) ? 0 : 1;
    const int64_t mx = (tile.w + part.tile_w - 1) / part.tile_w, my = (tile.h + part.tile_h - 1) / part.tile_h;
    return (int32_t)part_local_count(mx * my, part.rank, part.world, root_period_of(part));
}

// Refinement trackers of one launch (all device pointers; rows indexed like the pixels).
struct AccelTrack {
    float *split_track, *sample_track;
    const int16_t *sample_counts;
    int32_t max_depth, max_sample_count;
    // sample emission (MODE 3) when samples != NULL
    int16_t *num_samples;
    float *samples;
    int16_t *cluster_indices;
    int32_t max_guided_samples, samples_dim, need_viewdir, appearance_embedding;
    const mnv_cluster_grid *grid;
    const FusedGuided *fused;  // non-NULL: guided_fused_kernel instead of the march (no trackers, one frame)
    int32_t *visited;          // visit marks (tracker / sample modes) ...
    const int32_t *parent;     // ... closed under the parent words after the march
};

constexpr int kUnsupportedBasis = -1000;  // not a hipError_t

int launch_accel(const mnv_accel *accel, const FrameParams &P, const CamBlock *cams, int n_frames, mnv_partition part,
                 const AccelTrack *track, hipStream_t stream) {
    if (P.tw <= 0 || P.th <= 0 || n_frames <= 0) return 0;
    AccelLaunch K;
    std::memset(static_cast<void *>(&K), 0, sizeof(K));
    K.P = P;
    if (track) {
        K.split_track = track->split_track;
        K.sample_track = track->sample_track;
        K.sample_counts = track->sample_counts;
        K.max_depth = track->max_depth;
        K.max_sample_count = track->max_sample_count;
        K.visited = track->visited;
        if (track->samples) {
            K.num_samples = track->num_samples;
            K.samples = track->samples;
            K.cluster_indices = track->cluster_indices;
            K.max_guided_samples = track->max_guided_samples;
            K.samples_dim = track->samples_dim;
            K.need_viewdir = track->need_viewdir;
            K.appearance_embedding = track->appearance_embedding;
            for (int i = 0; i < 2; ++i) K.grid_dim[i] = track->grid->grid_dim[i];
            for (int i = 0; i < 3; ++i) {
                K.min_position[i] = track->grid->min_position[i];
                K.range[i] = track->grid->range[i];
            }
        }
    }
    K.A = accel->view;
    K.part_rank = part.rank;
    K.part_world = is_partitioned(part) ? part.world : 0;
    K.part_period = root_period_of(part);
    static const int env_wlog = getenv("MNV_TILE_WLOG") ? atoi(getenv("MNV_TILE_WLOG")) : 3;
    K.tile_wlog = (env_wlog >= 0 && env_wlog <= 6) ? (uint32_t)env_wlog : 3u;
    if (!is_partitioned(part)) {
        const uint32_t tile_w = 1u << K.tile_wlog, tile_h = 64u >> K.tile_wlog;
        K.tiles_x = (uint32_t)((P.tw + tile_w - 1) / tile_w);
        const uint32_t tiles_y = (uint32_t)((P.th + tile_h - 1) / tile_h);
        K.n_tiles = K.tiles_x * tiles_y;
        // contiguous bands of tile rows per queue
        for (int q = 0; q <= kNumQueues; ++q) K.band_begin[q] = (uint32_t)(((uint64_t)tiles_y * q) / kNumQueues) * K.tiles_x;
    } else {
        const mnv_rect rect = {P.x0, P.y0, P.tw, P.th};
        const uint32_t local = (uint32_t)partition_local_tiles(rect, part);
        if (local == 0) return 0;
        K.macro_w = (uint32_t)part.tile_w;
        K.macro_h = (uint32_t)part.tile_h;
        K.macros_x = (uint32_t)((P.tw + part.tile_w - 1) / part.tile_w);
        K.micro_x = K.macro_w / 8;
        K.micro_per_macro = K.micro_x * (K.macro_h / 8);
        K.tiles_x = K.micro_x;
        K.n_tiles = local * K.micro_per_macro;
        // contiguous runs of micro tiles (in local macro-tile order) per queue
        for (int q = 0; q <= kNumQueues; ++q) K.band_begin[q] = (uint32_t)(((uint64_t)K.n_tiles * q) / kNumQueues);
    }
    static const int env_queues = getenv("MNV_QUEUES") ? atoi(getenv("MNV_QUEUES")) : kNumQueues;
    if (env_queues >= 1 && env_queues < kNumQueues) {
        // diagnostics: fewer, larger queues (queue q of the first env_queues covers 1/env_queues of the tiles)
        const uint32_t total = K.band_begin[kNumQueues];
        for (int q = 0; q <= kNumQueues; ++q)
            K.band_begin[q] = q >= env_queues ? total : (uint32_t)(((uint64_t)total * q) / env_queues);
    }
    // per-launch slot: zeroed queue heads + the camera blocks, written by stage_launch_kernel on the launch stream
    K.n_frames = (uint32_t)n_frames;
    // frames of a batch are j_max = ceil(macro tiles / world) local tiles apart on EVERY rank, so that the
    // per-rank buffers have one shape (what the gather needs) even when the tile count is ragged
    if (!is_partitioned(part)) {
        K.frame_stride_px = (uint32_t)P.tw * (uint32_t)P.th;
    } else {
        const uint32_t n_macro = K.macros_x * (uint32_t)((P.th + part.tile_h - 1) / part.tile_h);
        K.frame_stride_px = (uint32_t)part_j_max(n_macro, part.world, root_period_of(part)) * K.macro_w * K.macro_h;
    }
    // The per-launch slot bookkeeping is the handle's only mutable state on this path; launches from several host threads
    // (or one thread feeding several streams) serialise here from the slot's acquisition to the record of its event.
    mnv_accel *mut = const_cast<mnv_accel *>(accel);
    std::lock_guard<std::mutex> launch_lock(mut->launch_mutex);
    const uint32_t slot = mut->slot_counter.fetch_add(1) % kSlots;
    // a caller that runs more than kSlots launches ahead of the device waits here for the launch that last used the slot
    if (mut->slot_used[slot]) {
        hipError_t es = hipEventSynchronize(mut->slot_done[slot]);
        if (es != hipSuccess) return (int)es;
    }
    uint8_t *ds = accel->slots_dev + (size_t)slot * kSlotBytes;
    const size_t heads_bytes = (size_t)kNumQueues * 64;  // one head per queue; a queue spans the frames of the batch
    K.queue = reinterpret_cast<uint32_t *>(ds);
    CamBlock *dcams = reinterpret_cast<CamBlock *>(ds + heads_bytes);
    K.cams = dcams;
    for (int first = 0; first < n_frames; first += kStageCams) {
        StageCams sc;
        const int count = n_frames - first < kStageCams ? n_frames - first : kStageCams;
        std::memcpy(sc.c, cams + first, (size_t)count * sizeof(CamBlock));
        const int head_words = first == 0 ? (int)(heads_bytes / 4) : 0;
        const int n_threads = std::max(head_words, count * (int)(sizeof(CamBlock) / 4));
        hipLaunchKernelGGL(stage_launch_kernel, dim3((n_threads + 255) / 256), dim3(256), 0, stream, K.queue, head_words, dcams + first, sc, count);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;

    // diagnostics knobs; read once
    static const int env_level = getenv("MNV_LDS_LEVEL") ? atoi(getenv("MNV_LDS_LEVEL")) : -1;
    int lds_level = accel->view.grid_level < 3 ? accel->view.grid_level : 3;  // 2 KB; level 4 (16 KB) measured equal and costs occupancy
    if (env_level >= 1 && env_level <= accel->view.grid_level) lds_level = env_level;
    K.lds_level = lds_level;
    // sample emission and the depth image read no colour rows: one instantiation (BASIS 9) serves every row format
    const bool colourless = K.samples != nullptr || (P.render_depth && !K.split_track && !K.sample_track && !K.visited);
    const int nb_lds = colourless ? 9 : (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim > 0) ? accel->view.basis_dim : 1;
    const size_t lds_bytes = 256 + (nb_lds >= 16 ? 1024 : 0) + (size_t)(nb_lds + 2) * 256 * 4 + ((size_t)4 << (3 * lds_level));
    static const int env_bpc = getenv("MNV_BLOCKS_PER_CU") ? atoi(getenv("MNV_BLOCKS_PER_CU")) : 0;
    static const int env_refill = getenv("MNV_REFILL_MIN") ? atoi(getenv("MNV_REFILL_MIN")) : 0;
    static const int env_ablate = getenv("MNV_ABLATE") ? atoi(getenv("MNV_ABLATE")) : 0;
    K.ablate = env_ablate;
    static const bool env_stats = getenv("MNV_STATS") != nullptr;
    static const char *env_timeline = getenv("MNV_TIMELINE");
    K.stats = (env_stats || env_ablate || env_timeline) ? accel->stats : nullptr;  // all three run on the diagnostics instantiation
    static const int env_stats_level = getenv("MNV_STATS") ? std::max(1, atoi(getenv("MNV_STATS"))) : 0;
    K.count_stats = env_stats ? env_stats_level : 0;
    K.refill_min = (env_refill > 0 && n_frames == 1) ? env_refill : 64;  // batches refill whole tiles (a grab must not straddle frames);  // sweep in DESIGN.md: 16 -> 0.606 ms, 32 -> 0.535, 48 -> 0.507, 56 -> 0.504, 64 -> 0.506
    int blocks_per_cu = lds_level >= 5 ? 1 : (lds_level == 4 ? 6 : 8);
    if ((K.split_track || K.sample_track || K.samples || K.visited) && blocks_per_cu > MNV_TRACK_WAVES) blocks_per_cu = MNV_TRACK_WAVES;
    if (env_bpc > 0) blocks_per_cu = env_bpc;
    int n_blocks = accel->num_cus * blocks_per_cu;
    const uint64_t n_waves_needed = (uint64_t)K.n_tiles * (uint64_t)n_frames;  // one initial 8x8 tile per wave
    if ((uint64_t)n_blocks * 4u > n_waves_needed) n_blocks = (int)((n_waves_needed + 3) / 4);
    if (n_blocks < 1) n_blocks = 1;

    if (env_timeline && K.stats) {
        // diagnostics: (re)allocate the record buffer of this launch; mnv_accel_destroy writes the last launch's records to the file
        const size_t tiles = (size_t)K.n_tiles * (size_t)n_frames, waves = (size_t)n_blocks * 4;
        const size_t bytes = (tiles * 4 + waves * 2) * 8;
        if (mut->timeline_bytes < bytes) {
            if (mut->timeline) (void)hipFree(mut->timeline);
            mut->timeline = nullptr;
            if (hipMalloc((void **)&mut->timeline, bytes) != hipSuccess) return (int)hipErrorOutOfMemory;
            mut->timeline_bytes = bytes;
        }
        (void)hipMemsetAsync(mut->timeline, 0, bytes, stream);
        mut->timeline_tiles = tiles;
        mut->timeline_waves = waves;
        mut->timeline_tiles_per_frame = K.n_tiles;
        K.timeline = mut->timeline;
        K.timeline_tiles = (uint32_t)tiles;
    }
    int rc = kUnsupportedBasis;
    const int b = (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim >= 0) ? accel->view.basis_dim : -1;
    if (track && track->fused) {
        const int nb = b > 0 ? b : 1;
        const bool two = track->fused->S.nkk0 == 2;  // mnv_render_guided_fused admits 1 and 2
        const bool trk = K.split_track || K.sample_track || K.visited;
        // producer / consumer wavefronts when one sub-module's weights fit a workgroup's LDS beside the rings (at least two workgroups per CU)
        FusedGuided F = *track->fused;
        F.fault = accel->fault_dev;
        int slots = kF2NS;  // weight slots: as many as fit beside the rings (at least one per two consumers)
        const int f2_per_cu = (4 * kF2WavesPerSimd) / (kF2NP + kF2NC);  // workgroups per CU the kernel is built for
        while (slots > 1 && (size_t)f2_layout(nb, lds_level, F.S, slots).total * 4 > (size_t)160 * 1024 / f2_per_cu) --slots;
        F.weight_slots = slots;
        const size_t f2_bytes = (size_t)f2_layout(nb, lds_level, F.S, slots).total * 4;
        const int version = g_fused_kernel.load(std::memory_order_relaxed);
        const bool fits2 = f2_bytes <= (size_t)160 * 1024 / f2_per_cu && slots >= (kF2NS < 2 ? kF2NS : 2) && F.S.bias_floats <= 256;  // (a consumer refills a sub-module's biases with four loads per lane)
        if (fits2 && (version == 2 || (version == 0 && kF2Default))) {
            int per_cu = (int)((size_t)160 * 1024 / f2_bytes);
            const int by_regs = (4 * kF2WavesPerSimd) / (kF2NP + kF2NC);
            if (per_cu > by_regs) per_cu = by_regs;
            static const int env_f2 = getenv("MNV_F2_BLOCKS_PER_CU") ? atoi(getenv("MNV_F2_BLOCKS_PER_CU")) : 0;
            if (env_f2 > 0 && env_f2 < per_cu) per_cu = env_f2;
            int fb = accel->num_cus * per_cu;
            if ((uint64_t)fb * kF2NP > n_waves_needed) fb = (int)((n_waves_needed + kF2NP - 1) / kF2NP);
            if (fb < 1) fb = 1;
            auto go2 = [&](auto kern) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f2_bytes);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(kern, dim3(fb), dim3(kF2Block), f2_bytes, stream, K, F);
                return (int)hipGetLastError();
            };
#define MNV_FUSED2_CASE(B)                                                                                        \
    case B:                                                                                                       \
        rc = trk ? (two ? go2(guided_fused2_kernel<B, 2, true>) : go2(guided_fused2_kernel<B, 1, true>))          \
                 : (two ? go2(guided_fused2_kernel<B, 2, false>) : go2(guided_fused2_kernel<B, 1, false>));       \
        break;
            switch (b) {
                MNV_FUSED2_CASE(-1)
                MNV_FUSED2_CASE(1)
                MNV_FUSED2_CASE(4)
                MNV_FUSED2_CASE(9)
                MNV_FUSED2_CASE(16)
                default: break;
            }
#undef MNV_FUSED2_CASE
        } else {
        // the one-role kernel: 2 workgroups per CU (LDS: network tiles; 248 VGPRs), one 8x8 tile per wavefront at a time
        const size_t fl = fused_lds_bytes(nb, lds_level, F.S.mt_out, F.S.nkk0);
        int fb = accel->num_cus * MNV_FUSED_WAVES;
        if ((uint64_t)fb * 4u > n_waves_needed) fb = (int)((n_waves_needed + 3) / 4);
        if (fb < 1) fb = 1;
        auto go = [&](auto kern) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, dim3(fb), dim3(256), fl, stream, K, F);
            return (int)hipGetLastError();
        };
#define MNV_FUSED_CASE(B)                                                                                       \
    case B:                                                                                                     \
        rc = trk ? (two ? go(guided_fused_kernel<B, 2, true>) : go(guided_fused_kernel<B, 1, true>))            \
                 : (two ? go(guided_fused_kernel<B, 2, false>) : go(guided_fused_kernel<B, 1, false>));         \
        break;
        switch (b) {
            MNV_FUSED_CASE(-1)
            MNV_FUSED_CASE(1)
            MNV_FUSED_CASE(4)
            MNV_FUSED_CASE(9)
            MNV_FUSED_CASE(16)
            default: break;
        }
#undef MNV_FUSED_CASE
        }
    } else if (colourless) rc = launch_variant<9>(K, n_blocks, lds_bytes, stream);
    else
        switch (b) {
            case -1: rc = launch_variant<-1>(K, n_blocks, lds_bytes, stream); break;
            case 1: rc = launch_variant<1>(K, n_blocks, lds_bytes, stream); break;
            case 4: rc = launch_variant<4>(K, n_blocks, lds_bytes, stream); break;
            case 9: rc = launch_variant<9>(K, n_blocks, lds_bytes, stream); break;
            case 16: rc = launch_variant<16>(K, n_blocks, lds_bytes, stream); break;
            case 25: rc = launch_variant<25>(K, n_blocks, lds_bytes, stream); break;
            default: break;
        }
    if (rc == 0 && K.visited && track->parent) {
        hipLaunchKernelGGL(close_visit_marks, dim3((unsigned)((accel->view.capacity + 255) / 256)), dim3(256), 0, stream, K.visited, track->parent,
                           accel->view.capacity);
        rc = (int)hipGetLastError();
    }
    if (rc == 0 && track && track->fused)  // the fault word's pinned mirror follows the frame on its stream (guided_fused, mnv_accel_fused_faults)
        rc = (int)hipMemcpyAsync(mut->fault_host, mut->fault_dev, 4, hipMemcpyDeviceToHost, stream);
    if (rc == 0) {
        rc = (int)hipEventRecord(mut->slot_done[slot], stream);
        mut->slot_used[slot] = true;
    }
    return rc;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

int mnv_accel_create(const mnv_tree_view *t, void *hip_stream, mnv_accel **out) {
    return mnv_accel_create_reserved(t, t ? t->capacity : 0, hip_stream, out);
}

// (Re)build every derived array of `a` from the tree: chunk depths, node words, colour rows, lookup grids.  The big arrays
// (nodes, rows, depth) are sized for a->reserved chunks and kept; the grids are reallocated only when their level changes.
static int accel_build(mnv_accel *a, const mnv_tree_view *t, hipStream_t stream) {
    int rc = MNV_OK;
    const int b = (t->format == MNV_FORMAT_SH && t->basis_dim >= 0) ? t->basis_dim : -1;
    const int64_t cap = t->capacity, nvox = cap * 8, max_capacity = a->reserved;
    const int row_bytes = row_bytes_for(b);
    int32_t *depth = a->depth, *changed = a->flags;
    auto fail = [&](int code) { return code; };
    // chunk depths: root chunk holds depth-1 voxels
    if ((rc = check_hip(hipMemsetAsync(depth, 0, max_capacity * 4, stream), "memset depth"))) return fail(rc);
    const int32_t one = 1;
    if ((rc = check_hip(hipMemcpyAsync(depth, &one, 4, hipMemcpyHostToDevice, stream), "seed depth"))) return fail(rc);
    const unsigned nb = (unsigned)((nvox + 255) / 256);
    int max_depth = 1;
    for (int level = 1; level < 25; ++level) {
        int32_t flag = 0;
        if ((rc = check_hip(hipMemsetAsync(changed, 0, 4, stream), "memset flag"))) return fail(rc);
        hipLaunchKernelGGL(accel_depth_pass, dim3(nb), dim3(256), 0, stream, t->child, depth, t->capacity, level, changed);
        if ((rc = check_hip(hipMemcpyAsync(&flag, changed, 4, hipMemcpyDeviceToHost, stream), "read flag"))) return fail(rc);
        if ((rc = check_hip(hipStreamSynchronize(stream), "accel_depth_pass"))) return fail(rc);
        if (!flag) break;
        max_depth = level + 1;
    }
    if (max_depth > 23) return fail(set_error(MNV_E_UNSUPPORTED, "accel supports trees up to depth 23; use mnv_render_voxels"));
    hipLaunchKernelGGL(accel_pack_nodes, dim3(nb), dim3(256), 0, stream, t->child, t->data, depth, a->nodes, t->capacity, t->data_dim);
    hipLaunchKernelGGL(accel_pack_rows, dim3((unsigned)((nvox * 3 + 255) / 256)), dim3(256), 0, stream, t->data,
                       reinterpret_cast<uint16_t *>(a->rows), nvox, t->data_dim, b > 0 ? b : 1, b > 0 ? chan_bytes_for(b) / 2 : 1,
                       row_bytes / 2);
    int L = max_depth < kMaxGridLevel ? max_depth : kMaxGridLevel;
    const int64_t gcells = (int64_t)1 << (3 * L);
    if (!a->grid || a->view.grid_level != L) {
        if (a->grid) (void)hipFree(a->grid);
        if (a->grid_vox) (void)hipFree(a->grid_vox);
        a->grid = a->grid_vox = nullptr;
        if ((rc = check_hip(hipMalloc((void **)&a->grid, gcells * 4), "hipMalloc(grid)"))) return fail(rc);
        if ((rc = check_hip(hipMalloc((void **)&a->grid_vox, gcells * 4), "hipMalloc(grid_vox)"))) return fail(rc);
    }
    hipLaunchKernelGGL(accel_build_grid, dim3((unsigned)((gcells + 255) / 256)), dim3(256), 0, stream, a->nodes, a->grid, a->grid_vox, L);
    // second lookup grid at level L2 <= min(max_depth - 1, 9): 8^L2 words per array (64 MiB at level 8,
    // 512 MiB at level 9).  Pick the deepest level whose two arrays stay below max(128 MiB, 2 x the packed
    // tree): HBM is 288 GB, and every level moved into the grid removes a dependent load from deep steps
    // (cfg2: level 8 -> 0.506 ms/frame, level 9 -> 0.471 ms/frame).
    int L2 = max_depth - 1 < kMaxGrid2Level ? max_depth - 1 : kMaxGrid2Level;
    const int64_t budget = std::max<int64_t>((int64_t)128 << 20, 2 * (nvox * 4 + nvox * row_bytes));
    while (L2 > L && ((int64_t)8 << (3 * L2)) > budget) --L2;
    static const int env_l2 = getenv("MNV_GRID2_LEVEL") ? atoi(getenv("MNV_GRID2_LEVEL")) : -1;
    if (env_l2 >= 0 && env_l2 <= kMaxGrid2Level && env_l2 < max_depth) L2 = env_l2;
    if (L2 <= L || L2 < 2) L2 = 0;
    int64_t g2cells = 0;
    if (a->grid2 && a->view.grid2_level != L2) {
        (void)hipFree(a->grid2);
        (void)hipFree(a->grid2_vox);
        a->grid2 = a->grid2_vox = nullptr;
    }
    if (L2 > 0) {
        g2cells = (int64_t)1 << (3 * L2);
        if (!a->grid2) {
            if ((rc = check_hip(hipMalloc((void **)&a->grid2, g2cells * 4), "hipMalloc(grid2)"))) return fail(rc);
            if ((rc = check_hip(hipMalloc((void **)&a->grid2_vox, g2cells * 4), "hipMalloc(grid2_vox)"))) return fail(rc);
        }
        hipLaunchKernelGGL(accel_build_grid2, dim3((unsigned)((g2cells + 255) / 256)), dim3(256), 0, stream, a->nodes, a->grid2, a->grid2_vox, L2);
    }
    if ((rc = check_hip(hipGetLastError(), "accel build launch"))) return fail(rc);
    if ((rc = check_hip(hipStreamSynchronize(stream), "accel build"))) return fail(rc);

    a->view.nodes = a->nodes;
    a->view.rows = a->rows;
    a->view.grid = a->grid;
    a->view.grid_vox = a->grid_vox;
    a->view.grid_level = L;
    a->view.grid2 = a->grid2;
    a->view.grid2_vox = a->grid2_vox;
    a->view.grid2_level = L2;
    a->view.max_depth = max_depth;
    a->view.row_bytes = row_bytes;
    for (int i = 0; i < 3; ++i) {
        a->view.offset[i] = t->offset[i];
        a->view.scale[i] = t->scale[i];
    }
    a->view.data_dim = t->data_dim;
    a->view.basis_dim = t->basis_dim;
    a->view.format = t->format;
    a->view.capacity = t->capacity;
    a->bytes = (size_t)(nvox * 4 + nvox * row_bytes + gcells * 8 + g2cells * 8);
    return MNV_OK;
}

int mnv_accel_create_reserved(const mnv_tree_view *t, int64_t max_capacity, void *hip_stream, mnv_accel **out) {
    if (!t || !out) return set_error(MNV_E_INVALID, "null argument");
    if (max_capacity < t->capacity) return set_error(MNV_E_INVALID, "max_capacity is smaller than the tree");
    if (t->N != 2) return set_error(MNV_E_UNSUPPORTED, "accel needs N == 2");
    if (!t->data || !t->child || t->capacity < 1 || t->data_dim < 1) return set_error(MNV_E_INVALID, "invalid device tree view");
    const int b = (t->format == MNV_FORMAT_SH && t->basis_dim >= 0) ? t->basis_dim : -1;
    if (!(b == -1 || b == 1 || b == 4 || b == 9 || b == 16 || b == 25))
        return set_error(MNV_E_UNSUPPORTED, "accel supports RGBA and SH1/4/9/16/25 rows; use mnv_render_voxels for others");
    if (b >= 0 && t->data_dim != 3 * b + 1) return set_error(MNV_E_UNSUPPORTED, "accel needs data_dim == 3 * basis_dim + 1");
    if (b < 0 && t->data_dim != 4) return set_error(MNV_E_UNSUPPORTED, "accel needs data_dim == 4 for RGBA rows");
    hipStream_t stream = (hipStream_t)hip_stream;
    mnv_accel *a = new mnv_accel();
    int rc = MNV_OK;
    auto fail = [&](int code) {
        mnv_accel_destroy(a);
        return code;
    };
    if ((rc = check_hip(hipGetDevice(&a->device), "hipGetDevice"))) return fail(rc);
    hipDeviceProp_t prop;
    if ((rc = check_hip(hipGetDeviceProperties(&prop, a->device), "hipGetDeviceProperties"))) return fail(rc);
    a->num_cus = a->device_cus = prop.multiProcessorCount;

    const int row_bytes = row_bytes_for(b);
    a->reserved = max_capacity;
    if ((rc = check_hip(hipMalloc((void **)&a->nodes, max_capacity * 8 * 4), "hipMalloc(nodes)"))) return fail(rc);
    if ((rc = check_hip(hipMalloc((void **)&a->rows, max_capacity * 8 * row_bytes), "hipMalloc(rows)"))) return fail(rc);
    if ((rc = check_hip(hipMalloc((void **)&a->depth, max_capacity * 4), "hipMalloc(depth)"))) return fail(rc);
    if ((rc = check_hip(hipMalloc((void **)&a->flags, 32), "hipMalloc(flag)"))) return fail(rc);
    if ((rc = check_hip(hipMalloc((void **)&a->fault_dev, 4), "hipMalloc(fault)"))) return fail(rc);
    if ((rc = check_hip(hipMemsetAsync(a->fault_dev, 0, 4, stream), "memset fault"))) return fail(rc);
    if ((rc = check_hip(hipHostMalloc((void **)&a->fault_host, 4, hipHostMallocDefault), "hipHostMalloc(fault)"))) return fail(rc);
    *a->fault_host = 0u;
    if ((rc = check_hip(hipMalloc((void **)&a->slots_dev, (size_t)kSlots * kSlotBytes), "hipMalloc(slots)"))) return fail(rc);
    for (int i = 0; i < kSlots; ++i)
        if ((rc = check_hip(hipEventCreateWithFlags(&a->slot_done[i], hipEventDisableTiming), "hipEventCreate(slot)"))) return fail(rc);

    if ((rc = check_hip(hipMalloc((void **)&a->stats, 32 * sizeof(unsigned long long)), "hipMalloc(stats)"))) return fail(rc);
    if ((rc = check_hip(hipMemsetAsync(a->stats, 0, 32 * sizeof(unsigned long long), stream), "memset stats"))) return fail(rc);

    if ((rc = accel_build(a, t, stream))) return fail(rc);
    *out = a;
    return MNV_OK;
}

int mnv_accel_rebuild(mnv_accel *a, const mnv_tree_view *t, void *hip_stream) {
    if (!a || !t) return set_error(MNV_E_INVALID, "null argument");
    if (!t->data || !t->child || t->capacity < 1 || t->capacity > a->reserved)
        return set_error(MNV_E_INVALID, "invalid tree view, or the tree outgrew the reserved capacity");
    if (t->data_dim != a->view.data_dim || t->format != a->view.format || t->basis_dim != a->view.basis_dim)
        return set_error(MNV_E_INVALID, "tree view does not match the accel");
    return accel_build(a, t, (hipStream_t)hip_stream);
}

int mnv_accel_refresh(mnv_accel *a, const mnv_tree_view *t, int32_t old_capacity, const int32_t *changed_nodes, int32_t n_changed,
                      void *hip_stream) {
    if (!a || !t) return set_error(MNV_E_INVALID, "null argument");
    if (old_capacity != a->view.capacity) return set_error(MNV_E_INVALID, "old_capacity is not the capacity the accel was last built or refreshed for");
    if (t->capacity < old_capacity || t->capacity > a->reserved)
        return set_error(MNV_E_INVALID, "the tree shrank or outgrew the reserved capacity (rebuild with mnv_accel_create_reserved)");
    if (t->data_dim != a->view.data_dim || t->format != a->view.format || t->basis_dim != a->view.basis_dim || !t->data || !t->child)
        return set_error(MNV_E_INVALID, "tree view does not match the accel");
    if (t->capacity > old_capacity && !t->parent) return set_error(MNV_E_INVALID, "appended chunks need the parent array");
    if (n_changed < 0 || (n_changed > 0 && !changed_nodes)) return set_error(MNV_E_INVALID, "invalid changed_nodes");
    if (t->capacity == old_capacity && n_changed == 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc;
    const int b = (t->format == MNV_FORMAT_SH && t->basis_dim >= 0) ? t->basis_dim : -1;
    const int per_chan = b > 0 ? b : 1, chan_halfs = b > 0 ? chan_bytes_for(b) / 2 : 1, row_halfs = a->view.row_bytes / 2;
    const int grid_depth = a->view.grid_level;  // leaves this shallow sit in the small (LDS-staged) lookup grid
    // [1] deepest depth, [2] the small lookup grid is affected, [3] shallowest affected voxel, [4] / [5] patch items of the appended / changed voxels
    int32_t h[8] = {0, a->view.max_depth, 0, 127, 0, 0, 0, 0};
    if ((rc = check_hip(hipMemcpyAsync(a->flags, h, sizeof(h), hipMemcpyHostToDevice, stream), "refresh flags"))) return rc;
    const int32_t n_new = t->capacity - old_capacity;
    // the level-L2 grid: only the cells the affected voxels cover, cut into items (accel_patch_grid2).  The refresh kernels count every
    // voxel's items, accel_patch_plan scans the counts, the totals come back with the flags.
    const bool patch_changed = n_changed > 0 && t->parent != nullptr && a->view.grid2_level > 0;
    const bool patch_new = n_new > 0 && a->view.grid2_level > 0;
    uint32_t *prefix_new = nullptr, *prefix_changed = nullptr;
    if (patch_new || patch_changed) {
        const size_t words = (size_t)n_new + 1 + (size_t)(patch_changed ? n_changed : 0) + 1;
        if (a->patch_prefix_words < words) {
            if (a->patch_prefix) {
                if ((rc = check_hip(hipStreamSynchronize(stream), "accel refresh"))) return rc;
                (void)hipFree(a->patch_prefix);
                a->patch_prefix = nullptr;
                a->patch_prefix_words = 0;
            }
            if ((rc = check_hip(hipMalloc((void **)&a->patch_prefix, (words + words / 4) * 4), "hipMalloc(patch items)"))) return rc;
            a->patch_prefix_words = words + words / 4;
        }
        if (patch_new) prefix_new = a->patch_prefix;
        if (patch_changed) prefix_changed = a->patch_prefix + n_new + 1;
    }
    if (n_new > 0) {
        hipLaunchKernelGGL(accel_refresh_depth, dim3((n_new + 255) / 256), dim3(256), 0, stream, t->parent, a->depth, old_capacity, t->capacity, a->flags);
        const int64_t nv = (int64_t)n_new * 8;
        hipLaunchKernelGGL(accel_refresh_nodes, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, t->child, t->parent, t->data, a->depth, a->nodes,
                           old_capacity, t->capacity, t->data_dim, grid_depth, a->flags, prefix_new, a->view.grid2_level);
        hipLaunchKernelGGL(accel_pack_rows, dim3((unsigned)((nv * 3 + 255) / 256)), dim3(256), 0, stream, t->data + (int64_t)old_capacity * 8 * t->data_dim,
                           reinterpret_cast<uint16_t *>(a->rows) + (int64_t)old_capacity * 8 * row_halfs, nv, t->data_dim, per_chan, chan_halfs, row_halfs);
        if (patch_new) hipLaunchKernelGGL(accel_patch_plan, dim3(1), dim3(1024), 0, stream, n_new, prefix_new, a->flags + 4);
    }
    if (n_changed > 0) {
        hipLaunchKernelGGL(accel_refresh_changed, dim3((n_changed + 255) / 256), dim3(256), 0, stream, changed_nodes, n_changed, t->child, t->data, a->depth,
                           a->nodes, reinterpret_cast<uint16_t *>(a->rows), t->data_dim, per_chan, chan_halfs, row_halfs, grid_depth, a->flags, prefix_changed,
                           a->view.grid2_level);
        if (patch_changed) hipLaunchKernelGGL(accel_patch_plan, dim3(1), dim3(1024), 0, stream, n_changed, prefix_changed, a->flags + 5);
    }
    if ((rc = check_hip(hipMemcpyAsync(h, a->flags, sizeof(h), hipMemcpyDeviceToHost, stream), "read flags"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "accel refresh"))) return rc;
    if (h[1] > 23) return set_error(MNV_E_UNSUPPORTED, "accel supports trees up to depth 23; use mnv_render_voxels");
    if (h[2]) {  // an affected voxel is held by the small lookup grid: 32^3 cells at most, rebuilt whole
        const int64_t gcells = (int64_t)1 << (3 * a->view.grid_level);
        hipLaunchKernelGGL(accel_build_grid, dim3((unsigned)((gcells + 255) / 256)), dim3(256), 0, stream, a->nodes, a->grid, a->grid_vox, a->view.grid_level);
    }
    if (a->view.grid2_level > 0) {
        static const bool dbg = getenv("MNV_REFRESH_DEBUG") != nullptr;
        if (dbg)
            fprintf(stderr, "[mnv refresh] n_new %d n_changed %d shallowest %d grid2_level %d patch items %d + %d\n", n_new, n_changed, h[3], a->view.grid2_level, h[4], h[5]);
        if (h[4] > 0)
            hipLaunchKernelGGL(accel_patch_grid2, dim3((unsigned)h[4]), dim3(256), 0, stream, (const int32_t *)nullptr, old_capacity, n_new, prefix_new, t->parent,
                               a->depth, a->nodes, a->grid2, a->grid2_vox, a->view.grid2_level);
        if (h[5] > 0)
            hipLaunchKernelGGL(accel_patch_grid2, dim3((unsigned)h[5]), dim3(256), 0, stream, changed_nodes, 0, n_changed, prefix_changed, t->parent, a->depth,
                               a->nodes, a->grid2, a->grid2_vox, a->view.grid2_level);
        if (n_changed > 0 && !t->parent && h[3] <= a->view.grid2_level) {  // no parent array to walk up: the whole grid
            const int64_t g2cells = (int64_t)1 << (3 * a->view.grid2_level);
            hipLaunchKernelGGL(accel_build_grid2, dim3((unsigned)((g2cells + 255) / 256)), dim3(256), 0, stream, a->nodes, a->grid2, a->grid2_vox,
                               a->view.grid2_level);
        }
    }
    a->view.max_depth = std::max(a->view.max_depth, h[1]);
    a->view.capacity = t->capacity;
    return check_hip(hipGetLastError(), "accel refresh launch");
}

void mnv_accel_destroy(mnv_accel *a) {
    if (!a) return;
    if (a->stats && getenv("MNV_STATS")) {
        unsigned long long h[16];
        if (hipMemcpy(h, a->stats, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long ph[8] = {};
            if (hipMemcpy(ph, a->stats + 16, sizeof(ph), hipMemcpyDeviceToHost) == hipSuccess && ph[4] > 0)
                fprintf(stderr, "[mnv phases] share of the wavefronts' time (serialised by the stamps): lookup %.3f, step arithmetic + opacity %.3f, row wait %.3f, "
                                "colour arithmetic %.3f, other (refill, ray set-up, flush, loop) %.3f; wave-steps %llu\n",
                        (double)ph[0] / ph[4], (double)ph[1] / ph[4], (double)ph[2] / ph[4], (double)ph[3] / ph[4], 1.0 - (double)(ph[0] + ph[1] + ph[2] + ph[3]) / ph[4], ph[5]);
            const char *names[] = {"outer_iter", "refill", "march_step", "node_load", "dense", "colour_pass"};
            for (int i = 0; i < 6; ++i)
                fprintf(stderr, "[mnv stats] %-13s wave-level %llu lane-level %llu (%.1f lanes)\n", names[i], h[2 * i], h[2 * i + 1],
                        h[2 * i] ? (double)h[2 * i + 1] / (double)h[2 * i] : 0.0);
        }
    }
    if (a->timeline && getenv("MNV_TIMELINE")) {
        const size_t words = a->timeline_tiles * 4 + a->timeline_waves * 2;
        std::vector<unsigned long long> h(words + 3);
        h[0] = a->timeline_tiles;
        h[1] = a->timeline_waves;
        h[2] = a->timeline_tiles_per_frame;
        if (hipMemcpy(h.data() + 3, a->timeline, words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(getenv("MNV_TIMELINE"), "wb")) {
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
    }
    if (a->timeline) (void)hipFree(a->timeline);
    if (a->nodes_spare) (void)hipFree(a->nodes_spare);
    if (a->rows_spare) (void)hipFree(a->rows_spare);
    if (a->depth_spare) (void)hipFree(a->depth_spare);
    if (a->stats) (void)hipFree(a->stats);
    if (a->depth) (void)hipFree(a->depth);
    if (a->flags) (void)hipFree(a->flags);
    if (a->patch_prefix) (void)hipFree(a->patch_prefix);
    if (a->fault_dev) (void)hipFree(a->fault_dev);
    if (a->fault_host) (void)hipHostFree(a->fault_host);
    if (a->nodes) (void)hipFree(a->nodes);
    if (a->rows) (void)hipFree(a->rows);
    if (a->grid) (void)hipFree(a->grid);
    if (a->grid_vox) (void)hipFree(a->grid_vox);
    if (a->grid2) (void)hipFree(a->grid2);
    if (a->grid2_vox) (void)hipFree(a->grid2_vox);
    if (a->slots_dev) (void)hipFree(a->slots_dev);
    for (int i = 0; i < kSlots; ++i)
        if (a->slot_done[i]) (void)hipEventDestroy(a->slot_done[i]);

    delete a;
}

size_t mnv_accel_device_bytes(const mnv_accel *a) { return a ? a->bytes : 0; }
int32_t mnv_accel_grid2_level(const mnv_accel *a) { return a ? a->view.grid2_level : -1; }

int mnv_accel_set_cu_budget(mnv_accel *a, int32_t num_cus) {
    if (!a) return set_error(MNV_E_INVALID, "accel is null");
    if (num_cus > a->device_cus) return set_error(MNV_E_INVALID, "the device has fewer compute units");
    a->num_cus = num_cus <= 0 ? a->device_cus : num_cus;
    return MNV_OK;
}

int32_t mnv_partition_local_tiles(mnv_rect tile, mnv_partition part) { return partition_local_tiles(tile, part); }

void mnv_set_colour_math(int fast) { g_fast_colour.store(fast ? 1 : 0, std::memory_order_relaxed); }
void mnv_set_fused_kernel(int version) { g_fused_kernel.store(version == 1 || version == 2 ? version : 0, std::memory_order_relaxed); }
void mnv_set_fused_diag(unsigned long long *words32) { g_fused_diag.store(words32, std::memory_order_relaxed); }

int mnv_accel_fused_faults(const mnv_accel *accel, uint32_t *count_out) {
    if (!accel || !count_out) return set_error(MNV_E_INVALID, "null argument");
    uint32_t v = 0;
    const int rc = check_hip(hipMemcpy(&v, accel->fault_dev, 4, hipMemcpyDeviceToHost), "read fault word");  // waits for the device
    if (rc) return rc;
    *count_out = v;
    return MNV_OK;
}

int mnv_assemble_tiles(const void *gathered, void *frames, int32_t width, int32_t height, mnv_partition part, int32_t n_frames,
                       int32_t bytes_per_pixel, void *hip_stream) {
    if (!gathered || !frames || width < 1 || height < 1 || n_frames < 1 || part.world < 1 || part.tile_w < 8 || part.tile_h < 8 ||
        part.tile_w % 8 || part.tile_h % 8)
        return set_error(MNV_E_INVALID, "invalid tile-assembly arguments");
    if (bytes_per_pixel != 4 && bytes_per_pixel != 16) return set_error(MNV_E_UNSUPPORTED, "pixels are RGBA8 (4 bytes) or float RGBA (16 bytes)");
    const int32_t macros_x = (width + part.tile_w - 1) / part.tile_w, macros_y = (height + part.tile_h - 1) / part.tile_h;
    if (part.root_period < 0) return set_error(MNV_E_INVALID, "root_period must be 0 or >= 2");
    const int32_t period = root_period_of(part);
    const int32_t j_max = (int32_t)part_j_max((int64_t)macros_x * macros_y, part.world, period);
    hipStream_t stream = (hipStream_t)hip_stream;
    static const bool env_narrow = getenv("MNV_ASSEMBLE_NARROW") != nullptr;  // diagnostics: one RGBA8 pixel per thread
    if (bytes_per_pixel == 4 && width % 4 == 0 && !env_narrow) {
        // RGBA8: tile rows and frame rows are contiguous runs of pixels and tile_w is a multiple of 8, so the same index arithmetic
        // holds in units of four pixels: 16 bytes per thread instead of 4 (rank 0 runs this beside its march on the few compute
        // units the march leaves free)
        const int64_t n4 = (int64_t)(width / 4) * height * n_frames;
        hipLaunchKernelGGL(assemble_tiles_kernel<u32x4>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, static_cast<const u32x4 *>(gathered),
                           static_cast<u32x4 *>(frames), width / 4, height, part.tile_w / 4, part.tile_h, macros_x, j_max, part.world, period, n_frames);
        return check_hip(hipGetLastError(), "assemble_tiles_kernel");
    }
    const int64_t n = (int64_t)width * height * n_frames;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (bytes_per_pixel == 4)
        hipLaunchKernelGGL(assemble_tiles_kernel<uint32_t>, grid, block, 0, stream, static_cast<const uint32_t *>(gathered), static_cast<uint32_t *>(frames),
                           width, height, part.tile_w, part.tile_h, macros_x, j_max, part.world, period, n_frames);
    else
        hipLaunchKernelGGL(assemble_tiles_kernel<u32x4>, grid, block, 0, stream, static_cast<const u32x4 *>(gathered), static_cast<u32x4 *>(frames), width,
                           height, part.tile_w, part.tile_h, macros_x, j_max, part.world, period, n_frames);
    return check_hip(hipGetLastError(), "assemble_tiles_kernel");
}

int mnv_render_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                            mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return mnv_render_voxels_accel_batch(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, hip_stream);
}

int mnv_render_voxels_accel_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                 mnv_rect tile, mnv_partition part, float *rgba_out, uint8_t *rgba8_out,
                                 void *hip_stream) {
    return mnv_render_voxels_accel_batch(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, hip_stream);
}

static int render_accel(const mnv_accel *accel, const mnv_camera *cams, int32_t n_cams, const mnv_render_options *opt,
                        mnv_rect tile, mnv_partition part, float *rgba_out, uint8_t *rgba8_out, const AccelTrack *track,
                        void *hip_stream) {
    if (!accel) return set_error(MNV_E_INVALID, "accel is null");
    if (!cams || n_cams < 1 || n_cams > MNV_MAX_BATCH) return set_error(MNV_E_INVALID, "need 1 .. MNV_MAX_BATCH cameras");
    if (is_partitioned(part) && (part.rank < 0 || part.rank >= part.world || part.tile_w < 8 || part.tile_h < 8 ||
                           part.tile_w % 8 || part.tile_h % 8 || part.root_period < 0 || part.root_period == 1))
        return set_error(MNV_E_INVALID, "partition needs 0 <= rank < world, macro tiles that are multiples of 8 pixels and root_period 0 or >= 2");
    for (int i = 1; i < n_cams; ++i)
        if (cams[i].width != cams[0].width || cams[i].height != cams[0].height)
            return set_error(MNV_E_INVALID, "all cameras of a batch must have the same image size");
    FrameParams P;
    std::memset(&P, 0, sizeof(P));
    int rc = fill_params(P, &cams[0], opt, tile);
    if (rc) return rc;
    // pixel indices of a launch are 32 bits wide (frame f starts at f * pixels per frame)
    if ((uint64_t)(tile.w > 0 ? tile.w : 0) * (uint64_t)(tile.h > 0 ? tile.h : 0) * (uint64_t)n_cams > 0xffffffffull)
        return set_error(MNV_E_UNSUPPORTED, "more than 2^32 pixels in one launch: render fewer frames per call");
    std::memcpy(P.offset, accel->view.offset, sizeof(P.offset));
    std::memcpy(P.scale, accel->view.scale, sizeof(P.scale));
    P.rgba = rgba_out;
    P.rgba8 = rgba8_out;
    CamBlock blocks[MNV_MAX_BATCH];
    for (int i = 0; i < n_cams; ++i) {
        fill_camera(blocks[i], &cams[i]);
        fill_origin(blocks[i], P.offset, P.scale);
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    LaunchTimer timer(stream);
    rc = launch_accel(accel, P, blocks, n_cams, part, track, stream);
    if (rc == kUnsupportedBasis) return set_error(MNV_E_UNSUPPORTED, "unsupported basis_dim for the accel path");
    return check_hip((hipError_t)rc, "march_accel_kernel");
}

int mnv_render_voxels_accel_batch(const mnv_accel *accel, const mnv_camera *cams, int32_t n_cams,
                                  const mnv_render_options *opt, mnv_rect tile, mnv_partition part, float *rgba_out,
                                  uint8_t *rgba8_out, void *hip_stream) {
    return render_accel(accel, cams, n_cams, opt, tile, part, rgba_out, rgba8_out, nullptr, hip_stream);
}

int mnv_render_voxels_accel_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt,
                                  mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                  float *sample_track, const int16_t *sample_counts, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (!split_track && !sample_track)
        return mnv_render_voxels_accel(accel, cam, opt, tile, rgba_out, rgba8_out, hip_stream);
    const mnv_partition whole = {0, 1, 0, 0, 0};
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_render_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, float *rgba_out,
                                  uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                  const int32_t *parent, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (!visited) return mnv_render_voxels_accel_track(accel, cam, opt, tile, rgba_out, rgba8_out, split_track, sample_track, sample_counts, hip_stream);
    if (!parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    const mnv_partition whole = {0, 1, 0, 0, 0};
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, whole, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_render_voxels_accel_visit_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track, const int16_t *sample_counts,
                                       int32_t *visited, const int32_t *parent, void *hip_stream) {
    if (!opt) return set_error(MNV_E_INVALID, "options are null");
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (!split_track && !sample_track && !visited) return mnv_render_voxels_accel_part(accel, cam, opt, tile, part, rgba_out, rgba8_out, hip_stream);
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, &track, hip_stream);
}

int mnv_get_samples_from_voxels_accel(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                      float *split_track, float *sample_track, const int16_t *sample_counts, int16_t *num_samples,
                                      float *samples, int32_t samples_dim, int16_t *cluster_indices, const mnv_cluster_grid *grid,
                                      void *hip_stream) {
    return mnv_get_samples_from_voxels_accel_visit(accel, cam, opt, tile, split_track, sample_track, sample_counts, nullptr, nullptr, num_samples,
                                                   samples, samples_dim, cluster_indices, grid, hip_stream);
}

int mnv_get_samples_from_voxels_accel_visit(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                                            float *split_track, float *sample_track, const int16_t *sample_counts, int32_t *visited,
                                            const int32_t *parent, int16_t *num_samples, float *samples, int32_t samples_dim,
                                            int16_t *cluster_indices, const mnv_cluster_grid *grid, void *hip_stream) {
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (!opt || !num_samples || !samples || !cluster_indices || !grid) return set_error(MNV_E_INVALID, "null argument");
    const int need = 4 + (opt->need_viewdir ? 3 : 0) + (opt->appearance_embedding != -1 ? 1 : 0);
    if (samples_dim != need) return set_error(MNV_E_INVALID, "samples_dim must be 4 + 3 * need_viewdir + (appearance_embedding != -1)");
    if (opt->max_guided_samples < 1) return set_error(MNV_E_INVALID, "max_guided_samples must be positive");
    AccelTrack track = {};
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.num_samples = num_samples;
    track.samples = samples;
    track.cluster_indices = cluster_indices;
    track.max_guided_samples = opt->max_guided_samples;
    track.samples_dim = samples_dim;
    track.need_viewdir = opt->need_viewdir ? 1 : 0;
    track.appearance_embedding = opt->appearance_embedding;
    track.grid = grid;
    track.visited = visited;
    track.parent = parent;
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return render_accel(accel, cam, 1, opt, tile, whole, nullptr, nullptr, &track, hip_stream);
}

int mnv_render_guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                            const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, unsigned long long *sample_counter,
                            void *hip_stream) {
    return mnv_render_guided_fused_track(accel, cam, opt, tile, mlp, grid, rgba_out, rgba8_out, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         sample_counter, hip_stream);
}

static int guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                        const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                        float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                        unsigned long long *sample_counter, void *hip_stream);

int mnv_render_guided_fused_track(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, const mnv_mlp *mlp,
                                  const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                                  const int16_t *sample_counts, int32_t *visited, const int32_t *parent, unsigned long long *sample_counter,
                                  void *hip_stream) {
    const mnv_partition whole = {0, 1, 0, 0, 0};
    return guided_fused(accel, cam, opt, tile, whole, mlp, grid, rgba_out, rgba8_out, split_track, sample_track, sample_counts, visited, parent,
                        sample_counter, hip_stream);
}

int mnv_render_guided_fused_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                 const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out,
                                 unsigned long long *sample_counter, void *hip_stream) {
    return guided_fused(accel, cam, opt, tile, part, mlp, grid, rgba_out, rgba8_out, nullptr, nullptr, nullptr, nullptr, nullptr, sample_counter,
                        hip_stream);
}

int mnv_render_guided_fused_track_part(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                                       const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                                       float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                                       unsigned long long *sample_counter, void *hip_stream) {
    return guided_fused(accel, cam, opt, tile, part, mlp, grid, rgba_out, rgba8_out, split_track, sample_track, sample_counts, visited, parent,
                        sample_counter, hip_stream);
}

static int guided_fused(const mnv_accel *accel, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile, mnv_partition part,
                        const mnv_mlp *mlp, const mnv_cluster_grid *grid, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                        float *sample_track, const int16_t *sample_counts, int32_t *visited, const int32_t *parent,
                        unsigned long long *sample_counter, void *hip_stream) {
    if (!accel || !cam || !opt || !mlp || !grid) return set_error(MNV_E_INVALID, "null argument");
    {
        // A spin-wait that the watchdog of guided_fused2_kernel abandoned leaves wrong pixels behind.  The launches are asynchronous, so the
        // frame itself cannot answer for it: the NEXT call on this accel does, once per fault, and mnv_accel_fused_faults reads the count.
        mnv_accel *mut = const_cast<mnv_accel *>(accel);
        std::lock_guard<std::mutex> g(mut->launch_mutex);
        const uint32_t seen = *static_cast<volatile uint32_t *>(mut->fault_host);
        if (seen != mut->fault_reported) {
            const uint32_t n = seen - mut->fault_reported;
            mut->fault_reported = seen;
            fprintf(stderr, "libmnv: %u wavefront(s) of an earlier fused guided-sampling frame on this accel abandoned a spin-wait (watchdog): that frame is wrong\n", n);
            return set_error(MNV_E_FAULT, "an earlier fused guided-sampling frame on this accel ran into the kernel's watchdog and is wrong; mnv_set_fused_kernel(1) selects the kernel without spin-waits");
        }
    }
    if (visited && !parent) return set_error(MNV_E_INVALID, "visit marks on the packed layout need the parent array (the ancestors of the marked chunks)");
    if (opt->render_depth) return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame has no depth mode; use the four-step path");
    if (opt->max_guided_samples < 1) return set_error(MNV_E_INVALID, "max_guided_samples must be positive");
    const MlpShape &S = mlp->shape;
    if (S.hidden_width != 64 || S.nkk0 > 2)
        return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame runs 64-wide networks with at most 64 encoded inputs; use the four-step path");
    if (S.out_dim != accel->view.data_dim + 1) return set_error(MNV_E_INVALID, "the model's out_dim must be the tree's data_dim + 1 (cuda_renderer.cpp:255-257)");
    if ((S.need_viewdir != 0) != (opt->need_viewdir != 0)) return set_error(MNV_E_INVALID, "options.need_viewdir does not match the model");
    if (S.n_embeddings > 0 && opt->appearance_embedding == -1) return set_error(MNV_E_INVALID, "the model needs an appearance embedding");
    const int b = (accel->view.format == MNV_FORMAT_SH && accel->view.basis_dim >= 0) ? accel->view.basis_dim : -1;
    if (!(b == -1 || b == 1 || b == 4 || b == 9 || b == 16))
        return set_error(MNV_E_UNSUPPORTED, "the fused guided-sampling frame supports RGBA and SH1/4/9/16 trees; use the four-step path");
    FusedGuided F;
    std::memset(static_cast<void *>(&F), 0, sizeof(F));
    F.S = S;
    F.frags = mlp->frags;
    F.biases = mlp->biases;
    F.embeddings = mlp->embeddings;
    for (int i = 0; i < 2; ++i) F.grid_dim[i] = grid->grid_dim[i];
    for (int i = 0; i < 3; ++i) {
        F.min_position[i] = grid->min_position[i];
        F.range[i] = grid->range[i];
    }
    F.max_guided_samples = opt->max_guided_samples;
    F.appearance_embedding = opt->appearance_embedding;
    static const int env_batch = getenv("MNV_FUSED_BATCH_MIN") ? atoi(getenv("MNV_FUSED_BATCH_MIN")) : kFW;
    F.batch_min = env_batch < 1 ? 1 : (env_batch > kFW ? kFW : env_batch);
    F.sample_counter = sample_counter;
    F.diag = g_fused_diag.load(std::memory_order_relaxed);
    static const int env_switch = getenv("MNV_F2_SWITCH_MIN") ? atoi(getenv("MNV_F2_SWITCH_MIN")) : 32;
    F.switch_min = env_switch;
    AccelTrack track = {};
    track.fused = &F;
    track.split_track = split_track;
    track.sample_track = sample_track;
    track.sample_counts = sample_counts;
    track.max_depth = opt->max_depth;
    track.max_sample_count = opt->max_sample_count;
    track.visited = visited;
    track.parent = parent;
    return render_accel(accel, cam, 1, opt, tile, part, rgba_out, rgba8_out, &track, hip_stream);
}

}  // extern "C"
