// mnv_accel_refresh_prune.hip -- the packed layout follows the tree edits of the refinement loop in place: mnv_accel_refresh patches
// appended chunks, rewritten rows and exactly the lookup cells they cover; accel_apply_prune (called by mnv_prune_tree_accel) renumbers.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mnv_accel_launch.h"
#include "mnv_knobs.h"

namespace mnv {

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
__device__ __forceinline__ uint32_t patch_items(int32_t d, int32_t L2, int32_t L2i);

__global__ void accel_refresh_nodes(const int32_t *child, const int32_t *parent, const uint16_t *data, const int32_t *depth, uint32_t *nodes,
                                    int32_t first, int32_t capacity, int32_t data_dim, int32_t grid_depth, int32_t *flags, uint32_t *items, int32_t L2,
                                    int32_t L2i) {
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
        if (items) items[c - first] = patch_items(depth[pv >> 3], L2, L2i);  // (accel_patch_plan turns the counts into first items)
    }
}

// existing leaves whose data row was rewritten (mnv_apply_sample_results): sigma in the node word, colour row
__global__ void accel_refresh_changed(const int32_t *changed_nodes, int32_t n, const int32_t *child, const uint16_t *data, const int32_t *depth,
                                      uint32_t *nodes, uint16_t *rows, int32_t data_dim, int32_t per_chan, int32_t chan_halfs,
                                      int32_t row_halfs, int32_t grid_depth, int32_t *flags, uint32_t *items, int32_t L2, int32_t L2i) {
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
    rows[v * row_halfs + 3 * chan_halfs] = data[v * data_dim + data_dim - 1];  // sigma in the half behind the channel blocks (accel_pack_rows)
    if (depth[c] <= grid_depth) flags[2] = 1;
    atomicMin(&flags[3], depth[c]);
    if (items) items[i] = patch_items(depth[c], L2, L2i);
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

// L2i = L2 + 1 when the grid carries inline cell words (grid2i): a voxel one level below the grid is described by the word of the ONE cell
// above it (its chunk's eight-leaves flag and sigma mask), which is then an item too
__device__ __forceinline__ uint32_t patch_items(int32_t d, int32_t L2, int32_t L2i) {
    if (d >= 1 && d <= L2) return (uint32_t)((((uint64_t)1 << (3 * (L2 - d))) + kPatchCells - 1) / kPatchCells);
    return d == L2 + 1 && L2i > L2 ? 1u : 0u;
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
                                                         const int32_t *depth, const uint32_t *nodes, uint32_t *grid2, uint32_t *grid2_vox, uint32_t *grid2i,
                                                         int32_t L2, uint32_t inline_base) {
    __shared__ uint32_t s_box[6];  // x, y, z at the voxel's own level; its depth; the voxel; the item's number among the voxel's items
    if (threadIdx.x == 0) {
        int32_t lo = 0, hi = n;  // the last voxel whose first item is <= this one (voxels without cells have no items and are never met)
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        int64_t pv = patch_voxel(vox_pairs, first_chunk, parent, lo);
        int32_t cur = (int32_t)(pv >> 3);
        int32_t d = depth[cur];
        if (d == L2 + 1) {  // a voxel one level below the grid (it has an item only when grid2i exists): the cell of the voxel above it
            pv = (int64_t)parent[cur];
            cur = (int32_t)(pv >> 3);
            d = L2;
        }
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
        if (grid2i) grid2i[o] = inline_cell_word(nodes, word, inline_base);
    }
}

// Brick records after a refinement step: entry v of the records describes voxel v of a depth-(L2 + 1) chunk and the eight voxels below it, so
// an affected voxel of depth L2 + 1 rewrites its own entry, one of depth L2 + 2 the entry of the voxel above it, and a chunk appended at
// depth L2 + 1 gets its eight entries.  One thread per affected voxel (vox_pairs, or the parent voxel of chunk first_chunk + b); threads
// that meet in an entry write the same value (every node word is final when this runs).
__global__ void accel_patch_recs(const int32_t *vox_pairs, int32_t first_chunk, int32_t n, const int32_t *parent, const int32_t *depth, const uint32_t *nodes,
                                 uint2 *recs, int32_t L2) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int64_t pv = patch_voxel(vox_pairs, first_chunk, parent, b);
    const int32_t c = (int32_t)(pv >> 3);
    if (!vox_pairs) {
        const int32_t nc = first_chunk + b;
        if (depth[nc] == L2 + 1)
            for (int s = 0; s < 8; ++s) recs[(int64_t)nc * 8 + s] = brick_record_entry(nodes, (int64_t)nc * 8 + s);
        if (depth[nc] == 0) return;  // not linked
    }
    const int32_t d = depth[c];
    if (d == L2 + 1) {
        recs[pv] = brick_record_entry(nodes, pv);
    } else if (d == L2 + 2) {
        const int64_t ppv = (int64_t)parent[c];
        recs[ppv] = brick_record_entry(nodes, ppv);
    }
}

// Test-hook builds (MNV_REFRESH_DEBUG=2): every lookup word the edits patched against a fresh derivation from the node words.
// bad[0..4]: wrong grid2 words, grid2_vox words, grid2i words, record entries, small-grid words
__global__ void accel_verify_lookup(const uint32_t *nodes, const int32_t *depth, const uint32_t *grid, const uint32_t *grid_vox, int32_t L,
                                    const uint32_t *grid2, const uint32_t *grid2_vox, const uint32_t *grid2i, const uint2 *recs, int32_t L2,
                                    int32_t capacity, uint32_t inline_base, int32_t *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cells2 = L2 > 0 ? (int64_t)1 << (3 * L2) : 0, cells1 = (int64_t)1 << (3 * L);
    auto walk = [&](uint32_t ix, uint32_t iy, uint32_t iz, int32_t levels, uint32_t &word, uint32_t &vox) {
        uint32_t chunk = 0;
        word = vox = 0;
        for (int32_t l = 1; l <= levels; ++l) {
            const int32_t s = levels - l;
            vox = chunk * 8u + ((((ix >> s) & 1u) << 2) | (((iy >> s) & 1u) << 1) | ((iz >> s) & 1u));
            word = nodes[vox];
            if (word & kLeafBit) break;
            chunk = word;
        }
    };
    if (i < cells2) {
        const uint32_t G = 1u << L2, iz = (uint32_t)i & (G - 1), iy = ((uint32_t)i >> L2) & (G - 1), ix = (uint32_t)(i >> (2 * L2));
        uint32_t word, vox;
        walk(ix, iy, iz, L2, word, vox);
        const uint32_t o = grid2_index(ix, iy, iz, L2);
        if (grid2[o] != word) atomicAdd(&bad[0], 1);
        if ((word & kLeafBit) && grid2_vox[o] != vox) atomicAdd(&bad[1], 1);
        if (grid2i && grid2i[o] != inline_cell_word(nodes, word, inline_base)) atomicAdd(&bad[2], 1);
    }
    if (i < cells1) {
        const uint32_t G = 1u << L, iz = (uint32_t)i & (G - 1), iy = ((uint32_t)i >> L) & (G - 1), ix = (uint32_t)(i >> (2 * L));
        uint32_t word, vox;
        walk(ix, iy, iz, L, word, vox);
        if (grid[i] != word || ((word & kLeafBit) && grid_vox[i] != vox)) atomicAdd(&bad[4], 1);
    }
    if (recs && i < (int64_t)capacity * 8 && depth[i >> 3] == L2 + 1) {
        const uint2 want = brick_record_entry(nodes, i), got = recs[i];
        if (want.x != got.x || want.y != got.y) atomicAdd(&bad[3], 1);
    }
}

int verify_lookup(mnv_accel *a, hipStream_t stream, const char *where) {
    int32_t *bad = nullptr, h[5] = {};
    int rc;
    if ((rc = check_hip(hipMalloc((void **)&bad, sizeof(h)), "hipMalloc(verify)"))) return rc;
    (void)hipMemsetAsync(bad, 0, sizeof(h), stream);
    const int L2 = a->view.grid2_level;
    int64_t n = std::max<int64_t>((int64_t)1 << (3 * a->view.grid_level), (int64_t)a->view.capacity * 8);
    if (L2 > 0) n = std::max<int64_t>(n, (int64_t)1 << (3 * L2));
    hipLaunchKernelGGL(accel_verify_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a->nodes, a->depth, a->grid, a->grid_vox, a->view.grid_level,
                       a->grid2, a->grid2_vox, a->view.grid2i, a->view.recs, L2, a->view.capacity, a->view.inline_base, bad);
    rc = check_hip(hipMemcpyAsync(h, bad, sizeof(h), hipMemcpyDeviceToHost, stream), "verify read");
    if (!rc) rc = check_hip(hipStreamSynchronize(stream), "verify");
    (void)hipFree(bad);
    if (rc) return rc;
    if (h[0] | h[1] | h[2] | h[3] | h[4]) {
        fprintf(stderr, "[mnv verify] %s: patched lookup words differ from a fresh derivation: grid2 %d, grid2_vox %d, grid2i %d, records %d, grid %d\n", where,
                h[0], h[1], h[2], h[3], h[4]);
        return set_error(MNV_E_FAULT, "patched lookup words differ from a fresh derivation (refresh-debug knob of the test-hook build)");
    }
    return MNV_OK;
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

// Called by mnv_prune_tree_accel between its scan and its fix-up / compaction (old numbering everywhere).
int accel_apply_prune(mnv_accel *a, const int32_t *parent, const uint16_t *data, int32_t data_dim, const uint8_t *to_delete, const int32_t *shifts,
                      int32_t old_capacity, int32_t n_deleted, hipStream_t stream) {
    if (!a || old_capacity != a->view.capacity) return set_error(MNV_E_INVALID, "the accel does not describe the tree that is being pruned");
    std::lock_guard<std::mutex> view_lock(a->launch_mutex);  // launch_accel copies the view under the same lock
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
    // inline cell words and brick records: derived again from the renumbered node words and the patched grid, on the same stream (a pass over
    // the grid like accel_prune_grid's own; prunes are rare)
    if (a->view.grid2i) {
        // (the chunks were renumbered: the 22-bit chunk field is re-based on the smallest number of depth L2 + 1 there is now)
        uint32_t base = 0;
        if ((rc = min_chunk_of_depth(a->depth, a->view.capacity, a->view.grid2_level + 1, a->flags + 6, stream, &base))) return rc;
        a->view.inline_base = base;
        launch_build_grid2i(a->nodes, a->grid2, a->grid2i, a->view.grid2_level, base, stream);
    }
    if (a->view.recs) launch_build_recs(a->nodes, a->depth, a->recs, a->view.capacity, a->view.grid2_level, stream);
    if ((rc = check_hip(hipGetLastError(), "accel prune launch"))) return rc;
    static const int dbg = knob_int(KNOB_REFRESH_DEBUG, 0);
    if (dbg >= 2) return verify_lookup(a, stream, "prune");
    return MNV_OK;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

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
    std::lock_guard<std::mutex> view_lock(a->launch_mutex);  // launch_accel copies the view under the same lock
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc;
    const int b = (t->format == MNV_FORMAT_SH && t->basis_dim >= 0) ? t->basis_dim : -1;
    const int per_chan = b > 0 ? b : 1, chan_halfs = chan_bytes_for(b) / 2, row_halfs = a->view.row_bytes / 2;
    const int grid_depth = a->view.grid_level;  // leaves this shallow sit in the small (LDS-staged) lookup grid
    // [1] deepest depth, [2] the small lookup grid is affected, [3] shallowest affected voxel, [4] / [5] patch items of the appended / changed voxels
    int32_t h[8] = {0, a->view.max_depth, 0, 127, 0, 0, 0, 0};
    if ((rc = check_hip(hipMemcpyAsync(a->flags, h, sizeof(h), hipMemcpyHostToDevice, stream), "refresh flags"))) return rc;
    const int32_t n_new = t->capacity - old_capacity;
    // the level-L2 grid: only the cells the affected voxels cover, cut into items (accel_patch_grid2).  The refresh kernels count every
    // voxel's items, accel_patch_plan scans the counts, the totals come back with the flags.
    const int32_t L2 = a->view.grid2_level, L2i = a->view.grid2i ? L2 + 1 : L2;
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
                           old_capacity, t->capacity, t->data_dim, grid_depth, a->flags, prefix_new, L2, L2i);
        launch_pack_rows(t->data + (int64_t)old_capacity * 8 * t->data_dim, reinterpret_cast<uint16_t *>(a->rows) + (int64_t)old_capacity * 8 * row_halfs, nv,
                         t->data_dim, per_chan, chan_halfs, row_halfs, stream);
        if (patch_new) hipLaunchKernelGGL(accel_patch_plan, dim3(1), dim3(1024), 0, stream, n_new, prefix_new, a->flags + 4);
    }
    if (n_changed > 0) {
        hipLaunchKernelGGL(accel_refresh_changed, dim3((n_changed + 255) / 256), dim3(256), 0, stream, changed_nodes, n_changed, t->child, t->data, a->depth,
                           a->nodes, reinterpret_cast<uint16_t *>(a->rows), t->data_dim, per_chan, chan_halfs, row_halfs, grid_depth, a->flags, prefix_changed,
                           L2, L2i);
        if (patch_changed) hipLaunchKernelGGL(accel_patch_plan, dim3(1), dim3(1024), 0, stream, n_changed, prefix_changed, a->flags + 5);
    }
    if ((rc = check_hip(hipMemcpyAsync(h, a->flags, sizeof(h), hipMemcpyDeviceToHost, stream), "read flags"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "accel refresh"))) return rc;
    if (h[1] > 23) {
        // nodes, rows and depths are patched, the lookup grids are not: no frame may read them as they are (mnv_accel_rebuild refuses the tree too)
        a->view.grid2i = nullptr;
        a->view.recs = nullptr;
        a->view.capacity = t->capacity;
        return set_error(MNV_E_UNSUPPORTED, "accel supports trees up to depth 23; use mnv_render_voxels");
    }
    if (h[2]) {  // an affected voxel is held by the small lookup grid: 32^3 cells at most, rebuilt whole
        launch_build_grid(a->nodes, a->grid, a->grid_vox, a->view.grid_level, stream);
    }
    if (a->view.grid2_level > 0) {
        static const int dbg = knob_int(KNOB_REFRESH_DEBUG, 0);
        if (dbg)
            fprintf(stderr, "[mnv refresh] n_new %d n_changed %d shallowest %d grid2_level %d patch items %d + %d\n", n_new, n_changed, h[3], a->view.grid2_level, h[4], h[5]);
        if (h[4] > 0)
            hipLaunchKernelGGL(accel_patch_grid2, dim3((unsigned)h[4]), dim3(256), 0, stream, (const int32_t *)nullptr, old_capacity, n_new, prefix_new, t->parent,
                               a->depth, a->nodes, a->grid2, a->grid2_vox, const_cast<uint32_t *>(a->view.grid2i), a->view.grid2_level, a->view.inline_base);
        if (h[5] > 0)
            hipLaunchKernelGGL(accel_patch_grid2, dim3((unsigned)h[5]), dim3(256), 0, stream, changed_nodes, 0, n_changed, prefix_changed, t->parent, a->depth,
                               a->nodes, a->grid2, a->grid2_vox, const_cast<uint32_t *>(a->view.grid2i), a->view.grid2_level, a->view.inline_base);
        if (n_changed > 0 && !t->parent && h[3] <= L2i) {  // no parent array to walk up: the whole grid
            if (h[3] <= L2) launch_build_grid2(a->nodes, a->grid2, a->grid2_vox, a->view.grid2_level, stream);
            if (a->view.grid2i) launch_build_grid2i(a->nodes, a->grid2, a->grid2i, a->view.grid2_level, a->view.inline_base, stream);
        }
    }
    a->view.max_depth = std::max(a->view.max_depth, h[1]);
    a->view.capacity = t->capacity;
    // brick records: the entries the edit touches -- or, for a tree that only now reaches two levels below the grid, all of them
    static const int env_bricks = knob_int(KNOB_BRICK_LEVELS, 3);
    if (a->view.grid2i && (env_bricks & 2) && a->view.max_depth >= L2 + 2) {
        if (!a->view.recs) {
            if (!a->recs) {
                if ((rc = check_hip(hipMalloc((void **)&a->recs, (size_t)a->reserved * kRecWords * 4), "hipMalloc(brick records)"))) return rc;
                a->bytes += (size_t)a->reserved * kRecWords * 4;
            }
            launch_build_recs(a->nodes, a->depth, a->recs, t->capacity, L2, stream);
            a->view.recs = a->recs;
        } else if (n_changed > 0 && !t->parent) {
            launch_build_recs(a->nodes, a->depth, a->recs, t->capacity, L2, stream);
        } else {
            if (n_new > 0)
                hipLaunchKernelGGL(accel_patch_recs, dim3((n_new + 255) / 256), dim3(256), 0, stream, (const int32_t *)nullptr, old_capacity, n_new, t->parent, a->depth,
                                   a->nodes, a->recs, L2);
            if (n_changed > 0)
                hipLaunchKernelGGL(accel_patch_recs, dim3((n_changed + 255) / 256), dim3(256), 0, stream, changed_nodes, 0, n_changed, t->parent, a->depth, a->nodes,
                                   a->recs, L2);
        }
    }
    if ((rc = check_hip(hipGetLastError(), "accel refresh launch"))) return rc;
    static const int dbg_verify = knob_int(KNOB_REFRESH_DEBUG, 0);
    if (dbg_verify >= 2) return verify_lookup(a, stream, "refresh");
    return MNV_OK;
}

}  // extern "C"
