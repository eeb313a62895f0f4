// mnv_accel_build.hip -- the packed layout ("accel", mnv_accel.h) is derived from the reference's arrays here: chunk depths, node words,
// colour rows, the two lookup grids; mnv_accel_create / _rebuild / _destroy.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mnv_accel_launch.h"
#include "mnv_knobs.h"

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
    if (c == 2) {
        // the half behind the three channel blocks carries the voxel's sigma (every format has it spare): a dense sample found through a
        // brick record reads it with its colours instead of a node word
        rows[v * row_halfs + 3 * chan_halfs] = data[v * data_dim + data_dim - 1];
        for (int32_t k = 3 * chan_halfs + 1; k < row_halfs; ++k) rows[v * row_halfs + k] = 0;
    }
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

// grid2i = grid2 with the last level folded into the words of the cells whose chunk holds eight leaves (mnv_accel.h); one thread per cell
__global__ void accel_build_grid2i(const uint32_t *nodes, const uint32_t *grid2, uint32_t *grid2i, int64_t cells, uint32_t inline_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    grid2i[i] = inline_cell_word(nodes, grid2[i], inline_base);
}

__global__ void accel_min_chunk_of_depth(const int32_t *depth, int32_t capacity, int32_t d, int32_t *out) {
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < capacity && depth[c] == d) atomicMin(out, c);
}

int min_chunk_of_depth(const int32_t *depth, int32_t capacity, int32_t d, int32_t *scratch, hipStream_t stream, uint32_t *out) {
    int rc;
    int32_t h = capacity;
    if ((rc = check_hip(hipMemcpyAsync(scratch, &h, 4, hipMemcpyHostToDevice, stream), "seed min chunk"))) return rc;
    hipLaunchKernelGGL(accel_min_chunk_of_depth, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, stream, depth, capacity, d, scratch);
    if ((rc = check_hip(hipMemcpyAsync(&h, scratch, 4, hipMemcpyDeviceToHost, stream), "read min chunk"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "min chunk of depth"))) return rc;
    *out = (uint32_t)h;
    return MNV_OK;
}

// what the inline words and records cover (mnv_accel_lookup_coverage): [0] non-leaf cells of the second grid, [1] of them inline, [2] of them NOT
// inline although their chunk holds eight leaves (its number lies outside the 22-bit field), [3] chunks of depth L2 + 1 (= brick records in use)
__global__ void accel_count_coverage(const uint32_t *nodes, const uint32_t *grid2, const uint32_t *grid2i, int64_t cells, unsigned long long *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const uint32_t w = grid2[i];
    if (w & kLeafBit) return;
    // (a chunk is named by exactly one cell of level L2: [0] also counts the chunks of depth L2 + 1)
    atomicAdd(&out[0], 1ull);
    if (grid2i && (grid2i[i] & kInlineBit)) {
        atomicAdd(&out[1], 1ull);
    } else {
        bool leaves = true;
        for (int s = 0; s < 8; ++s) leaves = leaves && (nodes[(int64_t)w * 8 + s] & kLeafBit) != 0u;
        if (leaves) atomicAdd(&out[2], 1ull);
    }
}

// brick record of every chunk c of depth L2 + 1 (layout: mnv_accel.h); one thread per (chunk, voxel s1)
__global__ void accel_build_recs(const uint32_t *nodes, const int32_t *depth, uint2 *recs, int32_t capacity, int32_t L2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)capacity * 8) return;
    if (depth[i >> 3] != L2 + 1) return;
    recs[i] = brick_record_entry(nodes, i);
}

void launch_build_recs(const uint32_t *nodes, const int32_t *depth, uint2 *recs, int32_t capacity, int32_t L2, hipStream_t stream) {
    const int64_t n = (int64_t)capacity * 8;
    hipLaunchKernelGGL(accel_build_recs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, nodes, depth, recs, capacity, L2);
}

void launch_build_grid2i(const uint32_t *nodes, const uint32_t *grid2, uint32_t *grid2i, int32_t L2, uint32_t inline_base, hipStream_t stream) {
    const int64_t cells = (int64_t)1 << (3 * L2);
    hipLaunchKernelGGL(accel_build_grid2i, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, nodes, grid2, grid2i, cells, inline_base);
}

void launch_pack_rows(const uint16_t *data, uint16_t *rows, int64_t nvox, int32_t data_dim, int32_t per_chan, int32_t chan_halfs, int32_t row_halfs,
                      hipStream_t stream) {
    hipLaunchKernelGGL(accel_pack_rows, dim3((unsigned)((nvox * 3 + 255) / 256)), dim3(256), 0, stream, data, rows, nvox, data_dim, per_chan, chan_halfs, row_halfs);
}
void launch_build_grid(const uint32_t *nodes, uint32_t *grid, uint32_t *grid_vox, int32_t L, hipStream_t stream) {
    const int64_t cells = (int64_t)1 << (3 * L);
    hipLaunchKernelGGL(accel_build_grid, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, nodes, grid, grid_vox, L);
}
void launch_build_grid2(const uint32_t *nodes, uint32_t *grid2, uint32_t *grid2_vox, int32_t L2, hipStream_t stream) {
    const int64_t cells = (int64_t)1 << (3 * L2);
    hipLaunchKernelGGL(accel_build_grid2, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, nodes, grid2, grid2_vox, L2);
}

// (Re)build every derived array of `a` from the tree: chunk depths, node words, colour rows, lookup grids.  The big arrays
// (nodes, rows, depth) are sized for a->reserved chunks and kept; the grids are reallocated only when their level changes.
int accel_build(mnv_accel *a, const mnv_tree_view *t, hipStream_t stream) {
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
                       reinterpret_cast<uint16_t *>(a->rows), nvox, t->data_dim, b > 0 ? b : 1, chan_bytes_for(b) / 2,
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
    static const int env_l2 = knob_int(KNOB_GRID2_LEVEL, -1);
    if (env_l2 >= 0 && env_l2 <= kMaxGrid2Level && env_l2 < max_depth) L2 = env_l2;
    if (L2 <= L || L2 < 2) L2 = 0;
    int64_t g2cells = 0;
    if (a->grid2 && a->view.grid2_level != L2) {
        (void)hipFree(a->grid2);
        (void)hipFree(a->grid2_vox);
        a->grid2 = a->grid2_vox = nullptr;
        if (a->grid2i) (void)hipFree(a->grid2i);
        a->grid2i = nullptr;
    }
    if (L2 > 0) {
        g2cells = (int64_t)1 << (3 * L2);
        if (!a->grid2) {
            if ((rc = check_hip(hipMalloc((void **)&a->grid2, g2cells * 4), "hipMalloc(grid2)"))) return fail(rc);
            if ((rc = check_hip(hipMalloc((void **)&a->grid2_vox, g2cells * 4), "hipMalloc(grid2_vox)"))) return fail(rc);
        }
        hipLaunchKernelGGL(accel_build_grid2, dim3((unsigned)((g2cells + 255) / 256)), dim3(256), 0, stream, a->nodes, a->grid2, a->grid2_vox, L2);
    }
    // What frames read instead of node words (mnv_accel_refresh patches both with the tree edit, a prune derives them again):
    //  * grid2i: grid2 with the last level folded into the words of the cells whose chunk holds eight leaves -- a depth-10 tree (cfg2, L2 = 9)
    //    then never reads a node word: LDS grid -> grid2i [-> row];
    //  * brick records: the two levels below the grid from one 64-byte record per depth-(L2 + 1) chunk, for trees that have two such levels
    //    (depth >= L2 + 2; cfg3): LDS grid -> grid2i -> record [-> row] instead of LDS grid -> grid2 -> node -> node [-> row].
    static const int env_bricks = knob_int(KNOB_BRICK_LEVELS, 3);  // bit 0: inline cell words, bit 1: records
    const bool want_inline = (env_bricks & 1) && L2 > 0, want_recs = (env_bricks & 2) && want_inline && max_depth >= L2 + 2;
    if (want_inline) {
        if (!a->grid2i && (rc = check_hip(hipMalloc((void **)&a->grid2i, g2cells * 4), "hipMalloc(grid2i)"))) return fail(rc);
        uint32_t base = 0;
        if ((rc = min_chunk_of_depth(depth, t->capacity, L2 + 1, a->flags + 6, stream, &base))) return fail(rc);
        a->view.inline_base = base;
        launch_build_grid2i(a->nodes, a->grid2, a->grid2i, L2, base, stream);
    } else if (a->grid2i) {
        (void)hipFree(a->grid2i);
        a->grid2i = nullptr;
    }
    if (want_recs) {
        if (!a->recs && (rc = check_hip(hipMalloc((void **)&a->recs, (size_t)max_capacity * kRecWords * 4), "hipMalloc(brick records)"))) return fail(rc);
        launch_build_recs(a->nodes, depth, a->recs, t->capacity, L2, stream);
    } else if (a->recs) {
        (void)hipFree(a->recs);
        a->recs = nullptr;
    }
    if ((rc = check_hip(hipGetLastError(), "accel build launch"))) return fail(rc);
    if ((rc = check_hip(hipStreamSynchronize(stream), "accel build"))) return fail(rc);
    a->view.grid2i = want_inline ? a->grid2i : nullptr;
    a->view.recs = want_recs ? a->recs : nullptr;

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
    a->view.sigma_off = 3 * chan_bytes_for(b);
    for (int i = 0; i < 3; ++i) {
        a->view.offset[i] = t->offset[i];
        a->view.scale[i] = t->scale[i];
    }
    a->view.data_dim = t->data_dim;
    a->view.basis_dim = t->basis_dim;
    a->view.format = t->format;
    a->view.capacity = t->capacity;
    a->bytes = (size_t)(nvox * 4 + nvox * row_bytes + gcells * 8 + g2cells * 8) + (a->recs ? (size_t)max_capacity * kRecWords * 4 : 0) + (a->grid2i ? (size_t)g2cells * 4 : 0);
    return MNV_OK;
}

}  // namespace mnv

using namespace mnv;

extern "C" {

int mnv_accel_create(const mnv_tree_view *t, void *hip_stream, mnv_accel **out) {
    return mnv_accel_create_reserved(t, t ? t->capacity : 0, hip_stream, out);
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
    std::lock_guard<std::mutex> view_lock(a->launch_mutex);  // launch_accel copies the view under the same lock
    return accel_build(a, t, (hipStream_t)hip_stream);
}

void mnv_accel_destroy(mnv_accel *a) {
    if (!a) return;
    if (a->stats && knob_set(KNOB_STATS)) {
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
    if (a->timeline && knob_str(KNOB_TIMELINE)) {
        const size_t words = a->timeline_tiles * 4 + a->timeline_waves * 2;
        std::vector<unsigned long long> h(words + 3);
        h[0] = a->timeline_tiles;
        h[1] = a->timeline_waves;
        h[2] = a->timeline_tiles_per_frame;
        if (hipMemcpy(h.data() + 3, a->timeline, words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(knob_str(KNOB_TIMELINE), "wb")) {
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
    }
    if (a->line_bits && knob_str(KNOB_FOOTPRINT)) {
        // unique 128-byte lines per array, as JSON (tools/footprint.py reads it)
        const size_t words = ((size_t)a->line_base[6] + 31) / 32;
        std::vector<uint32_t> h(words);
        if (hipMemcpy(h.data(), a->line_bits, words * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *names[6] = {"grid2", "records", "nodes", "rows", "grid2_vox", "grid_vox"};
            unsigned long long lines[6] = {};
            for (int r = 0; r < 6; ++r)
                for (uint32_t l = a->line_base[r]; l < a->line_base[r + 1]; ++l) lines[r] += (h[l >> 5] >> (l & 31u)) & 1u;
            if (FILE *f = fopen(knob_str(KNOB_FOOTPRINT), "w")) {
                fprintf(f, "{");
                for (int r = 0; r < 6; ++r) fprintf(f, "\"%s_lines\": %llu, ", names[r], lines[r]);
                fprintf(f, "\"line_bytes\": 128}\n");
                fclose(f);
            }
        }
    }
    if (a->line_bits) (void)hipFree(a->line_bits);
    if (a->timeline) (void)hipFree(a->timeline);
    if (a->recs) (void)hipFree(a->recs);
    if (a->grid2i) (void)hipFree(a->grid2i);
    if (a->shadow_nodes) (void)hipFree(a->shadow_nodes);
    if (a->shadow_rows) (void)hipFree(a->shadow_rows);
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
int32_t mnv_accel_brick_levels(const mnv_accel *a) { return a ? (a->view.recs ? 2 : a->view.grid2i ? 1 : 0) : -1; }

int mnv_accel_lookup_coverage(const mnv_accel *a, int64_t out4[4]) {
    if (!a || !out4) return set_error(MNV_E_INVALID, "null argument");
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (a->view.grid2_level <= 0) return MNV_OK;
    mnv_accel *mut = const_cast<mnv_accel *>(a);
    std::lock_guard<std::mutex> view_lock(mut->launch_mutex);
    unsigned long long *dev = nullptr, h[4] = {};
    int rc;
    if ((rc = check_hip(hipMalloc((void **)&dev, sizeof(h)), "hipMalloc(coverage)"))) return rc;
    rc = check_hip(hipMemset(dev, 0, sizeof(h)), "memset coverage");
    const int64_t cells = (int64_t)1 << (3 * a->view.grid2_level);
    if (!rc) {
        hipLaunchKernelGGL(accel_count_coverage, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, nullptr, a->nodes, a->grid2, a->view.grid2i, cells, dev);
        rc = check_hip(hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost), "read coverage");  // (the null stream: behind every launch of the device)
    }
    (void)hipFree(dev);
    if (rc) return rc;
    out4[0] = (int64_t)h[0];
    out4[1] = (int64_t)h[1];
    out4[2] = (int64_t)h[2];
    out4[3] = a->view.recs ? (int64_t)h[0] : 0;
    return MNV_OK;
}

int mnv_accel_set_cu_budget(mnv_accel *a, int32_t num_cus) {
    if (!a) return set_error(MNV_E_INVALID, "accel is null");
    if (num_cus > a->device_cus) return set_error(MNV_E_INVALID, "the device has fewer compute units");
    a->num_cus = num_cus <= 0 ? a->device_cus : num_cus;
    return MNV_OK;
}

}  // extern "C"
