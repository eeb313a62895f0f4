// mnv_accel.h -- the packed device layout ("accel") behind mnv_render_voxels_accel.
//
// HBM layout (built once per tree upload by mnv_accel_create):
//   nodes [capacity*8] u32   one word per voxel, replacing BOTH the child word
//                            (rt_core.cuh:146) and the sigma half (rt_core.cuh:231):
//                              internal: absolute index of the child chunk (1 .. 2^31-1)
//                              leaf    : 0x80000000 | depth << 16 | sigma(binary16 bits)
//   rows  [capacity*8][row_bytes]   the colour halfs of a voxel: three channel blocks, each padded
//                            to whole dwords (18 -> 20 B for SH9), the row rounded up to a power
//                            of two (60 -> 64 B) so that it never straddles a 128-B line; the lane
//                            evaluating one (sample, channel) pair loads its block's dwords
//   grid  [2^L]^3 u32        dense top-of-tree lookup at level L = grid_level: the node
//                            word of the depth-L voxel covering the cell, or the
//                            (shallower) leaf word that covers it; staged in LDS
//   grid_vox [2^L]^3 u32     voxel index of that covering leaf (read for dense samples only)
//   grid2 / grid2_vox [2^L2]^3 u32   the same two arrays at level L2 <= min(max_depth-1, 9), in
//                            4x4x4-cell brick order; a step below the LDS grid costs one load
//                            here plus one node load per level below L2
//   grid2i [2^L2]^3 u32      grid2 with the LAST level folded into the cell word (every frame kind reads it instead of grid2; mnv_accel_refresh
//                            patches it with the cells it rewrites, a prune derives it again): a non-leaf cell whose chunk holds eight LEAVES
//                            (and has a number below 2^22) reads
//                              0 | 1 << 30 | (voxel s1 has sigma bits != 0) << (22 + s1) | (chunk - inline_base)
//                            inline_base = the smallest chunk number of depth L2 + 1 when the words are derived: the 22-bit field then spans the
//                            4.19 M chunk numbers FROM there (the deepest levels carry the highest numbers; the reference budgets 20 M chunks,
//                            src/opts.cpp:24) instead of the first 4.19 M of the tree; a chunk outside it is simply not inline (records / node
//                            words answer; mnv_accel_lookup_coverage counts them)
//                            so a step into an empty leaf one level below the grid costs no load beyond the grid cell, and a dense one goes
//                            straight to its colour row (sigma is in the row).  cfg2 (depth 10, L2 = 9): every deep step; no node word is read.
//   recs [capacity][8] {u32 child, u32 codes}   brick records (trees with leaves two or more levels below L2 only; kept current by
//                            mnv_accel_refresh and the prune): record c belongs to chunk c of depth L2 + 1 -- the chunk a non-leaf grid2
//                            cell names -- and describes the 4x4x4 cells of level L2 + 2 under it in 64 bytes.  Entry s1 (one 8-byte load):
//                              child   chunk of the children of voxel s1 (0: that voxel is a leaf)
//                              codes   two bits per sub-cell s2 (bits 2 * s2):
//                                      0 walk the node words (inner voxel of depth L2 + 2, or voxel s1 is a leaf with sigma != 0)
//                                      1 voxel s1 is a leaf (depth L2 + 1) with sigma bits 0      2 leaf of depth L2 + 2, sigma bits 0
//                                      3 leaf of depth L2 + 2, sigma != 0: voxel child * 8 + s2; its sigma is read with its colour row
//                            Near a thin surface most fine steps land in EMPTY leaves of the last two levels: a 128-byte line of node words
//                            describes 32 voxels, a line of records 128 cells, and the level in between is not read at all (cfg3: node
//                            words were 41 % of the L2 misses, profiles/r05_traffic_cfg3_by_array.json).
//   rows: the half behind the three channel blocks (half 3 * chan_halfs; every format has it spare) holds the voxel's sigma, so that a
//                            dense sample found through a record needs no node word at all.
// The in-leaf coordinates the march needs are frac(pos * 2^depth); x*2, floorf and
// x - floorf(x) are exact in binary32 for x in [0,2), so any traversal that reaches the
// same leaf reproduces the reference's iterated descent bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "../../include/mnv.h"
#include "mnv_device.h"

namespace mnv {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr int kMaxGridLevel = 5;  // 32^3 * 4 B = 128 KiB of the CU's 160 KiB LDS
constexpr int kNumQueues = 8;     // one ray queue per XCD
constexpr int kSlots = 64;        // per-launch parameter slots in flight
constexpr size_t kSlotBytes = (size_t)MNV_MAX_BATCH * (kNumQueues * 64 + sizeof(mnv::CamBlock));
constexpr int kMaxGrid2Level = 9;  // 8^9 * 4 B = 512 MiB per array
constexpr uint32_t kInlineBit = 0x40000000u;  // grid2i: the cell's eight children are leaves, described by the word itself
constexpr int kInlineMaskShift = 22;           // ... bit (22 + s1): child s1 has sigma bits != 0; bits 0..21: the chunk
constexpr int kRecWords = 16;      // a brick record: 8 entries {child chunk, 8 two-bit sub-cell codes} = 64 B

// Interleaved macro-tile partition (mnv_partition in include/mnv.h).  Tiles are dealt in rounds of `world`; with a root period
// M >= 2 every M-th round leaves rank 0 out, so a period is L = world * M - 1 tiles of which rank 0 owns M - 1 and the others M.
__host__ __device__ inline uint32_t part_tile_of(uint32_t j, int32_t rank, int32_t world, int32_t M) {  // local tile j of `rank` -> macro tile
    if (M < 2) return (uint32_t)rank + j * (uint32_t)world;
    const uint32_t L = (uint32_t)world * (uint32_t)M - 1u, c = rank == 0 ? (uint32_t)M - 1u : (uint32_t)M;
    const uint32_t p = j / c, k = j - p * c;
    return p * L + (k + 1u < (uint32_t)M ? k * (uint32_t)world + (uint32_t)rank : ((uint32_t)M - 1u) * (uint32_t)world + (uint32_t)rank - 1u);
}
__host__ __device__ inline void part_owner_of(uint32_t m, int32_t world, int32_t M, uint32_t &rank, uint32_t &j) {  // macro tile -> (rank, local tile)
    if (M < 2) {
        rank = m % (uint32_t)world;
        j = m / (uint32_t)world;
        return;
    }
    const uint32_t L = (uint32_t)world * (uint32_t)M - 1u, p = m / L, o = m - p * L, full = ((uint32_t)M - 1u) * (uint32_t)world;
    uint32_t k;
    if (o < full) {
        k = o / (uint32_t)world;
        rank = o - k * (uint32_t)world;
    } else {
        k = (uint32_t)M - 1u;
        rank = o - full + 1u;
    }
    j = p * (rank == 0 ? (uint32_t)M - 1u : (uint32_t)M) + k;
}
// number of macro tiles < n_macro that `rank` owns
inline int64_t part_local_count(int64_t n_macro, int32_t rank, int32_t world, int32_t M) {
    if (M < 2) return rank >= n_macro ? 0 : (n_macro - rank + world - 1) / world;
    const int64_t L = (int64_t)world * M - 1, c = rank == 0 ? M - 1 : M;
    int64_t n = (n_macro / L) * c;
    const int64_t rem = n_macro % L;
    for (int64_t k = 0; k < c; ++k)
        if ((int64_t)part_tile_of((uint32_t)k, rank, world, M) < rem) ++n;
    return n;
}
inline int64_t part_j_max(int64_t n_macro, int32_t world, int32_t M) {
    int64_t best = 0;
    for (int32_t r = 0; r < world; ++r) best = best > part_local_count(n_macro, r, world, M) ? best : part_local_count(n_macro, r, world, M);
    return best;
}

struct AccelView {
    const uint32_t *nodes;
    const uint8_t *rows;
    const uint32_t *grid;
    const uint32_t *grid_vox;  // [2^L]^3: voxel index (chunk*8+child) of the leaf covering a grid cell
    int32_t grid_level;
    const uint32_t *grid2;      // [2^L2]^3 brick-ordered second lookup grid (NULL when grid2_level == 0)
    const uint32_t *grid2_vox;
    int32_t grid2_level;
    const uint32_t *grid2i;     // grid2 with inline last-level words (NULL: none; frames then read grid2 and walk the node words)
    uint32_t inline_base;       // chunk number the 22-bit chunk field of an inline word is relative to
    const uint2 *recs;          // [capacity][8] brick records of levels grid2_level + 1 and + 2 (NULL: none, the node words are walked)
    int32_t sigma_off;          // byte offset of the sigma half inside a colour row
    int32_t max_depth;          // deepest voxel depth of the tree (<= 23)
    int32_t row_bytes;
    float offset[3], scale[3];
    int32_t data_dim, basis_dim, format, capacity;
};

}  // namespace mnv

struct mnv_accel {
    mnv::AccelView view;
    uint32_t *nodes = nullptr;
    uint8_t *rows = nullptr;
    uint32_t *grid = nullptr;
    uint32_t *grid_vox = nullptr;
    uint32_t *grid2 = nullptr;
    uint32_t *grid2_vox = nullptr;
    uint32_t *grid2i = nullptr;           // [2^L2]^3
    uint2 *recs = nullptr;                // [reserved][8]
    uint32_t *shadow_nodes = nullptr;     // MNV_ABLATE shadow loads (test-hook build, diagnostics instantiation): copies of nodes / rows at other
    uint8_t *shadow_rows = nullptr;       // addresses, read with the same access pattern to attribute the HBM traffic by array
    uint32_t *nodes_spare = nullptr;      // second set of nodes / rows / depth, allocated by the first prune (accel_apply_prune writes the
    uint8_t *rows_spare = nullptr;        // survivors out of place, then the sets swap)
    int32_t *depth_spare = nullptr;
    int32_t *depth = nullptr;             // [reserved] depth of the voxels of each chunk (root chunk: 1); kept for mnv_accel_refresh
    int32_t *flags = nullptr;             // [8] device scratch of refresh: changed, deepest depth, grids dirty, shallowest voxel, patch items (appended / changed)
    uint32_t *patch_prefix = nullptr;     // refresh: first patch item of every affected voxel (grow-only)
    size_t patch_prefix_words = 0;
    int64_t reserved = 0;                 // chunks the nodes / rows / depth arrays have room for
    unsigned long long *stats = nullptr;  // MNV_STATS=1 diagnostics
    uint32_t *fault_dev = nullptr;        // [1] guided_fused2_kernel: spin-waits abandoned by the watchdog since creation (always counted, with or without
    uint32_t *fault_host = nullptr;       // mnv_set_fused_diag); pinned mirror, refreshed behind every fused launch -- mnv_accel_fused_faults
    uint32_t fault_reported = 0;          // faults already answered with MNV_E_FAULT
    uint32_t *line_bits = nullptr;        // MNV_FOOTPRINT=<file> diagnostics: one bit per 128-byte line of grid2i / grid2, recs, nodes, rows, grid2_vox, grid_vox
    uint32_t line_base[7] = {};           // first line of each of those regions in the bitmap; [6] = all lines
    unsigned long long *timeline = nullptr;  // MNV_TIMELINE=<file> diagnostics: tile / wavefront time stamps of the last launch
    size_t timeline_bytes = 0, timeline_tiles = 0, timeline_waves = 0, timeline_tiles_per_frame = 0;
    // per-launch slots: [kNumQueues] ray-queue heads (64 B apart; a queue spans the frames of a batch) + [n_frames] camera blocks,
    // written on the launch stream by stage_launch_kernel; kSlots launches may be in flight
    uint8_t *slots_dev = nullptr;
    std::mutex launch_mutex;              // guards the three slot fields below (launch_accel)
    std::atomic<uint32_t> slot_counter{0};
    hipEvent_t slot_done[mnv::kSlots] = {};  // recorded after the launch that used the slot
    bool slot_used[mnv::kSlots] = {};
    std::atomic<int> colour_math{0};      // mnv_accel_set_colour_math: 0 exact (bit-identical to the oracle), 1 hardware exp2 / rcp in the colour sigmoid
    std::atomic<int> fused_kernel{0};     // mnv_accel_set_fused_kernel: 0 = the first that fits, 2 = producer / consumer wavefronts, 1 = one role
    std::atomic<unsigned long long *> fused_diag{nullptr};  // mnv_accel_set_fused_diag: 32 device words the fused kernels add their clocks / counts to
    size_t bytes = 0;
    int device = 0;
    int num_cus = 0;     // units the persistent launch fills (mnv_accel_set_cu_budget)
    int device_cus = 0;  // units of the device
};
