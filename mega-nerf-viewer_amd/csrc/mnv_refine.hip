// mnv_refine.hip -- the tracker-consuming side of the refinement loop (BASELINE config 5), device-resident.
//
// The reference runs this as a chain of libtorch tensor ops on the host thread
// (src/renderer/cuda_renderer.cpp:205-381: expand_voxels, get_more_samples, prune_tree); here each step is
// one entry point working on the caller's device arrays:
//   mnv_select_split_candidates    cuda_renderer.cpp:205-227  (unique_dim vote count, count >= 2, order)
//   mnv_select_sample_candidates   cuda_renderer.cpp:281-296  (unique_dim, order)
//   mnv_apply_split_results        cuda_renderer.cpp:262-272  (mean over samples -> new f16 rows)
//   mnv_apply_sample_results       cuda_renderer.cpp:307-332  (running average)
//   mnv_prune_tree                 cuda_renderer.cpp:335-381  (visit-mark compaction)
// Sorting / run-length / scan primitives come from rocPRIM; the row arithmetic is written here.
// All of it is integer or byte work except the two averages (fp32, rounded once to binary16).

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include <rocprim/block/block_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "mnv_internal.h"
#include "mnv_knobs.h"

namespace mnv {

namespace {

// ---------------------------------------------------------------- grow-only device workspace, per device
struct Workspace {
    void *ptr = nullptr;
    size_t cap = 0;
};
std::mutex g_ws_mutex;
Workspace g_ws[16];

int ws_reserve(size_t bytes, uint8_t **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return set_error(MNV_E_NO_DEVICE, "no current HIP device");
    Workspace &w = g_ws[dev];
    if (w.cap < bytes) {
        if (w.ptr) (void)hipFree(w.ptr);
        w.ptr = nullptr;
        w.cap = 0;
        const size_t want = bytes + bytes / 4;
        int rc = check_hip(hipMalloc(&w.ptr, want), "hipMalloc(refine workspace)");
        if (rc) return rc;
        w.cap = want;
    }
    *out = static_cast<uint8_t *>(w.ptr);
    return MNV_OK;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    }
};

// ---------------------------------------------------------------- candidate selection

// Tracker rows are (priority, chunk, child) as floats holding integers (rt_core.cuh:239-251,310-320);
// lexicographic order of the rows == order of this packed key.  Rows with chunk < 0 are "no candidate".
// Two key layouts.  WIDE (52 bits: 16 of priority, 32 of chunk, 3 of child) holds whatever the contract of include/mnv.h admits and
// costs seven 8-bit radix passes.  COMPACT is what the march actually writes -- a voxel index below 2^27 (trees up to 16.7 M chunks)
// and a priority in [-1, 2^PB - 3] (a depth for the split tracker: PB = 5; a sample count for the sample tracker: PB = 9; the top value
// of the field marks "no candidate", which must sort last) -- in 27 + PB bits: four or five passes.  The pack kernel raises a flag for any row that does not fit and the caller repeats the call
// with the wide layout (never seen in practice; the flag costs nothing, it travels with the counts the host reads anyway).
constexpr int kVoxBits = 27;
constexpr uint64_t kNoCandidateWide = 1ull << 51;
constexpr int kKeyBitsWide = 52;

template <bool WIDE, typename KeyT = uint64_t>
__global__ void pack_tracker_keys(const float *__restrict__ track, int64_t n_rows, KeyT *__restrict__ keys, int prio_bits, uint32_t *__restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const float prio = track[i * 3 + 0], chunk = track[i * 3 + 1], child = track[i * 3 + 2];
    const uint64_t none = WIDE ? kNoCandidateWide : (((1ull << prio_bits) - 1) << kVoxBits);
    uint64_t key = none;
    if (chunk >= 0.f) {
        if (WIDE) {
            const uint64_t p = (uint64_t)((int32_t)prio + 32768) & 0xffffu;
            key = (p << 35) | ((uint64_t)(uint32_t)(int32_t)chunk << 3) | ((uint64_t)(int32_t)child & 7u);
        } else {
            const int64_t vox = (int64_t)(int32_t)chunk * 8 + ((int32_t)child & 7), p = (int64_t)(int32_t)prio + 1;
            if (vox >= ((int64_t)1 << kVoxBits) || p < 0 || p >= ((int64_t)1 << prio_bits) - 1) *overflow = 1u;  // (benign race: every writer stores 1)
            key = ((uint64_t)p << kVoxBits) | (uint64_t)vox;
        }
    }
    keys[i] = (KeyT)key;
}

// info[0] = runs that are real candidates, info[1] = those with count >= 2, info[3] = the largest count among the candidates
__global__ void count_candidates(const uint64_t *__restrict__ unique_keys, const uint32_t *__restrict__ counts,
                                 const uint32_t *__restrict__ n_runs, uint32_t *__restrict__ info, uint64_t none) {
    const uint32_t n = *n_runs;
    uint32_t valid = 0, voted = 0, most = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (unique_keys[i] != none) {
            ++valid;
            voted += counts[i] >= 2u;
            most = counts[i] > most ? counts[i] : most;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        valid += __shfl_down(valid, off);
        voted += __shfl_down(voted, off);
        const uint32_t other = __shfl_down(most, off);
        most = other > most ? other : most;
    }
    // one atomic per workgroup: a thousand atomics on one address are served one after the other (40 us for this kernel in round 3's trace)
    __shared__ uint32_t s_part[3][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_part[0][wave] = valid;
        s_part[1][wave] = voted;
        s_part[2][wave] = most;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0, w = 0, m = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
            v += s_part[0][k];
            w += s_part[1][k];
            m = s_part[2][k] > m ? s_part[2][k] : m;
        }
        if (v) atomicAdd(&info[0], v);
        if (w) atomicAdd(&info[1], w);
        if (m) atomicMax(&info[3], m);
    }
}

template <bool WIDE>
__global__ void unpack_nodes(const uint64_t *__restrict__ keys, int32_t n, int32_t *__restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    const uint64_t vox = WIDE ? (k & 0x7ffffffffull) : (k & ((1ull << kVoxBits) - 1));
    nodes[2 * i + 0] = (int32_t)(vox >> 3);
    nodes[2 * i + 1] = (int32_t)(vox & 7u);
}

template <bool WIDE>
int select_candidates_as(const float *track, int64_t n_rows, int32_t max_out, bool need_votes, int32_t *nodes_out, int32_t *n_out, int32_t *n_candidates,
                         bool *overflow, hipStream_t stream) {
    const size_t n = (size_t)n_rows;
    const int prio_bits = need_votes ? 5 : 9, key_bits = WIDE ? kKeyBitsWide : kVoxBits + prio_bits;
    const uint64_t none = WIDE ? kNoCandidateWide : (((1ull << prio_bits) - 1) << kVoxBits);

    size_t tmp_sort = 0, tmp_rle = 0, tmp_sort2 = 0;
    uint64_t *nk = nullptr;
    uint32_t *nc = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, tmp_sort, nk, nk, n, 0, key_bits, stream);
    (void)rocprim::run_length_encode(nullptr, tmp_rle, nk, (unsigned int)n, nk, nc, nc, stream);
    (void)rocprim::radix_sort_pairs_desc(nullptr, tmp_sort2, nc, nc, nk, nk, n, 0, 32, stream);
    const size_t tmp_bytes = std::max(tmp_sort, std::max(tmp_rle, tmp_sort2));

    Carver c;
    const size_t o_keys = c.take(n * 8), o_sorted = c.take(n * 8), o_unique = c.take(n * 8), o_counts = c.take(n * 4);
    const size_t o_counts2 = c.take(n * 4), o_info = c.take(64), o_tmp = c.take(tmp_bytes);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    uint8_t *ws = nullptr;
    int rc = ws_reserve(c.off, &ws);
    if (rc) return rc;
    uint64_t *keys = reinterpret_cast<uint64_t *>(ws + o_keys), *sorted = reinterpret_cast<uint64_t *>(ws + o_sorted);
    uint64_t *unique_keys = reinterpret_cast<uint64_t *>(ws + o_unique);
    uint32_t *counts = reinterpret_cast<uint32_t *>(ws + o_counts), *counts2 = reinterpret_cast<uint32_t *>(ws + o_counts2);
    uint32_t *info = reinterpret_cast<uint32_t *>(ws + o_info);  // [0] valid, [1] voted, [2] runs, [3] largest count, [4] a row did not fit the compact key
    void *tmp = ws + o_tmp;

    if ((rc = check_hip(hipMemsetAsync(info, 0, 64, stream), "memset"))) return rc;
    hipLaunchKernelGGL(pack_tracker_keys<WIDE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, track, n_rows, keys, prio_bits, info + 4);
    size_t t = tmp_bytes;
    if ((rc = check_hip(rocprim::radix_sort_keys(tmp, t, keys, sorted, n, 0, key_bits, stream), "radix_sort_keys"))) return rc;
    t = tmp_bytes;
    if ((rc = check_hip(rocprim::run_length_encode(tmp, t, sorted, (unsigned int)n, unique_keys, counts, info + 2, stream), "run_length_encode")))
        return rc;
    hipLaunchKernelGGL(count_candidates, dim3(64), dim3(256), 0, stream, unique_keys, counts, info + 2, info, none);
    uint32_t h[5] = {0, 0, 0, 0, 0};
    if ((rc = check_hip(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream), "copy counts"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "select_candidates"))) return rc;
    if (!WIDE && h[4]) {
        *overflow = true;  // a row outside the compact key's range: the caller repeats with the wide layout
        return MNV_OK;
    }
    const uint32_t n_valid = h[0], n_voted = h[1], most = h[3];

    const uint64_t *ordered = unique_keys;  // ascending (priority, chunk, child); "no candidate" sorts last
    uint32_t n_sel = n_valid;
    if (need_votes) {
        // cuda_renderer.cpp:213-217: rows (-count, priority, chunk, child) with count >= 2, sorted ascending ==
        // count descending, ties in key order -- a stable descending sort of the already key-ordered runs.
        // The "no candidate" run would sort first by its count, so it is cut off before the sort.  Only the bits the largest count
        // has are sorted (it came with the other counts): one or two passes instead of four.
        n_sel = n_voted;
        if (n_voted > 0) {
            int count_bits = 1;
            while (count_bits < 32 && (most >> count_bits) != 0u) ++count_bits;
            t = tmp_bytes;
            if ((rc = check_hip(rocprim::radix_sort_pairs_desc(tmp, t, counts, counts2, unique_keys, sorted, (size_t)n_valid, 0, count_bits, stream),
                                "radix_sort_pairs_desc")))
                return rc;
            ordered = sorted;
        }
    }
    const int32_t n_write = (int32_t)std::min<uint32_t>(n_sel, (uint32_t)max_out);
    if (n_write > 0) hipLaunchKernelGGL(unpack_nodes<WIDE>, dim3((n_write + 255) / 256), dim3(256), 0, stream, ordered, n_write, nodes_out);
    // the workspace is reused by the next call: finish before the lock is released
    if ((rc = check_hip(hipStreamSynchronize(stream), "unpack_nodes"))) return rc;
    if (n_out) *n_out = n_write;
    if (n_candidates) *n_candidates = (int32_t)n_sel;
    return MNV_OK;
}

// ---- the compact-key path without a host round trip in the middle and without a sort of the counts
// What the caller keeps of the vote is its first max_out rows (split_batch_size: 4096) and the number of voted voxels; sorting all
// ~2 x 10^5 runs by count (rocPRIM picks a merge sort at that size: a block sort and 18 merge launches, 155 us of a configs[4]
// frame's trace) to keep 4096 of them is a selection problem: a histogram of the counts gives the count T of the last row that fits,
// the runs above T (fewer than max_out) are sorted by (count descending, key) in ONE workgroup, and of the runs AT T the first ones in
// key order fill the rest -- an order-preserving compaction in which every wavefront owns a contiguous range of runs.  Every kernel takes
// the number of runs from device memory, so the host waits once, at the end.  Rows in the order of cuda_renderer.cpp:213-217 as before.
constexpr int kVoteBins = 2048;   // histogram of min(count, kVoteBins - 1); a threshold in the last bin cannot be resolved -> full sort
constexpr int kVoteTop = 8192;    // the single-workgroup sort's capacity == the largest max_out this path serves (render_options.hpp:49: 4192)
constexpr int kVoteGrid = 256;    // workgroups of 256 threads; wavefront w of the grid owns runs [w * per, (w + 1) * per)
constexpr int kVoteWaves = kVoteGrid * 4;
static_assert(kVoteGrid == 256, "vote_totals: one partial per thread of a 256-thread workgroup");
enum VoteWord { kValid = 0, kVoted = 1, kRuns = 2, kMost = 3, kOverflow = 4, kThresh = 5, kAbove = 6, kTies = 7, kFallback = 8, kWrite = 9, kCursor = 10, kVoteWords = 16 };

template <typename KeyT>
__global__ __launch_bounds__(256) void vote_stats(const KeyT *__restrict__ unique_keys, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ info,
                                                  uint32_t *__restrict__ hist, uint32_t *__restrict__ partial, KeyT none, int with_hist) {
    __shared__ uint32_t lh[kVoteBins];
    __shared__ uint32_t s_part[3][4];
    if (with_hist)
        for (int b = threadIdx.x; b < kVoteBins; b += 256) lh[b] = 0;
    __syncthreads();
    const uint32_t n = info[kRuns];
    uint32_t valid = 0, voted = 0, most = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (unique_keys[i] != none) {
            const uint32_t c = counts[i];
            ++valid;
            most = c > most ? c : most;
            if (c >= 2u) {
                ++voted;
                if (with_hist) atomicAdd(&lh[c < (uint32_t)kVoteBins - 1 ? c : (uint32_t)kVoteBins - 1], 1u);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        valid += __shfl_down(valid, off);
        voted += __shfl_down(voted, off);
        const uint32_t other = __shfl_down(most, off);
        most = other > most ? other : most;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_part[0][wave] = valid;
        s_part[1][wave] = voted;
        s_part[2][wave] = most;
    }
    __syncthreads();
    if (with_hist)
        for (int b = threadIdx.x; b < kVoteBins; b += 256)
            if (lh[b]) atomicAdd(&hist[b], lh[b]);
    if (threadIdx.x == 0) {  // per-workgroup partial sums: 768 atomics on three addresses are served one after the other (10 us)
        uint32_t v = 0, w = 0, m = 0;
        for (int k = 0; k < 4; ++k) {
            v += s_part[0][k];
            w += s_part[1][k];
            m = s_part[2][k] > m ? s_part[2][k] : m;
        }
        partial[blockIdx.x] = v;
        partial[kVoteGrid + blockIdx.x] = w;
        partial[2 * kVoteGrid + blockIdx.x] = m;
    }
}

// totals of vote_stats' partials, by a workgroup of 256 threads (kVoteGrid == 256); `record`: this workgroup writes them to info
__device__ inline void vote_totals(const uint32_t *__restrict__ partial, uint32_t *__restrict__ info, bool record, uint32_t *valid, uint32_t *voted) {
    __shared__ uint32_t s_tot[3][4];
    const int t = threadIdx.x;
    uint32_t v = partial[t], w = partial[kVoteGrid + t], m = partial[2 * kVoteGrid + t];
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off);
        w += __shfl_xor(w, off);
        const uint32_t other = __shfl_xor(m, off);
        m = other > m ? other : m;
    }
    if ((t & 63) == 0) {
        s_tot[0][t >> 6] = v;
        s_tot[1][t >> 6] = w;
        s_tot[2][t >> 6] = m;
    }
    __syncthreads();
    v = s_tot[0][0] + s_tot[0][1] + s_tot[0][2] + s_tot[0][3];
    w = s_tot[1][0] + s_tot[1][1] + s_tot[1][2] + s_tot[1][3];
    m = max(max(s_tot[2][0], s_tot[2][1]), max(s_tot[2][2], s_tot[2][3]));
    if (record && t == 0) {
        info[kValid] = v;
        info[kVoted] = w;
        info[kMost] = m;
    }
    *valid = v;
    *voted = w;
}

// the range of runs wavefront w of the grid owns (a multiple of 64 long, so a wavefront's loads stay aligned)
__device__ inline void vote_range(uint32_t n, int w, uint32_t *begin, uint32_t *end) {
    const uint32_t per = ((n + kVoteWaves - 1) / kVoteWaves + 63u) & ~63u;
    const uint64_t b = (uint64_t)w * per, e = b + per;
    *begin = b < n ? (uint32_t)b : n;
    *end = e < n ? (uint32_t)e : n;
}

// threshold from the histogram (every workgroup computes the same one; workgroup 0 records it), then: runs above it appended to
// `above` as (~count, key) words, runs at it counted per wavefront
__global__ __launch_bounds__(256) void vote_gather(const uint32_t *__restrict__ unique_keys, const uint32_t *__restrict__ counts, uint32_t *__restrict__ info,
                                                   const uint32_t *__restrict__ hist, const uint32_t *__restrict__ partial, uint32_t *__restrict__ tie_count,
                                                   uint64_t *__restrict__ above, uint32_t none, uint32_t max_out) {
    __shared__ uint32_t s_sum[2][256];
    __shared__ uint32_t s_t[3];
    const int t = threadIdx.x;
    uint32_t valid, voted;
    vote_totals(partial, info, blockIdx.x == 0, &valid, &voted);
    if (max_out == 0) return;  // (a count-only call: one workgroup, for the totals)
    uint32_t mine[8], part = 0;
    for (int j = 0; j < 8; ++j) {
        mine[j] = hist[8 * t + j];
        part += mine[j];
    }
    s_sum[0][t] = part;
    __syncthreads();
    int cur = 0;
    for (int step = 1; step < 256; step <<= 1) {  // inclusive suffix sums: s_sum[cur][t] = bins of chunks t .. 255
        s_sum[cur ^ 1][t] = s_sum[cur][t] + (t + step < 256 ? s_sum[cur][t + step] : 0u);
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t incl = s_sum[cur][t], excl = incl - part;
    if (voted <= max_out) {
        if (t == 0) {
            s_t[0] = 1u;  // every voted run is "above"
            s_t[1] = voted;
            s_t[2] = 0u;
        }
    } else if (excl < max_out && incl >= max_out) {  // the max_out-th largest count lies in this thread's bins (exactly one thread)
        uint32_t acc = excl;
        for (int j = 7; j >= 0; --j) {
            if (acc + mine[j] >= max_out) {
                s_t[0] = (uint32_t)(8 * t + j);
                s_t[1] = acc;
                s_t[2] = max_out - acc;
                break;
            }
            acc += mine[j];
        }
    }
    __syncthreads();
    const uint32_t T = s_t[0];
    if (blockIdx.x == 0 && t == 0) {
        info[kThresh] = T;
        info[kAbove] = s_t[1];
        info[kTies] = s_t[2];
        info[kFallback] = T >= (uint32_t)kVoteBins - 1 ? 1u : 0u;
        info[kWrite] = voted < max_out ? voted : max_out;
    }
    if (T >= (uint32_t)kVoteBins - 1) return;
    const int lane = t & 63, w = blockIdx.x * 4 + (t >> 6);
    uint32_t begin, end, ties = 0;
    vote_range(info[kRuns], w, &begin, &end);
    for (uint32_t i0 = begin; i0 < end; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t key = none, c = 0;
        if (i < end) {
            key = unique_keys[i];
            c = counts[i];
        }
        const bool ok = key != none, is_above = ok && c > T, is_tie = ok && c == T && T >= 2u;
        const uint64_t m = __ballot(is_above);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&info[kCursor], (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (is_above) above[base + __popcll(m & ((1ull << lane) - 1))] = ((uint64_t)(0xffffffffu - c) << 32) | key;
        }
        ties += (uint32_t)__popcll(__ballot(is_tie));
    }
    if (lane == 0) tie_count[w] = ties;
}

__device__ inline void vote_write_node(int32_t *__restrict__ nodes, uint32_t at, uint32_t key) {
    const uint32_t vox = key & ((1u << kVoxBits) - 1u);
    nodes[2 * at + 0] = (int32_t)(vox >> 3);
    nodes[2 * at + 1] = (int32_t)(vox & 7u);
}

// the runs above the threshold, sorted by (count descending, key ascending): rows 0 .. above-1 of the result
template <int PER>
using VoteSort = rocprim::block_sort<uint64_t, 1024, PER, rocprim::empty_type, rocprim::block_sort_algorithm::stable_merge_sort>;

template <int PER>  // 1024 * PER rows at most (PER 4: 32 KB of LDS; PER 8: 64 KB and a bit, hence not a static array)
__global__ __launch_bounds__(1024) void vote_sort_above(const uint64_t *__restrict__ above, const uint32_t *__restrict__ info, int32_t *__restrict__ nodes) {
    extern __shared__ __align__(16) unsigned char vote_lds[];
    typename VoteSort<PER>::storage_type &storage = *reinterpret_cast<typename VoteSort<PER>::storage_type *>(vote_lds);
    if (info[kFallback]) return;
    const uint32_t n = info[kAbove];
    uint64_t v[PER];
    for (int j = 0; j < PER; ++j) {
        const uint32_t at = threadIdx.x * PER + j;
        v[j] = at < n ? above[at] : ~0ull;
    }
    VoteSort<PER>().sort(v, storage);
    for (int j = 0; j < PER; ++j) {
        const uint32_t at = threadIdx.x * PER + j;
        if (at < n) vote_write_node(nodes, at, (uint32_t)v[j]);
    }
}

template <int PER>
int launch_vote_sort(const uint64_t *above, const uint32_t *info, int32_t *nodes, hipStream_t stream) {
    const int bytes = (int)sizeof(typename VoteSort<PER>::storage_type);
    const int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&vote_sort_above<PER>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "lds attr");
    if (rc) return rc;
    hipLaunchKernelGGL(vote_sort_above<PER>, dim3(1), dim3(1024), bytes, stream, above, info, nodes);
    return MNV_OK;
}

// the first info[kTies] runs AT the threshold, in key order: rows above .. above+ties-1
__global__ __launch_bounds__(256) void vote_ties(const uint32_t *__restrict__ unique_keys, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ info,
                                                 const uint32_t *__restrict__ tie_count, int32_t *__restrict__ nodes, uint32_t none) {
    const uint32_t want = info[kTies], T = info[kThresh], first = info[kAbove];
    if (want == 0 || info[kFallback]) return;
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t base = 0;
    for (int u = lane; u < w; u += 64) base += tie_count[u];
    for (int off = 32; off > 0; off >>= 1) base += __shfl_xor(base, off);
    uint32_t begin, end;
    vote_range(info[kRuns], w, &begin, &end);
    for (uint32_t i0 = begin; i0 < end && base < want; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t key = none, c = 0;
        if (i < end) {
            key = unique_keys[i];
            c = counts[i];
        }
        const bool is_tie = key != none && c == T;
        const uint64_t m = __ballot(is_tie);
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (is_tie && at < want) vote_write_node(nodes, first + at, key);
        base += (uint32_t)__popcll(m);
    }
}

// no vote (the sample tracker): the first valid runs in key order
template <typename KeyT>
__global__ __launch_bounds__(256) void vote_head(const KeyT *__restrict__ unique_keys, const uint32_t *__restrict__ partial, uint32_t *__restrict__ info,
                                                 int32_t *__restrict__ nodes, uint32_t max_out) {
    uint32_t valid, voted;
    vote_totals(partial, info, blockIdx.x == 0, &valid, &voted);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, n = valid < max_out ? valid : max_out;
    if (i == 0) info[kWrite] = n;
    if (i < n) vote_write_node(nodes, i, (uint32_t)(unique_keys[i] & ((1u << kVoxBits) - 1u)));
}

template <typename KeyT>
int select_candidates_compact(const float *track, int64_t n_rows, int32_t max_out, bool need_votes, int32_t *nodes_out, int32_t *n_out, int32_t *n_candidates,
                              bool *overflow, bool *full_sort, hipStream_t stream) {
    const size_t n = (size_t)n_rows;
    const int prio_bits = need_votes ? 5 : 9, key_bits = kVoxBits + prio_bits;
    const KeyT none = (KeyT)(((1ull << prio_bits) - 1) << kVoxBits);

    size_t tmp_sort = 0, tmp_rle = 0;
    KeyT *nk = nullptr;
    uint32_t *nc = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, tmp_sort, nk, nk, n, 0, key_bits, stream);
    (void)rocprim::run_length_encode(nullptr, tmp_rle, nk, (unsigned int)n, nk, nc, nc, stream);
    const size_t tmp_bytes = std::max(tmp_sort, tmp_rle);

    Carver c;
    const size_t o_keys = c.take(n * sizeof(KeyT)), o_sorted = c.take(n * sizeof(KeyT)), o_unique = c.take(n * sizeof(KeyT)), o_counts = c.take(n * 4);
    const size_t state_bytes = (size_t)(kVoteWords + kVoteBins + kVoteWaves + 3 * kVoteGrid) * 4;
    const size_t o_state = c.take(state_bytes), o_above = c.take((size_t)kVoteTop * 8), o_tmp = c.take(tmp_bytes);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    uint8_t *ws = nullptr;
    int rc = ws_reserve(c.off, &ws);
    if (rc) return rc;
    KeyT *keys = reinterpret_cast<KeyT *>(ws + o_keys), *sorted = reinterpret_cast<KeyT *>(ws + o_sorted), *unique_keys = reinterpret_cast<KeyT *>(ws + o_unique);
    uint32_t *counts = reinterpret_cast<uint32_t *>(ws + o_counts);
    uint32_t *info = reinterpret_cast<uint32_t *>(ws + o_state), *hist = info + kVoteWords, *tie_count = hist + kVoteBins, *partial = tie_count + kVoteWaves;
    uint64_t *above = reinterpret_cast<uint64_t *>(ws + o_above);
    void *tmp = ws + o_tmp;

    if ((rc = check_hip(hipMemsetAsync(info, 0, state_bytes, stream), "memset"))) return rc;
    hipLaunchKernelGGL((pack_tracker_keys<false, KeyT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, track, n_rows, keys, prio_bits, info + kOverflow);
    size_t t = tmp_bytes;
    if ((rc = check_hip(rocprim::radix_sort_keys(tmp, t, keys, sorted, n, 0, key_bits, stream), "radix_sort_keys"))) return rc;
    t = tmp_bytes;
    if ((rc = check_hip(rocprim::run_length_encode(tmp, t, sorted, (unsigned int)n, unique_keys, counts, info + kRuns, stream), "run_length_encode")))
        return rc;
    const bool select = need_votes && max_out > 0;
    hipLaunchKernelGGL(vote_stats<KeyT>, dim3(kVoteGrid), dim3(256), 0, stream, unique_keys, counts, info, hist, partial, none, select ? 1 : 0);
    if constexpr (sizeof(KeyT) == 4) {
        if (need_votes)
            hipLaunchKernelGGL(vote_gather, dim3(select ? kVoteGrid : 1), dim3(256), 0, stream, unique_keys, counts, info, hist, partial, tie_count, above, none,
                               (uint32_t)max_out);
        if (select) {
            // fewer rows above the threshold than max_out: the smaller sort when the batch allows it
            if ((rc = max_out <= 1024 ? launch_vote_sort<1>(above, info, nodes_out, stream)
                      : max_out <= 4096 ? launch_vote_sort<4>(above, info, nodes_out, stream) : launch_vote_sort<kVoteTop / 1024>(above, info, nodes_out, stream)))
                return rc;
            hipLaunchKernelGGL(vote_ties, dim3(kVoteGrid), dim3(256), 0, stream, unique_keys, counts, info, tie_count, nodes_out, none);
        }
    }
    if (!need_votes)  // (max_out == 0: one workgroup, for the totals)
        hipLaunchKernelGGL(vote_head<KeyT>, dim3((unsigned)std::max((max_out + 255) / 256, 1)), dim3(256), 0, stream, unique_keys, partial, info, nodes_out,
                           (uint32_t)max_out);
    uint32_t h[kVoteWords] = {};
    if ((rc = check_hip(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream), "copy counts"))) return rc;
    // (the workspace is reused by the next call: finished before the lock is released)
    if ((rc = check_hip(hipStreamSynchronize(stream), "select_candidates"))) return rc;
    if (h[kOverflow]) {
        *overflow = true;  // a row outside the compact key's range: the caller repeats with the wide layout
        return MNV_OK;
    }
    if (select && h[kFallback]) {
        *full_sort = true;  // more than max_out voxels with >= kVoteBins - 1 votes each: the caller repeats with the sort of all counts
        return MNV_OK;
    }
    if (n_out) *n_out = max_out > 0 ? (int32_t)h[kWrite] : 0;
    if (n_candidates) *n_candidates = (int32_t)(need_votes ? h[kVoted] : h[kValid]);
    return MNV_OK;
}

int select_candidates(const float *track, int64_t n_rows, int32_t max_out, bool need_votes, int32_t *nodes_out,
                      int32_t *n_out, int32_t *n_candidates, hipStream_t stream) {
    if (n_out) *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (!track || n_rows < 0 || max_out < 0 || (max_out > 0 && !nodes_out)) return set_error(MNV_E_INVALID, "invalid tracker arguments");
    if (n_rows == 0) return MNV_OK;
    if (n_rows > 0x7fffffff) return set_error(MNV_E_UNSUPPORTED, "more than 2^31 - 1 tracker rows");
    static const bool force_wide = knob_set(KNOB_VOTE_WIDE_KEYS);       // (tests: the fallback paths on ordinary inputs)
    static const bool force_full_sort = knob_set(KNOB_VOTE_FULL_SORT);
    bool overflow = force_wide, full_sort = force_full_sort || (need_votes && max_out > kVoteTop);
    if (!overflow && !full_sort) {
        const int rc = need_votes ? select_candidates_compact<uint32_t>(track, n_rows, max_out, true, nodes_out, n_out, n_candidates, &overflow, &full_sort, stream)
                                  : select_candidates_compact<uint64_t>(track, n_rows, max_out, false, nodes_out, n_out, n_candidates, &overflow, &full_sort, stream);
        if (rc || (!overflow && !full_sort)) return rc;
    }
    if (!overflow) {
        const int rc = select_candidates_as<false>(track, n_rows, max_out, need_votes, nodes_out, n_out, n_candidates, &overflow, stream);
        if (rc || !overflow) return rc;
    }
    return select_candidates_as<true>(track, n_rows, max_out, need_votes, nodes_out, n_out, n_candidates, &overflow, stream);
}

// union of the ranks' visit marks (mnv_merge_visit_marks)
__global__ void merge_marks_kernel(const int32_t *__restrict__ table, int32_t world, int32_t capacity, int32_t *__restrict__ visited) {
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= capacity) return;
    int32_t m = 0;
    for (int32_t r = 0; r < world; ++r) m = max(m, table[(int64_t)r * capacity + c]);
    visited[c] = m;
}

// ---------------------------------------------------------------- data updates

__device__ inline uint16_t float_to_half_bits(float f) {
    return __half_as_ushort(__float2half_rn(f));
}

// cuda_renderer.cpp:262-266: data[capacity*8 + i][c] = mean_j results[i][j][c]
__global__ void split_mean_kernel(uint16_t *__restrict__ data, int64_t first_row, int64_t n_rows, const float *__restrict__ results,
                                  int32_t spc, int32_t data_dim, int32_t stride) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * data_dim) return;
    const int64_t i = idx / data_dim;
    const int c = (int)(idx - i * data_dim);
    const float *src = results + i * spc * stride + c;
    float sum = 0.f;
    for (int j = 0; j < spc; ++j) sum += src[(int64_t)j * stride];
    data[(first_row + i) * data_dim + c] = float_to_half_bits(sum / (float)spc);
}

__global__ void fill_i16_kernel(int16_t *__restrict__ p, int64_t n, int16_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// cuda_renderer.cpp:307-332: new average = old average + (sum of new - n_new * old average) / new count
__global__ void sample_average_kernel(uint16_t *__restrict__ data, const int16_t *__restrict__ sample_counts,
                                      const int32_t *__restrict__ nodes, int32_t n_items, const float *__restrict__ results,
                                      int32_t spc, int32_t data_dim, int32_t stride) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_items * data_dim) return;
    const int64_t i = idx / data_dim;
    const int c = (int)(idx - i * data_dim);
    const int64_t row = (int64_t)nodes[2 * i] * 8 + nodes[2 * i + 1];
    const float *src = results + i * spc * stride + c;
    float sum = 0.f;
    for (int j = 0; j < spc; ++j) sum += src[(int64_t)j * stride];
    const float old = half_bits_to_float(data[row * data_dim + c]);
    // `samples_per_corner * data` is a binary16 tensor in the reference expression (:322-324)
    const float scaled_old = half_bits_to_float(float_to_half_bits((float)spc * old));
    const int16_t new_count = (int16_t)(sample_counts[row] + (int16_t)spc);
    const float update = (sum - scaled_old) / (float)new_count;
    data[row * data_dim + c] = float_to_half_bits(old + update);
}

__global__ void bump_counts_kernel(int16_t *__restrict__ sample_counts, const int32_t *__restrict__ nodes, int32_t n_items, int16_t add) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const int64_t row = (int64_t)nodes[2 * i] * 8 + nodes[2 * i + 1];
    sample_counts[row] = (int16_t)(sample_counts[row] + add);
}

// ---------------------------------------------------------------- pruning

struct UnvisitedFlag {
    __device__ int32_t operator()(int32_t v) const { return v == 0 ? 1 : 0; }
};

__global__ void to_delete_kernel(const int32_t *__restrict__ visited, int32_t n, uint8_t *__restrict__ to_delete) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) to_delete[i] = visited[i] == 0;
}

// Gather the surviving rows of source chunks [s, e) into scratch (dense, in destination order).
__global__ void prune_gather_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ scratch, const uint8_t *__restrict__ to_delete,
                                    const int32_t *__restrict__ shifts, int32_t s, int32_t e, int32_t dest_base, int32_t row_words) {
    // one wavefront-sized group of threads per chunk row; rows are multiples of 4 bytes
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t chunk = s + (int32_t)(gid / row_words);
    const int32_t w = (int32_t)(gid % row_words);
    if (chunk >= e || to_delete[chunk]) return;
    const int32_t dest = chunk - shifts[chunk] - dest_base;
    reinterpret_cast<uint32_t *>(scratch)[(int64_t)dest * row_words + w] = reinterpret_cast<const uint32_t *>(src)[(int64_t)chunk * row_words + w];
}

// ---------------------------------------------------------------- uniform numbers, guided-sample compaction

__device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// out[i] = 24 random bits * 2^-24 in [0, 1); a pure function of (seed, i)
__global__ void fill_uniform_kernel(float *__restrict__ out, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = splitmix64(splitmix64(seed) ^ (uint64_t)i);
    out[i] = (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f;
}

struct CountToI64 {
    __device__ int64_t operator()(int16_t v) const { return (int64_t)v; }
};

// 16 lanes per ray (a few samples per ray is typical): rows [0, num_samples[ray]) of the ray's sample block -> position
// offsets[ray] - n
__global__ void compact_samples_kernel(const int16_t *__restrict__ num_samples, const int64_t *__restrict__ offsets,
                                       const float *__restrict__ samples, const int16_t *__restrict__ clusters, int64_t n_rays,
                                       int32_t max_samples, int32_t dim, float *__restrict__ z_vals, float *__restrict__ rows_out,
                                       int16_t *__restrict__ clusters_out) {
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int lane = threadIdx.x & 15;
    if (ray >= n_rays) return;
    const int n = num_samples[ray];
    if (n <= 0) return;
    const int64_t base = offsets[ray] - n;
    const float *src = samples + ray * (int64_t)max_samples * dim;
    const int cols = dim - 1;
    for (int i = lane; i < n; i += 16) {
        z_vals[base + i] = src[(int64_t)i * dim];
        clusters_out[base + i] = clusters[ray * (int64_t)max_samples + i];
    }
    for (int e = lane; e < n * cols; e += 16) {
        const int i = e / cols, c = e - i * cols;
        rows_out[(base + i) * cols + c] = src[(int64_t)i * dim + 1 + c];
    }
}

}  // namespace

}  // namespace mnv

using namespace mnv;

extern "C" {

int mnv_merge_visit_marks(const int32_t *table, int32_t world, int32_t capacity, int32_t *visited, void *hip_stream) {
    if (!table || !visited || world < 1 || capacity < 0) return set_error(MNV_E_INVALID, "invalid visit-mark arguments");
    if (capacity == 0) return MNV_OK;
    hipLaunchKernelGGL(merge_marks_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, table, world, capacity, visited);
    return check_hip(hipGetLastError(), "merge_marks_kernel");
}

int mnv_select_split_candidates(const float *split_track, int64_t n_rows, int32_t max_out, int32_t *nodes_out,
                                int32_t *n_out, int32_t *n_candidates, void *hip_stream) {
    return select_candidates(split_track, n_rows, max_out, true, nodes_out, n_out, n_candidates, (hipStream_t)hip_stream);
}

int mnv_select_sample_candidates(const float *sample_track, int64_t n_rows, int32_t max_out, int32_t *nodes_out,
                                 int32_t *n_out, int32_t *n_candidates, void *hip_stream) {
    return select_candidates(sample_track, n_rows, max_out, false, nodes_out, n_out, n_candidates, (hipStream_t)hip_stream);
}

int mnv_apply_split_results(uint16_t *data, int16_t *sample_counts, int32_t capacity, int32_t num_parents,
                            const float *results, int32_t result_stride, int32_t samples_per_corner, int32_t data_dim,
                            void *hip_stream) {
    if (!data || !results || capacity < 0 || num_parents < 0 || samples_per_corner < 1 || data_dim < 1 || result_stride < data_dim)
        return set_error(MNV_E_INVALID, "invalid split-result arguments");
    if (num_parents == 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    const int64_t rows = (int64_t)num_parents * 8, elems = rows * data_dim;
    hipLaunchKernelGGL(split_mean_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, data, (int64_t)capacity * 8, rows,
                       results, samples_per_corner, data_dim, result_stride);
    if (sample_counts)  // cuda_renderer.cpp:268-269
        hipLaunchKernelGGL(fill_i16_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, sample_counts + (int64_t)capacity * 8,
                           rows, (int16_t)samples_per_corner);
    return check_hip(hipGetLastError(), "apply_split_results");
}

int mnv_apply_sample_results(uint16_t *data, int16_t *sample_counts, const int32_t *nodes, int32_t num_items,
                             const float *results, int32_t result_stride, int32_t samples_per_corner, int32_t data_dim,
                             void *hip_stream) {
    if (!data || !sample_counts || !nodes || !results || num_items < 0 || samples_per_corner < 1 || data_dim < 1 ||
        result_stride < data_dim)
        return set_error(MNV_E_INVALID, "invalid sample-result arguments");
    if (num_items == 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    const int64_t elems = (int64_t)num_items * data_dim;
    hipLaunchKernelGGL(sample_average_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, data, sample_counts, nodes,
                       num_items, results, samples_per_corner, data_dim, result_stride);
    hipLaunchKernelGGL(bump_counts_kernel, dim3((num_items + 255) / 256), dim3(256), 0, stream, sample_counts, nodes, num_items,
                       (int16_t)samples_per_corner);
    return check_hip(hipGetLastError(), "apply_sample_results");
}

int mnv_prune_tree(const mnv_tree_edit *tree, uint16_t *data, int32_t data_dim, int16_t *sample_counts, int32_t *visited,
                   int32_t max_capacity, int32_t *new_capacity, int32_t *num_deleted, void *hip_stream) {
    return mnv_prune_tree_accel(tree, data, data_dim, sample_counts, visited, max_capacity, nullptr, new_capacity, num_deleted, hip_stream);
}

int mnv_prune_tree_accel(const mnv_tree_edit *tree, uint16_t *data, int32_t data_dim, int16_t *sample_counts, int32_t *visited,
                         int32_t max_capacity, mnv_accel *accel, int32_t *new_capacity, int32_t *num_deleted, void *hip_stream) {
    if (num_deleted) *num_deleted = 0;
    if (!tree || !tree->child || !tree->parent || !data || !visited || data_dim < 1 || tree->capacity < 1 ||
        max_capacity < tree->capacity)
        return set_error(MNV_E_INVALID, "invalid prune arguments");
    if (tree->N != 2) return set_error(MNV_E_UNSUPPORTED, "only N == 2 trees are supported");
    if ((data_dim * 8 * 2) % 4) return set_error(MNV_E_UNSUPPORTED, "chunk rows must be multiples of 4 bytes");
    hipStream_t stream = (hipStream_t)hip_stream;
    const int32_t cap = tree->capacity;
    if (new_capacity) *new_capacity = cap;

    constexpr int32_t kSegment = 1 << 18;  // chunks gathered per pass (the reference's PRUNE_CHUNK_SIZE role)
    const size_t data_row = (size_t)data_dim * 8 * 2;
    size_t tmp_scan = 0;
    {
        auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, UnvisitedFlag());
        (void)rocprim::inclusive_scan(nullptr, tmp_scan, in, (int32_t *)nullptr, (size_t)cap, rocprim::plus<int32_t>(), stream);
    }
    Carver c;
    const size_t o_del = c.take((size_t)cap), o_shift = c.take((size_t)cap * 4), o_tmp = c.take(tmp_scan);
    const size_t o_scratch = c.take((size_t)kSegment * data_row);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    uint8_t *ws = nullptr;
    int rc = ws_reserve(c.off, &ws);
    if (rc) return rc;
    uint8_t *to_delete = ws + o_del;
    int32_t *shifts = reinterpret_cast<int32_t *>(ws + o_shift);
    uint8_t *scratch = ws + o_scratch;

    // to_delete = visited[:capacity] == 0; index_shifts = cumsum(to_delete)   (cuda_renderer.cpp:337,348)
    hipLaunchKernelGGL(to_delete_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, visited, cap, to_delete);
    {
        auto in = rocprim::make_transform_iterator((const int32_t *)visited, UnvisitedFlag());
        size_t t = tmp_scan;
        if ((rc = check_hip(rocprim::inclusive_scan(ws + o_tmp, t, in, shifts, (size_t)cap, rocprim::plus<int32_t>(), stream), "inclusive_scan")))
            return rc;
    }
    int32_t n_del = 0, root_visited = 0;
    if ((rc = check_hip(hipMemcpyAsync(&n_del, shifts + (cap - 1), 4, hipMemcpyDeviceToHost, stream), "copy"))) return rc;
    if ((rc = check_hip(hipMemcpyAsync(&root_visited, visited, 4, hipMemcpyDeviceToHost, stream), "copy"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "prune scan"))) return rc;
    auto clear_marks = [&]() {  // cuda_renderer.cpp:343,378: everything but the root's mark
        return max_capacity > 1 ? check_hip(hipMemsetAsync(visited + 1, 0, (size_t)(max_capacity - 1) * 4, stream), "clear visit marks") : MNV_OK;
    };
    if (n_del == 0) return clear_marks();  // "Nothing can be pruned"
    if (!root_visited)
        return set_error(MNV_E_INVALID, "the root chunk is not marked visited: render a track_visit frame before pruning");

    // the packed layout follows first: its kernels read the old numbering of parent / data, which the next two steps rewrite
    if (accel && (rc = accel_apply_prune(accel, tree->parent, data, data_dim, to_delete, shifts, cap, n_del, stream))) return rc;

    // argmin of a non-decreasing cumsum is its first element: the reference's first_shift_index is 0
    // (cuda_renderer.cpp:350), which is also what the fix-up needs -- unshifted chunks can point at shifted ones
    if ((rc = mnv_adjust_parents_and_children(tree, 0, to_delete, shifts, hip_stream))) return rc;

    struct Array {
        uint8_t *base;
        size_t row;
    } arrays[4] = {{reinterpret_cast<uint8_t *>(data), data_row},
                   {reinterpret_cast<uint8_t *>(tree->child), 32},
                   {reinterpret_cast<uint8_t *>(tree->parent), 4},
                   {reinterpret_cast<uint8_t *>(sample_counts), 16}};
    // dest_base of a segment = index of its first source chunk minus the deletions before it; the host needs the
    // shift at every segment start
    const int32_t n_seg = (cap + kSegment - 1) / kSegment;
    std::vector<int32_t> seg_shift((size_t)n_seg + 1, 0);
    for (int32_t k = 1; k < n_seg; ++k)
        if ((rc = check_hip(hipMemcpyAsync(&seg_shift[k], shifts + ((int64_t)k * kSegment - 1), 4, hipMemcpyDeviceToHost, stream), "copy"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "prune segments"))) return rc;
    seg_shift[n_seg] = n_del;
    for (int a = 0; a < 4; ++a) {
        if (!arrays[a].base) continue;  // sample_counts is optional (the reference leaves it uncompacted)
        const int32_t row_words = (int32_t)(arrays[a].row / 4);
        for (int32_t k = 0; k < n_seg; ++k) {
            const int32_t s = k * kSegment, e = std::min(cap, s + kSegment);
            const int32_t survivors = (e - s) - (seg_shift[k + 1] - seg_shift[k]);
            if (survivors == 0 || seg_shift[k + 1] == 0) continue;  // nothing to move / nothing deleted up to here
            const int32_t dest_base = s - seg_shift[k];
            const int64_t threads = (int64_t)(e - s) * row_words;
            hipLaunchKernelGGL(prune_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, arrays[a].base, scratch,
                               to_delete, shifts, s, e, dest_base, row_words);
            if ((rc = check_hip(hipMemcpyAsync(arrays[a].base + (size_t)dest_base * arrays[a].row, scratch, (size_t)survivors * arrays[a].row,
                                               hipMemcpyDeviceToDevice, stream),
                                "prune copy")))
                return rc;
        }
    }
    if ((rc = clear_marks())) return rc;
    // the workspace is reused by the next call
    if ((rc = check_hip(hipStreamSynchronize(stream), "prune_tree"))) return rc;
    if (new_capacity) *new_capacity = cap - n_del;
    if (num_deleted) *num_deleted = n_del;
    return MNV_OK;
}

int mnv_fill_uniform(float *out, int64_t n, uint64_t seed, void *hip_stream) {
    if (n < 0 || (n > 0 && !out)) return set_error(MNV_E_INVALID, "invalid buffer");
    if (n == 0) return MNV_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, out, n, seed);
    return check_hip(hipGetLastError(), "fill_uniform_kernel");
}

int mnv_compact_guided_samples(const int16_t *num_samples, const float *samples, const int16_t *cluster_indices, int64_t n_rays,
                               int32_t max_guided_samples, int32_t samples_dim, int64_t *offsets_out, float *z_vals_out,
                               float *rows_out, int16_t *clusters_out, int64_t rows_capacity, int64_t *total_out,
                               void *hip_stream) {
    if (total_out) *total_out = 0;
    if (!num_samples || !samples || !cluster_indices || !offsets_out || n_rays < 0 || max_guided_samples < 1 || samples_dim < 2)
        return set_error(MNV_E_INVALID, "invalid guided-sample arguments");
    if (n_rays == 0) return MNV_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    auto in = rocprim::make_transform_iterator(num_samples, CountToI64());
    size_t tmp_bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, tmp_bytes, in, offsets_out, (size_t)n_rays, rocprim::plus<int64_t>(), stream);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    uint8_t *ws = nullptr;
    int rc = ws_reserve(tmp_bytes + 256, &ws);
    if (rc) return rc;
    // offsets = cumsum(num_samples)   (cuda_renderer.cpp:116)
    if ((rc = check_hip(rocprim::inclusive_scan(ws, tmp_bytes, in, offsets_out, (size_t)n_rays, rocprim::plus<int64_t>(), stream), "inclusive_scan")))
        return rc;
    int64_t total = 0;
    if ((rc = check_hip(hipMemcpyAsync(&total, offsets_out + (n_rays - 1), 8, hipMemcpyDeviceToHost, stream), "copy"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(stream), "compact_guided_samples"))) return rc;
    if (total_out) *total_out = total;
    if (total == 0 || (!z_vals_out && !rows_out && !clusters_out)) return MNV_OK;  // offsets / total only
    if (!z_vals_out || !rows_out || !clusters_out || total > rows_capacity)
        return set_error(MNV_E_INVALID, "output buffers are missing or smaller than the sample total");
    // the rows with z >= 0 in ray-major order (cuda_renderer.cpp:117-121), i.e. each ray's first num_samples rows
    const int64_t threads = n_rays * 16;
    hipLaunchKernelGGL(compact_samples_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, num_samples, offsets_out, samples,
                       cluster_indices, n_rays, max_guided_samples, samples_dim, z_vals_out, rows_out, clusters_out);
    return check_hip(hipGetLastError(), "compact_samples_kernel");
}

}  // extern "C"
