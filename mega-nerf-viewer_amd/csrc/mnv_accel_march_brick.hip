// mnv_accel_march_brick.hip -- the instantiations of march_accel_kernel on inline cell words (AccelView::grid2i) and brick records
// (AccelView::recs: the two levels below the second lookup grid from one 64-byte record per chunk): plain, fast-colour, depth, tracker and
// sample frames of the per-lane row formats (RGBA, SH1 / 4 / 9), and the diagnostics instantiation of the SH9 kernel.  SH16 / SH25 trees
// and frames with a negative sigma_thresh walk the node words (mnv_accel_march.hip): every other lookup array says the same.
#include "mnv_march_accel_kernel.h"

namespace mnv {

template <int BASIS, int MODE>
static int launch_brick2(const AccelLaunch &K, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    constexpr int BLOCK = 256;
    auto kern = march_accel_kernel<BASIS, BLOCK, MODE, true>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(BLOCK), lds_bytes, stream, K);
    return (int)hipGetLastError();
}

template <int BASIS>
static int launch_brick(const AccelLaunch &K, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    if constexpr (BASIS == 9) {
        if (K.stats) return launch_brick2<BASIS, 1>(K, n_blocks, lds_bytes, stream);
        if (K.samples) return launch_brick2<BASIS, 3>(K, n_blocks, lds_bytes, stream);  // (reads no colour rows: serves every row format)
    }
    if (K.split_track || K.sample_track || K.visited) return launch_brick2<BASIS, 2>(K, n_blocks, lds_bytes, stream);
    if constexpr (BASIS == 9) {
        if (K.P.render_depth) return launch_brick2<BASIS, 5>(K, n_blocks, lds_bytes, stream);
    }
    if constexpr (BASIS >= 1) {
        if (K.fast_colour) return launch_brick2<BASIS, 4>(K, n_blocks, lds_bytes, stream);
    }
    return launch_brick2<BASIS, 0>(K, n_blocks, lds_bytes, stream);
}

int launch_march_brick(const AccelLaunch &K, int b, bool colourless, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    if (colourless) return launch_brick<9>(K, n_blocks, lds_bytes, stream);
    switch (b) {
        case -1: return launch_brick<-1>(K, n_blocks, lds_bytes, stream);
        case 1: return launch_brick<1>(K, n_blocks, lds_bytes, stream);
        case 4: return launch_brick<4>(K, n_blocks, lds_bytes, stream);
        case 9: return launch_brick<9>(K, n_blocks, lds_bytes, stream);
        default: break;
    }
    return kUnsupportedBasis;
}

}  // namespace mnv
