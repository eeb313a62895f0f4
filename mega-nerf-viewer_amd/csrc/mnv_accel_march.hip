// mnv_accel_march.hip -- the instantiations of march_accel_kernel and the choice among them.
#include <atomic>

#include "mnv_march_accel_kernel.h"

namespace mnv {

template <int BASIS, int MODE>
static int launch_variant2(const AccelLaunch &K, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    constexpr int BLOCK = 256;
    auto kern = march_accel_kernel<BASIS, BLOCK, MODE, false>;
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
        if (K.fast_colour) return launch_variant2<BASIS, 4>(K, n_blocks, lds_bytes, stream);
    }
    return launch_variant2<BASIS, 0>(K, n_blocks, lds_bytes, stream);
}

int launch_march(const AccelLaunch &K, int b, bool colourless, int n_blocks, size_t lds_bytes, hipStream_t stream) {
    if (colourless) return launch_variant<9>(K, n_blocks, lds_bytes, stream);
    switch (b) {
        case -1: return launch_variant<-1>(K, n_blocks, lds_bytes, stream);
        case 1: return launch_variant<1>(K, n_blocks, lds_bytes, stream);
        case 4: return launch_variant<4>(K, n_blocks, lds_bytes, stream);
        case 9: return launch_variant<9>(K, n_blocks, lds_bytes, stream);
        case 16: return launch_variant<16>(K, n_blocks, lds_bytes, stream);
        case 25: return launch_variant<25>(K, n_blocks, lds_bytes, stream);
        default: break;
    }
    return kUnsupportedBasis;
}

}  // namespace mnv

