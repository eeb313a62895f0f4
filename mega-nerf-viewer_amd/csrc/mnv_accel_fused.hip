// mnv_accel_fused.hip -- the instantiations of the fused guided-sampling kernels and the choice among them.
#include <atomic>
#include <cstdlib>

#include "mnv_guided_fused.h"   // guided_fused_kernel: march + per-sample network + composite in one kernel, every wavefront both roles
#include "mnv_guided_fused2.h"  // guided_fused2_kernel: the same frame with producer (march) and consumer (network) wavefronts
#include "mnv_knobs.h"

namespace mnv {


int launch_fused(const mnv_accel *accel, const AccelLaunch &K, const FusedGuided &fused, int b, int lds_level, uint64_t n_waves_needed,
                 hipStream_t stream) {
    int rc = kUnsupportedBasis;
    const int nb = b > 0 ? b : 1;
    const bool two = fused.S.nkk0 == 2;  // mnv_render_guided_fused admits 1 and 2
    const bool trk = K.split_track || K.sample_track || K.visited;
    // producer / consumer wavefronts when one sub-module's weights fit a workgroup's LDS beside the rings (at least two workgroups per CU)
    FusedGuided F = fused;
    F.fault = accel->fault_dev;
    int slots = kF2NS;  // weight slots: as many as fit beside the rings (at least one per two consumers)
    const int f2_per_cu = (4 * kF2WavesPerSimd) / (kF2NP + kF2NC);  // workgroups per CU the kernel is built for
    while (slots > 1 && (size_t)f2_layout(nb, lds_level, F.S, slots).total * 4 > (size_t)160 * 1024 / f2_per_cu) --slots;
    F.weight_slots = slots;
    const size_t f2_bytes = (size_t)f2_layout(nb, lds_level, F.S, slots).total * 4;
    const int version = accel->fused_kernel.load(std::memory_order_relaxed);
    const bool fits2 = f2_bytes <= (size_t)160 * 1024 / f2_per_cu && slots >= (kF2NS < 2 ? kF2NS : 2) && F.S.bias_floats <= 256;  // (a consumer refills a sub-module's biases with four loads per lane)
    if (fits2 && (version == 2 || (version == 0 && kF2Default))) {
        int per_cu = (int)((size_t)160 * 1024 / f2_bytes);
        const int by_regs = (4 * kF2WavesPerSimd) / (kF2NP + kF2NC);
        if (per_cu > by_regs) per_cu = by_regs;
        static const int env_f2 = knob_int(KNOB_F2_BLOCKS_PER_CU, 0);
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
    return rc;
}

}  // namespace mnv

