// Stand-in for an RCCL channel workgroup in the CU-mask experiments (tools/cumask/probe.py): 256 threads that hold
// `lds_bytes` of LDS (RCCL's generic kernel on gfx950: ~20 KB LDS and ~288 VGPRs per lane, i.e. it cannot become resident on a
// compute unit that already holds more than three of the march kernel's 64-VGPR workgroups) and spin for `usec`.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC spin.hip -o libspin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void spin_kernel(uint64_t ticks, uint32_t *sink) {
    extern __shared__ uint32_t s_hold[];
    s_hold[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint64_t t0 = wall_clock64();  // 100 MHz
    uint32_t acc = 0;
    while (wall_clock64() - t0 < ticks) acc += s_hold[(threadIdx.x + acc) & 255u];
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

extern "C" int spin_launch(void *stream, int n_blocks, int lds_bytes, int usec, void *sink) {
    static int configured = 0;
    if (lds_bytes > 65536 && configured < lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = lds_bytes;
    }
    hipLaunchKernelGGL(spin_kernel, dim3(n_blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (uint64_t)usec * 100ull, (uint32_t *)sink);
    return (int)hipGetLastError();
}

extern "C" int masked_stream_create(void **stream, int n_words, const uint32_t *mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask);
    *stream = (void *)s;
    return (int)e;
}

// where does a workgroup run?  out[block] = XCC_ID << 16 | HW_ID[15:0] (gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])
__global__ void whoami_kernel(uint32_t *out, uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        out[blockIdx.x] = (xcc << 16) | hw;
    }
}

extern "C" int whoami_launch(void *stream, int n_blocks, int usec, void *out) {
    hipLaunchKernelGGL(whoami_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, (uint32_t *)out, (uint64_t)usec * 100ull);
    return (int)hipGetLastError();
}

extern "C" int stream_destroy(void *stream) { return (int)hipStreamDestroy((hipStream_t)stream); }
