// Probe: S stores and 12 loads (ds_*_b128) in flight at once, then s_waitcnt lgkmcnt(9) and a use of the first three loads' registers --
// the shape the compiler gives layer 0 of the network in mnv_guided_fused2.h.  lgkmcnt is a 4-bit counter on gfx9: does the wait still hold
// when more than 15 LDS operations are outstanding?  The registers hold a marker beforehand; counts lanes that still saw the marker.
// build: hipcc -O2 --offload-arch=gfx950 lgkmcnt_probe.hip -o lgkmcnt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int STORES>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    extern __shared__ float4 lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *mine = lds + wave * (64 * 16);
    for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = make_float4(1.f, 1.f, 1.f, 1.f);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(mine + lane);
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float a, b, c;
        asm volatile(
            "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b32 v44, 0x447a0000\n\tv_mov_b32 v48, 0x447a0000\n\tv_mov_b32 v52, 0x447a0000\n\t"
            ".rept %c4\n\t"
            "ds_write_b128 %3, v[40:43] offset:15360\n\t"
            ".endr\n\t"
            "ds_read_b128 v[44:47], %3\n\t"
            "ds_read_b128 v[48:51], %3 offset:1024\n\t"
            "ds_read_b128 v[52:55], %3 offset:2048\n\t"
            "ds_read_b128 v[56:59], %3 offset:3072\n\t"
            "ds_read_b128 v[60:63], %3 offset:4096\n\t"
            "ds_read_b128 v[64:67], %3 offset:5120\n\t"
            "ds_read_b128 v[68:71], %3 offset:6144\n\t"
            "ds_read_b128 v[72:75], %3 offset:7168\n\t"
            "ds_read_b128 v[76:79], %3 offset:8192\n\t"
            "ds_read_b128 v[80:83], %3 offset:9216\n\t"
            "ds_read_b128 v[84:87], %3 offset:10240\n\t"
            "ds_read_b128 v[88:91], %3 offset:11264\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v48\n\tv_mov_b32 %2, v52\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : "=v"(a), "=v"(b), "=v"(c)
            : "v"(addr), "n"(STORES)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62",
              "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85",
              "v86", "v87", "v88", "v89", "v90", "v91", "memory");
        wrong += (a != 1.f) + (b != 1.f) + (c != 1.f);
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int STORES>
void run(int waves) {
    unsigned long long *bad, h = 0;
    (void)hipMalloc(&bad, 8);
    (void)hipMemset(bad, 0, 8);
    const size_t lds = (size_t)waves * 64 * 16 * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<STORES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((probe<STORES>), dim3(256), dim3(64 * waves), lds, 0, bad, 200000);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("stores ahead %d  loads 12  waves/CU %2d : stale values after s_waitcnt lgkmcnt(9): %llu\n", STORES, waves, h);
    (void)hipFree(bad);
}

int main() {
    for (int w = 1; w <= 8; w *= 2) {
        run<0>(w); run<3>(w); run<4>(w); run<6>(w); run<8>(w);
    }
    return 0;
}
