// Probe: how many wait states after a v_mfma_f32_16x16x32_f16 may a VALU instruction (or the data read of a ds_write) read its result,
// alone and with other wavefronts on the SIMD issuing MFMAs too, with K MFMAs of the same wavefront queued ahead (the last of a dependent
// chain, as at the end of a layer)?  The compiler separates them by its table value (s_nop 2 after a chain in mnv_guided_fused2.h's listing
// plus the instructions in between); this finds the smallest N at which no stale value is seen, per configuration.
// A = B = ones (K = 32), C = lane: result 32 + lane; the destination holds 1000 beforehand.
// build: hipcc -O2 --offload-arch=gfx950 mfma_raw_probe.hip -o mfma_raw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N, int K, int READER>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    __shared__ float spill[1024];
    unsigned long long wrong = 0, rows = 0;
    const float c = (float)(threadIdx.x & 63);
    const uint32_t addr = (uint32_t)(uintptr_t)(&spill[threadIdx.x]);
    for (int it = 0; it < iters; ++it) {
        float d;
        asm volatile(
            "v_mov_b32 v60, %1\n\tv_mov_b32 v61, %1\n\tv_mov_b32 v62, %1\n\tv_mov_b32 v63, %1\n\t"
            "v_mov_b32 v44, 0x447a0000\n\tv_mov_b32 v45, 0x447a0000\n\tv_mov_b32 v46, 0x447a0000\n\tv_mov_b32 v47, 0x447a0000\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v64, 0x3c003c00\n\tv_mov_b32 v65, 0x3c003c00\n\tv_mov_b32 v66, 0x3c003c00\n\tv_mov_b32 v67, 0x3c003c00\n\t"
            "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            ".rept %c4\n\t"
            "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[64:67], v[68:71]\n\t"
            ".endr\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[56:59], v[64:67], v[60:63]\n\t"
            ".if %c3 > 0\n\t"
            "s_nop %c3 - 1\n\t"
            ".endif\n\t"
            ".if %c5 == 0\n\t"
            "v_mov_b32 %0, v47\n\t"
            ".else\n\t"
            "ds_write_b32 %2, v47\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_read_b32 %0, %2\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            ".endif\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            : "=v"(d)
            : "v"(c), "v"(addr), "n"(N), "n"(K), "n"(READER)
            : "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
              "v75", "memory");
        const int w = d != 32.f + c;
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int N, int K, int READER>
void run(int waves_per_simd) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N, K, READER>), dim3(512), dim3(256 * waves_per_simd), 0, 0, bad, 50000);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("reader %s  wait states %2d  MFMAs queued ahead %d  waves/SIMD %d : stale values %llu (lane rows mask %llx)\n", READER ? "ds_write" : "v_mov   ", N, K, waves_per_simd, h[0],
           h[1]);
    (void)hipFree(bad);
}

template <int K, int READER>
void sweep(int w) {
    run<0, K, READER>(w); run<2, K, READER>(w); run<4, K, READER>(w); run<5, K, READER>(w); run<6, K, READER>(w); run<7, K, READER>(w); run<8, K, READER>(w);
    run<10, K, READER>(w); run<12, K, READER>(w); run<16, K, READER>(w);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        sweep<0, 0>(w); sweep<3, 0>(w); sweep<0, 1>(w); sweep<3, 1>(w);
    }
    return 0;
}
