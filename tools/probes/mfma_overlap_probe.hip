// Probe: v_mfma_f32_16x16x32_f16 whose destination registers are its own SrcA (or SrcB) registers -- the register allocator emits this
// (the 128-bit destination is not marked early-clobber) -- run back to back by 1, 2 or 4 wavefronts per SIMD, optionally behind K other
// MFMAs of the same wavefront.  A = B = ones (K = 32), C = lane number: every result must be 32 + lane.
// build: hipcc -O2 --offload-arch=gfx950 mfma_overlap_probe.hip -o mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WHICH, int K>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    unsigned long long wrong = 0, rows = 0;
    const float c = (float)(threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v60, %4\n\tv_mov_b32 v61, %4\n\tv_mov_b32 v62, %4\n\tv_mov_b32 v63, %4\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v64, 0x3c003c00\n\tv_mov_b32 v65, 0x3c003c00\n\tv_mov_b32 v66, 0x3c003c00\n\tv_mov_b32 v67, 0x3c003c00\n\t"
            "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
            "s_nop 15\n\t"
            ".rept %c6\n\t"
            "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[64:67], v[68:71]\n\t"
            ".endr\n\t"
            ".if %c5 == 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[56:59], v[56:59], v[64:67], v[60:63]\n\t"  // destination = SrcA
            ".else\n\t"
            "v_mfma_f32_16x16x32_f16 v[56:59], v[64:67], v[56:59], v[60:63]\n\t"  // destination = SrcB
            ".endif\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
            : "v"(c), "n"(WHICH), "n"(K)
            : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "memory");
        const float want = 32.f + c;
        const int w = (d0 != want) + (d1 != want) + (d2 != want) + (d3 != want);
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int WHICH, int K>
void run(int waves_per_simd) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<WHICH, K>), dim3(512), dim3(256 * waves_per_simd), 0, 0, bad, 100000);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("destination = %s  MFMAs queued ahead %2d  waves/SIMD %d : wrong values %llu  (lane rows mask %llx)\n", WHICH ? "SrcB" : "SrcA", K, waves_per_simd, h[0], h[1]);
    (void)hipFree(bad);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0>(w); run<0, 1>(w); run<0, 4>(w);
        run<1, 0>(w); run<1, 1>(w); run<1, 4>(w);
    }
    return 0;
}
