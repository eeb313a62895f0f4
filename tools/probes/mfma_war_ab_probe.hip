// Probe: a ds_read_b128 whose destination is the SrcA or SrcB register of an MFMA issued just before it (N wait states, K other MFMAs
// queued ahead) -- the register allocator emits exactly this (B tile of the next column half loaded over the one in use; the next block's
// weight fragment over the current one) and the hazard recogniser inserts nothing for it.  A = B = ones: every result must be 32; the load
// brings zeros, so a source read after the load's return shows as a smaller value.  (tools/probes/mfma_war_probe.hip tests SrcC; its
// SrcA case multiplied by a zero B and could not fail.)
// build: hipcc -O2 --offload-arch=gfx950 mfma_war_ab_probe.hip -o mfma_war_ab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N, int K, int WHICH>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    __shared__ float4 zeros[1024];
    zeros[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(&zeros[threadIdx.x]);
    unsigned long long wrong = 0, rows = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v40, 0x3c003c00\n\tv_mov_b32 v41, 0x3c003c00\n\tv_mov_b32 v42, 0x3c003c00\n\tv_mov_b32 v43, 0x3c003c00\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t"
            ".rept %c6\n\t"
            "v_mfma_f32_16x16x32_f16 v[72:75], v[56:59], v[56:59], v[60:63]\n\t"
            ".endr\n\t"
            ".if %c7 == 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[40:43], v[56:59], v[60:63]\n\t"  // SrcA = v[40:43]
            ".else\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[56:59], v[40:43], v[60:63]\n\t"  // SrcB = v[40:43]
            ".endif\n\t"
            ".if %c5 > 0\n\t"
            "s_nop %c5 - 1\n\t"
            ".endif\n\t"
            "ds_read_b128 v[40:43], %4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
            : "v"(addr), "n"(N), "n"(K), "n"(WHICH)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v72", "v73", "v74", "v75", "memory");
        const int w = (d0 != 32.f) + (d1 != 32.f) + (d2 != 32.f) + (d3 != 32.f);
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int N, int K, int WHICH>
void run(int waves_per_simd, int blocks) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N, K, WHICH>), dim3(blocks), dim3(256 * waves_per_simd), 0, 0, bad, 100000);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("overwritten %s  wait states %2d  MFMAs queued ahead %2d  waves/SIMD %d  workgroups %4d : wrong values %llu (lane rows mask %llx)\n", WHICH ? "SrcB" : "SrcA", N, K,
           waves_per_simd, blocks, h[0], h[1]);
    (void)hipFree(bad);
}

template <int WHICH>
void sweep(int w, int blocks) {
    run<0, 0, WHICH>(w, blocks); run<0, 1, WHICH>(w, blocks); run<0, 3, WHICH>(w, blocks); run<0, 7, WHICH>(w, blocks); run<0, 15, WHICH>(w, blocks);
    run<2, 7, WHICH>(w, blocks); run<8, 7, WHICH>(w, blocks); run<16, 15, WHICH>(w, blocks);
}

int main() {
    for (int blocks = 1; blocks <= 256; blocks *= 256)
        for (int w = 1; w <= 4; w *= 2) {
            sweep<0>(w, blocks);
            sweep<1>(w, blocks);
        }
    return 0;
}
