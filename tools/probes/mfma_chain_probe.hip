// Probe: the four-MFMA pattern the compiler emits for one 16-row block of a layer in mnv_guided_fused2.h / mnv_mlp.hip
//     T  = A0 * B0 + C          (T: a temporary; C: the bias)
//     C  = A0 * B1 + C          (in place)
//     A0 = A1 * B0' + T         (SrcC = the destination of an MFMA two instructions back, destination = registers that were SrcA)
//     C  = A1 * B1' + C
// run back to back by 1, 2 or 4 wavefronts per SIMD, with N wait states before the third instruction.  All inputs are ones (K = 32) and the
// bias is the lane number, so every result must be 64 + lane; T holds 1000 beforehand.  Prints how many values differed and in which lane rows.
// build: hipcc -O2 --offload-arch=gfx950 mfma_chain_probe.hip -o mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    unsigned long long wrong = 0, rows = 0;
    const float c = (float)(threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3, u0, u1, u2, u3;
        asm volatile(
            "v_mov_b32 v60, %8\n\tv_mov_b32 v61, %8\n\tv_mov_b32 v62, %8\n\tv_mov_b32 v63, %8\n\t"
            "v_mov_b32 v92, 0x447a0000\n\tv_mov_b32 v93, 0x447a0000\n\tv_mov_b32 v94, 0x447a0000\n\tv_mov_b32 v95, 0x447a0000\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v64, 0x3c003c00\n\tv_mov_b32 v65, 0x3c003c00\n\tv_mov_b32 v66, 0x3c003c00\n\tv_mov_b32 v67, 0x3c003c00\n\t"
            "v_mov_b32 v68, 0x3c003c00\n\tv_mov_b32 v69, 0x3c003c00\n\tv_mov_b32 v70, 0x3c003c00\n\tv_mov_b32 v71, 0x3c003c00\n\t"
            "v_mov_b32 v72, 0x3c003c00\n\tv_mov_b32 v73, 0x3c003c00\n\tv_mov_b32 v74, 0x3c003c00\n\tv_mov_b32 v75, 0x3c003c00\n\t"
            "v_mov_b32 v76, 0x3c003c00\n\tv_mov_b32 v77, 0x3c003c00\n\tv_mov_b32 v78, 0x3c003c00\n\tv_mov_b32 v79, 0x3c003c00\n\t"
            "v_mov_b32 v110, 0x3c003c00\n\tv_mov_b32 v111, 0x3c003c00\n\tv_mov_b32 v112, 0x3c003c00\n\tv_mov_b32 v113, 0x3c003c00\n\t"
            "s_nop 15\n\t"
            "v_mfma_f32_16x16x32_f16 v[92:95], v[56:59], v[64:67], v[60:63]\n\t"
            "v_mfma_f32_16x16x32_f16 v[60:63], v[56:59], v[68:71], v[60:63]\n\t"
            ".if %c9 > 0\n\t"
            "s_nop %c9 - 1\n\t"
            ".endif\n\t"
            "v_mfma_f32_16x16x32_f16 v[56:59], v[110:113], v[72:75], v[92:95]\n\t"
            "v_mfma_f32_16x16x32_f16 v[60:63], v[110:113], v[76:79], v[60:63]\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            "v_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3)
            : "v"(c), "n"(N)
            : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
              "v78", "v79", "v92", "v93", "v94", "v95", "v110", "v111", "v112", "v113", "memory");
        // the bias of output row r sits in lane group r / 4; every lane of the group holds c of ITS lane (the probe uses the lane number, so each
        // lane expects 64 + its own c)
        const float want = 64.f + c;
        const int w = (d0 != want) + (d1 != want) + (d2 != want) + (d3 != want) + (u0 != want) + (u1 != want) + (u2 != want) + (u3 != want);
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int N>
void run(int waves_per_simd) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N>), dim3(512), dim3(256 * waves_per_simd), 0, 0, bad, 100000);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("nops before the third MFMA %2d  waves/SIMD %d : wrong values %llu  (lane rows mask %llx)\n", N, waves_per_simd, h[0], h[1]);
    (void)hipFree(bad);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>(w); run<1>(w); run<2>(w); run<4>(w); run<8>(w); run<16>(w);
    }
    return 0;
}
