// Probe: s_sleep issued right after a v_mfma_f32_16x16x32_f16 (0..3 instructions later), alone or with ds_read_b128 loads in flight
// whose data feeds the NEXT MFMAs.  Found while bisecting mnv_guided_fused2.h's listing by hand: an s_sleep 2 inserted between the
// first MFMA of layer 0 and the s_waitcnt of the second made every frame wrong.
// A = B = ones, C = lane: first result 32 + lane; second MFMA accumulates a tile loaded from LDS (ones) on top: 64 + lane.
// build: hipcc -O2 --offload-arch=gfx950 mfma_sleep_probe.hip -o mfma_sleep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int SLEEP, int GAP, int LOADS>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    __shared__ uint4 ones[1024];
    ones[threadIdx.x] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(&ones[threadIdx.x]);
    const float c = (float)(threadIdx.x & 63);
    unsigned long long wrong = 0, rows = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v60, %4\n\tv_mov_b32 v61, %4\n\tv_mov_b32 v62, %4\n\tv_mov_b32 v63, %4\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
            "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t"
            ".if %c8 == 1\n\t"
            "ds_read_b128 v[64:67], %5\n\t"          // the second MFMA's A (ones), in flight across the sleep
            "ds_read_b128 v[68:71], %5\n\t"          // and its B
            ".else\n\t"
            "v_mov_b32 v64, 0x3c003c00\n\tv_mov_b32 v65, 0x3c003c00\n\tv_mov_b32 v66, 0x3c003c00\n\tv_mov_b32 v67, 0x3c003c00\n\t"
            "v_mov_b32 v68, 0x3c003c00\n\tv_mov_b32 v69, 0x3c003c00\n\tv_mov_b32 v70, 0x3c003c00\n\tv_mov_b32 v71, 0x3c003c00\n\t"
            "s_nop 4\n\t"
            ".endif\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[56:59], v[56:59], v[60:63]\n\t"   // 32 + lane
            ".if %c7 > 0\n\t"
            "s_nop %c7 - 1\n\t"
            ".endif\n\t"
            ".if %c6 >= 0\n\t"
            "s_sleep %c6\n\t"
            ".endif\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[64:67], v[68:71], v[44:47]\n\t"   // + 32
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
            : "v"(c), "v"(addr), "n"(SLEEP), "n"(GAP), "n"(LOADS)
            : "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "memory");
        const float want = 64.f + c;
        const int w = (d0 != want) + (d1 != want) + (d2 != want) + (d3 != want);
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int SLEEP, int GAP, int LOADS>
void run(int waves_per_simd) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<SLEEP, GAP, LOADS>), dim3(256), dim3(256 * waves_per_simd), 0, 0, bad, 20000);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("s_sleep %2d  instructions after the MFMA %d  loads in flight %d  waves/SIMD %d : wrong values %llu (lane rows mask %llx)\n", SLEEP, GAP, LOADS, waves_per_simd, h[0], h[1]);
    (void)hipFree(bad);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<-1, 0, 0>(w); run<-1, 0, 1>(w);
        run<0, 0, 0>(w); run<0, 0, 1>(w); run<2, 0, 0>(w); run<2, 0, 1>(w); run<2, 1, 1>(w); run<2, 3, 1>(w); run<2, 8, 1>(w); run<8, 0, 1>(w);
    }
    return 0;
}
