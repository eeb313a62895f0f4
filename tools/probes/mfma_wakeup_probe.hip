// Probe: does the FIRST v_mfma_f32_16x16x32_f16 after the matrix pipe of a SIMD sat idle for T cycles deliver its result later than
// the 8 wait states the compiler's hazard table (and tools/probes/mfma_raw_probe.hip, measured with the pipe busy) allow before a VALU
// read?  One wavefront per SIMD (nothing else keeps the pipe awake); idle = s_nop loops or s_sleep; then MFMA, N wait states, read.
// A = B = ones, C = lane: result 32 + lane; the destination holds 1000 beforehand.
// build: hipcc -O2 --offload-arch=gfx950 mfma_wakeup_probe.hip -o mfma_wakeup_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N, int CHAIN>
__global__ __launch_bounds__(256) void probe(unsigned long long *bad, int iters, int idle_loops, int use_sleep) {
    unsigned long long wrong = 0, rows = 0;
    const float c = (float)(threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < idle_loops; ++k) {
            if (use_sleep) __builtin_amdgcn_s_sleep(8);
            else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        }
        float d;
        asm volatile(
            "v_mov_b32 v60, %1\n\tv_mov_b32 v61, %1\n\tv_mov_b32 v62, %1\n\tv_mov_b32 v63, %1\n\t"
            "v_mov_b32 v44, 0x447a0000\n\tv_mov_b32 v45, 0x447a0000\n\tv_mov_b32 v46, 0x447a0000\n\tv_mov_b32 v47, 0x447a0000\n\t"
            "v_mov_b32 v56, 0x3c003c00\n\tv_mov_b32 v57, 0x3c003c00\n\tv_mov_b32 v58, 0x3c003c00\n\tv_mov_b32 v59, 0x3c003c00\n\t"
            "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
            "s_nop 15\n\t"
            ".rept %c3\n\t"
            "v_mfma_f32_16x16x32_f16 v[72:75], v[56:59], v[56:59], v[68:71]\n\t"
            ".endr\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[56:59], v[56:59], v[60:63]\n\t"
            ".if %c2 > 0\n\t"
            "s_nop %c2 - 1\n\t"
            ".endif\n\t"
            "v_mov_b32 %0, v47\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            : "=v"(d)
            : "v"(c), "n"(N), "n"(CHAIN)
            : "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "memory");
        const int w = d != 32.f + c;
        wrong += w;
        if (w) rows |= 1ull << ((threadIdx.x & 63) >> 4);
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicOr(bad + 1, rows);
    }
}

template <int N, int CHAIN>
void run(int idle_loops, int use_sleep, int blocks) {
    unsigned long long *bad, h[2] = {0, 0};
    (void)hipMalloc(&bad, 16);
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N, CHAIN>), dim3(blocks), dim3(256), 0, 0, bad, 2000, idle_loops, use_sleep);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("idle %6d x %s  MFMAs ahead in the burst %2d  wait states %2d  workgroups %3d : stale values %llu (lane rows mask %llx)\n", idle_loops, use_sleep ? "s_sleep 8 " : "64 nop clk", CHAIN, N,
           blocks, h[0], h[1]);
    (void)hipFree(bad);
}

int main() {
    const int idles[] = {0, 1, 4, 16, 64, 256, 1024};
    for (int blocks = 1; blocks <= 256; blocks *= 256)
        for (int s = 0; s < 2; ++s)
            for (int idle : idles) {
                run<8, 0>(idle, s, blocks);
                run<8, 15>(idle, s, blocks);
                run<9, 0>(idle, s, blocks);
            }
    return 0;
}
