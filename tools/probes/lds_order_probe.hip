// Probe: does the LDS serve ONE wavefront's instructions in program order when a later ds_read (other lanes' addresses) follows its own
// ds_write2_b32 without a wait in between?  The out tile of mnv_guided_fused2.h relies on it: every lane stores 16 accumulator values
// ([row][32 columns], two column tiles per ds_write2_b32), then -- after a wave barrier that is only a compiler barrier -- lanes read
// nine rows of ONE column that other lanes wrote.  Here: 8 stores of values derived from the iteration number, then 9 loads of rows
// written by other lane groups, no s_waitcnt between; every value is checked.  W wavefronts per workgroup share the CU's LDS (each
// its own 4 KB tile), so the stores and loads of different wavefronts interleave in the LDS queue.
// build: hipcc -O2 --offload-arch=gfx950 lds_order_probe.hip -o lds_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int GAP>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters, int pad_words) {
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *tile = lds + pad_words + wave * 1024;  // [32 rows][32 columns]
    const int g = lane >> 4, col = lane & 15, c32 = lane & 31, share = lane >> 5;
    const uint32_t st = (uint32_t)(uintptr_t)(tile + (4 * g) * 32 + col);            // rows 4g + r (+16 for the second block), columns col and 16 + col
    const uint32_t ld = (uint32_t)(uintptr_t)(tile + (share ? 9 : 0) * 32 + c32);    // nine rows from 0 or 9, column c32
    unsigned long long wrong = 0;
    for (int it = 1; it <= iters; ++it) {
        // value of (row, column) in iteration it: it * 4096 + row * 32 + column
        uint32_t v[16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[(mt * 4 + r) * 2 + 0] = (uint32_t)it * 4096u + (uint32_t)(16 * mt + 4 * g + r) * 32u + (uint32_t)col;
                v[(mt * 4 + r) * 2 + 1] = (uint32_t)it * 4096u + (uint32_t)(16 * mt + 4 * g + r) * 32u + 16u + (uint32_t)col;
            }
        uint32_t o[9];
        const uint32_t st1 = st + 2048u;  // rows + 16
        asm volatile(
            "ds_write2_b32 %5, %8, %9 offset1:16\n\t"
            "ds_write2_b32 %5, %10, %11 offset0:32 offset1:48\n\t"
            "ds_write2_b32 %5, %12, %13 offset0:64 offset1:80\n\t"
            "ds_write2_b32 %5, %14, %15 offset0:96 offset1:112\n\t"
            "ds_write2_b32 %6, %16, %17 offset1:16\n\t"
            "ds_write2_b32 %6, %18, %19 offset0:32 offset1:48\n\t"
            "ds_write2_b32 %6, %20, %21 offset0:64 offset1:80\n\t"
            "ds_write2_b32 %6, %22, %23 offset0:96 offset1:112\n\t"
            ".if %c24 == 1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            ".endif\n\t"
            "ds_read2_b32 %0, %7 offset1:32\n\t"
            "ds_read2_b32 %1, %7 offset0:64 offset1:96\n\t"
            "ds_read2_b32 %2, %7 offset0:128 offset1:160\n\t"
            "ds_read2_b32 %3, %7 offset0:192 offset1:224\n\t"
            "ds_read_b32 %4, %7 offset:1024\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : "=&v"(*(uint64_t *)&o[0]), "=&v"(*(uint64_t *)&o[2]), "=&v"(*(uint64_t *)&o[4]), "=&v"(*(uint64_t *)&o[6]), "=&v"(o[8])
            : "v"(st), "v"(st1), "v"(ld), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
              "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "n"(GAP)
            : "memory");
        int w = 0;
#pragma unroll
        for (int f = 0; f < 9; ++f) w += o[f] != (uint32_t)it * 4096u + (uint32_t)((share ? 9 : 0) + f) * 32u + (uint32_t)c32;
        wrong += (unsigned)w;
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int GAP>
void run(int waves, int pad_kb) {
    unsigned long long *bad, h = 0;
    (void)hipMalloc(&bad, 8);
    (void)hipMemset(bad, 0, 8);
    const size_t lds = (size_t)pad_kb * 1024 + (size_t)waves * 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<GAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((probe<GAP>), dim3(256), dim3(64 * waves), lds, 0, bad, 200000, pad_kb * 256);
    hipError_t e = hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("wait between stores and loads: %s  wavefronts/CU %2d  tiles start at %3d KB : wrong values %llu%s\n", GAP ? "yes" : "no ", waves, pad_kb, h, e == hipSuccess ? "" : "  (launch failed)");
    (void)hipFree(bad);
}

int main() {
    for (int pad = 0; pad <= 96; pad += 48)
        for (int w = 1; w <= 16; w *= 2) {
            run<0>(w, pad);
            run<1>(w, pad);
        }
    return 0;
}
