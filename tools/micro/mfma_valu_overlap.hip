// Microbenchmark behind DESIGN.md 5.6 / LAB_NOTEBOOK round 6: what two wavefronts of one SIMD can overlap on gfx950.
// One workgroup of 8 wavefronts per compute unit (100 KB of LDS keeps a second one out): wavefront w and w + 4 share SIMD w & 3.
// A "unit" is 64 MFMAs (v_mfma_f32_16x16x32_f16 on 4 independent accumulators) or 256 VALU instructions (independent FMAs).
//   mode 0: every wavefront MFMA only            mode 1: every wavefront VALU only
//   mode 2: wavefronts 0-3 MFMA, 4-7 VALU        mode 3: every wavefront alternates MFMA unit, VALU unit (lock step)
//   mode 4: as 3, wavefronts 4-7 start with the VALU unit (opposite phase)
//   mode 5: every wavefront: each MFMA followed by 3 independent VALU instructions (same instruction counts as mode 3)
// Prints ms, the shader clock (s_memtime cycles / s_memrealtime ticks) and cycles per unit pair.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_unit(f32x4 (&acc)[4], const half8 &a, const half8 &b) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
}
__device__ __forceinline__ void valu_unit(float (&v)[8], float m) {
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(m));
}
__device__ __forceinline__ void mixed_unit(f32x4 (&acc)[4], const half8 &a, const half8 &b, float (&v)[8], float m) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
            // 64 MFMAs x 4 = 256 VALU: the same counts as one MFMA unit + one VALU unit
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(4 * j) & 7]) : "v"(m));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(4 * j + 1) & 7]) : "v"(m));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(4 * j + 2) & 7]) : "v"(m));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(4 * j + 3) & 7]) : "v"(m));
        }
}

__global__ __launch_bounds__(512, 1) void probe(int mode, int units, float *sink, unsigned long long *clocks) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i)), b[i] = (_Float16)(0.002f * (threadIdx.x - i));
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.001f * (threadIdx.x + j);
    const float m = 0.999f;
    if (threadIdx.x == 0) lds[0] = 0;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const bool second = wave >= 4;
    for (int u = 0; u < units; ++u) {
        switch (mode) {
            case 0: mfma_unit(acc, a, b); mfma_unit(acc, a, b); break;
            case 1: valu_unit(v, m); valu_unit(v, m); break;
            case 2:
                if (second) { valu_unit(v, m); valu_unit(v, m); } else { mfma_unit(acc, a, b); mfma_unit(acc, a, b); }
                break;
            case 3: mfma_unit(acc, a, b); valu_unit(v, m); break;
            case 4:
                if (second) { valu_unit(v, m); mfma_unit(acc, a, b); } else { mfma_unit(acc, a, b); valu_unit(v, m); }
                break;
            default: mixed_unit(acc, a, b, v, m); break;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&clocks[0], c1 - c0);
        atomicAdd(&clocks[1], w1 - w0);
    }
}

int main(int argc, char **argv) {
    const int units = argc > 1 ? atoi(argv[1]) : 2000;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *sink;
    unsigned long long *clocks;
    (void)hipMalloc((void **)&sink, 4096);
    (void)hipMalloc((void **)&clocks, 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    static const char *const names[6] = {"all MFMA (2 units)", "all VALU (2 units)", "waves 0-3 MFMA x2, 4-7 VALU x2", "MFMA unit then VALU unit, lock step",
                                         "same, waves 4-7 in opposite phase", "each MFMA + 4 VALU interleaved"};
    for (int blocks : {cus, cus / 4}) {
        printf("%d workgroups of 8 wavefronts (%d compute units), %d trips\n", blocks, cus, units);
        for (int mode = 0; mode < 6; ++mode) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 100 * 1024, 0, mode, units / 10, sink, clocks);  // warm
            (void)hipMemset(clocks, 0, 16);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 100 * 1024, 0, mode, units, sink, clocks);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2];
            (void)hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
            const double waves = 8.0 * blocks;
            printf("  mode %d  %-40s %8.3f ms  %.3f GHz  %8.0f cycles per trip\n", mode, names[mode], ms, (double)h[0] / (double)h[1] * 0.1,
                   (double)h[0] / waves / units);
        }
    }
    return 0;
}
