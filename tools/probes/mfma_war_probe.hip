// Probe: may a register that an in-flight v_mfma_f32_16x16x32_f16 reads as SrcC (or SrcA) be overwritten N wait states after the MFMA
// was issued -- by a VALU move, or by a ds_read whose data returns later -- when other MFMAs (of this wavefront: K queued ahead; of the
// other wavefront on the SIMD: `waves` per SIMD) keep the matrix pipe busy?  The compiler's hazard recogniser separates such a write
// from the MFMA by 3 wait states (ISA listing of mnv_guided_fused2.h); this measures whether that holds on gfx950.
// Every lane checks d = 0 * 0 + c == 1.0 after c's registers were overwritten with 2.0; counts lanes that saw anything else.
// build: hipcc -O2 --offload-arch=gfx950 mfma_war_probe.hip -o mfma_war_probe ; run: ./mfma_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N, int K, int WRITER, int OPERAND>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, int iters) {
    __shared__ float4 two[64];
    two[threadIdx.x & 63] = make_float4(2.f, 2.f, 2.f, 2.f);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(&two[threadIdx.x & 63]);  // LDS byte address (low 32 bits of the generic pointer are the offset)
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t"
            "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
            "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
            "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
            "s_nop 15\n\t"
            ".rept %c6\n\t"
            "v_mfma_f32_16x16x32_f16 v[56:59], v[48:51], v[52:55], v[60:63]\n\t"  // K independent MFMAs ahead of the probed one
            ".endr\n\t"
            ".if %c8 == 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], v[40:43]\n\t"  // d = 0 * 0 + c   (c = v[40:43])
            ".else\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[40:43], v[52:55], v[60:63]\n\t"  // d = a * 0 + 0   (a = v[40:43]; expect 0)
            ".endif\n\t"
            ".if %c5 > 0\n\t"
            "s_nop %c5 - 1\n\t"
            ".endif\n\t"
            ".if %c7 == 0\n\t"
            "v_mov_b32 v40, 2.0\n\tv_mov_b32 v41, 2.0\n\tv_mov_b32 v42, 2.0\n\tv_mov_b32 v43, 2.0\n\t"
            ".else\n\t"
            "ds_read_b128 v[40:43], %4\n\t"
            ".endif\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
            : "v"(addr), "n"(N), "n"(K), "n"(WRITER), "n"(OPERAND)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
              "v60", "v61", "v62", "v63", "memory");
        const float want = OPERAND == 0 ? 1.f : 0.f;
        wrong += (d0 != want) + (d1 != want) + (d2 != want) + (d3 != want);
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int N, int K, int WRITER, int OPERAND>
void run(int waves_per_simd) {
    unsigned long long *bad;
    hipMalloc(&bad, 8);
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL((probe<N, K, WRITER, OPERAND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, bad, 20000);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("operand %s writer %s  nops %2d  queued-ahead %2d  waves/SIMD %d : wrong lane-values %llu\n", OPERAND ? "SrcA" : "SrcC", WRITER ? "ds_read" : "v_mov  ", N, K,
           waves_per_simd, h);
    hipFree(bad);
}

template <int WRITER, int OPERAND>
void sweep() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0, WRITER, OPERAND>(w); run<1, 0, WRITER, OPERAND>(w); run<3, 0, WRITER, OPERAND>(w); run<7, 0, WRITER, OPERAND>(w);
        run<0, 1, WRITER, OPERAND>(w); run<3, 1, WRITER, OPERAND>(w); run<7, 1, WRITER, OPERAND>(w); run<15, 1, WRITER, OPERAND>(w);
        run<3, 4, WRITER, OPERAND>(w); run<7, 4, WRITER, OPERAND>(w); run<15, 4, WRITER, OPERAND>(w);
        run<3, 15, WRITER, OPERAND>(w); run<15, 15, WRITER, OPERAND>(w);
    }
}

int main() {
    sweep<0, 0>();
    sweep<1, 0>();
    sweep<0, 1>();
    sweep<1, 1>();
    return 0;
}
