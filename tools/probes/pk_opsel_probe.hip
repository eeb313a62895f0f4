// Probe: does a VOP3P packed-FP32 add whose LOW half takes the HIGH dword of its second source
//     v_pk_add_f32 vD[0:1], vA[0:1], vB[0:1] op_sel:[0,1] op_sel_hi:[1,0]        (D.lo = A.lo + B.hi, D.hi = A.hi + B.lo)
// always read B.hi?  In guided_fused2_kernel (round 3's "rare wrong denominator", LAB_NOTEBOOK.md) exactly this instruction -- formed by
// the SLP vectoriser from sh_channel's scalar sums -- returned A.lo + 0.0 in lanes 48-63, sporadically, in a wavefront whose SIMD neighbours
// were running v_mfma_f32_16x16x32_f16; replacing it by v_add_f32, or by the same packed add WITHOUT op_sel on pre-swapped operands, removed
// every failure of the deterministic reproducer (tools/f2lab/).  This stand-alone kernel tries to show the same outside that kernel:
// a workgroup of 16 wavefronts (4 per SIMD), the first `n_mfma` of them in a loop of independent MFMAs, the others testing the packed
// add against two scalar adds, optionally at raised priority (the consumers ran at s_setprio 2) and with LDS loads in flight.
// build: hipcc -O2 --offload-arch=gfx950 pk_opsel_probe.hip -o pk_opsel_probe ; run: ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM, int PRIO, int LDS, int LATE>
__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, float *sink, int iters, int n_mfma) {
    __shared__ float pad[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) pad[i] = (float)i * 0.25f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < n_mfma) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) sink[0] = c0[0];
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(2);
    unsigned long long wrong = 0;
    float guard = 0.f;
    for (int it = 0; it < iters; ++it) {
        f32x2 A, B, D;
        A[0] = 1.0f + 0.001f * (float)((it + lane) & 1023);
        A[1] = 2.0f + 0.003f * (float)((it * 7 + lane) & 1023);
        B[0] = 0.5f + 0.002f * (float)((it * 3 + lane) & 1023);
        B[1] = -1.25f - 0.004f * (float)((it * 5 + lane) & 1023);
        if (LDS) {  // two LDS loads issued right before the packed add and still in flight when it reads its operands.  Their destinations
                    // are registers the compiler does not know about (it cannot see that the data arrive later): v126 / v127, clobbered here
                    // and at the wait below, far above what this kernel allocates
            const float *p = pad + ((it * 64 + lane) & 2047);
            asm volatile("ds_read_b32 v126, %0\n\tds_read_b32 v127, %0 offset:256\n\t" : : "v"((uint32_t)(uintptr_t)p) : "memory", "v126", "v127");
        }
        if (FORM == 0)      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(D) : "v"(A), "v"(B));
        else if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B));
        else                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(D) : "v"(A), "v"(B));
        if (LDS) {
            float l0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, v126, v127" : "=v"(l0) : : "memory", "v126", "v127");
            guard += l0;
        }
        // the expected values by plain VOP2 instructions (written out: the compiler would vectorise `A[0] + B[1]` into the very instruction under test)
        float want0, want1;
        if (FORM == 0) {
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want0) : "v"(A[0]), "v"(B[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want1) : "v"(A[1]), "v"(B[0]));
        } else if (FORM == 1) {
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want0) : "v"(A[0]), "v"(B[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want1) : "v"(A[1]), "v"(B[1]));
        } else {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want0) : "v"(A[0]), "v"(B[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want1) : "v"(A[1]), "v"(B[0]));
        }
        if (LATE) {  // which of the two is wrong?  A third evaluation, eight idle cycles later, judges the packed result and the early VOP2 result
            float late0;
            asm volatile("s_nop 7\n\tv_add_f32 %0, %1, %2" : "=v"(late0) : "v"(A[0]), "v"(B[1]));
            if (FORM == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(late0) : "v"(A[0]), "v"(B[1]));
            if (D[0] != late0) wrong += 1ull;           // low word: the packed instruction's low result differs from the late evaluation
            if (want0 != late0) wrong += 1ull << 32;    // high word: the VOP2 instruction right behind the packed one differs from it
        } else {
            if (D[0] != want0) wrong += 1ull;
            if (D[1] != want1) wrong += 1ull << 32;
        }
    }
    if (guard == 123.456f) sink[1] = guard;
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}

template <int FORM, int PRIO, int LDS, int LATE = 0>
void run(int n_mfma) {
    unsigned long long *bad;
    float *sink;
    hipMalloc(&bad, 32);
    hipMalloc(&sink, 8);
    hipMemset(bad, 0, 32);
    hipLaunchKernelGGL((probe<FORM, PRIO, LDS, LATE>), dim3(256), dim3(1024), 0, 0, bad, sink, 20000, n_mfma);
    unsigned long long h[4] = {0, 0, 0, 0};
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    const char *form = FORM == 0 ? "pk_add op_sel:[0,1]/[1,0]" : FORM == 1 ? "pk_add (no op_sel)      " : "pk_mul op_sel:[0,1]/[1,0]";
    printf(LATE ? "%s  prio %d  lds-in-flight %d  mfma wavefronts %2d of 16 : against a LATE evaluation, per lane quarter: packed low result wrong %llu %llu %llu %llu, the VOP2 add right behind it wrong %llu %llu %llu %llu\n"
                : "%s  prio %d  lds-in-flight %d  mfma wavefronts %2d of 16 : wrong low results per lane quarter %llu %llu %llu %llu, wrong high results %llu %llu %llu %llu\n", form, PRIO ? 2 : 0, LDS,
           n_mfma, h[0] & 0xffffffffull, h[1] & 0xffffffffull, h[2] & 0xffffffffull, h[3] & 0xffffffffull, h[0] >> 32, h[1] >> 32, h[2] >> 32, h[3] >> 32);
    hipFree(bad);
    hipFree(sink);
}

template <int FORM>
void sweep() {
    for (int n : {0, 4, 8, 12}) {
        run<FORM, 0, 0>(n);
        run<FORM, 2, 0>(n);
        run<FORM, 0, 1>(n);
        run<FORM, 2, 1>(n);
    }
}

int main() {
    sweep<0>();
    sweep<1>();
    sweep<2>();
    for (int n : {0, 4, 8, 12}) {
        run<0, 0, 0, 1>(n);
        run<2, 0, 0, 1>(n);
    }
    return 0;
}
