// Calibration of rocprofv3 FETCH_SIZE for the march kernel's access pattern (MI355X_MICROARCH.md:
// "calibrate on a known byte count in your own access pattern"): N distinct, randomly placed, 64-B
// aligned rows of a 2 GiB array are read once each, 3 lanes per row (20 B per lane, like the
// (sample, channel) tasks).  Known bytes = N * 64.   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct __attribute__((packed, aligned(4))) W5 { uint32_t w[5]; };

__global__ void gather_rows(const uint8_t *rows, const uint32_t *idx, uint32_t n, uint32_t *sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t / 3, c = t % 3;
    if (s >= n) return;
    const W5 v = *reinterpret_cast<const W5 *>(rows + (uint64_t)idx[s] * 64 + c * 20);
    uint32_t a = v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3] ^ v.w[4];
    if (a == 0x12345678u) sink[0] = a;
}

// the same gather with non-temporal loads: does the L2 then fetch 64-B sectors instead of 128-B lines?
__global__ void gather_rows_nt(const uint8_t *rows, const uint32_t *idx, uint32_t n, uint32_t *sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t / 4, c = t % 4;
    if (s >= n) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(rows + (uint64_t)idx[s] * 64 + c * 16);
    uint32_t a = __builtin_nontemporal_load(&p->x) ^ __builtin_nontemporal_load(&p->y) ^ __builtin_nontemporal_load(&p->z) ^ __builtin_nontemporal_load(&p->w);
    if (a == 0x12345678u) sink[0] = a;
}

// reads 16 B of half `half` (0/1) of the 128-B line that holds row idx[s]
__global__ void touch_half(const uint8_t *rows, const uint32_t *idx, uint32_t n, uint32_t half, uint32_t *sink) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint4 v = *reinterpret_cast<const uint4 *>(rows + (uint64_t)(idx[s] >> 1) * 128 + half * 64);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}

__global__ void stream_read(const uint4 *p, uint64_t n, uint32_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) sink[0] = a;
}

int main() {
    const uint64_t n_rows = 1ull << 25;  // 2 GiB of 64-B rows
    const uint32_t n = 1u << 22;         // 4 Mi distinct rows -> 256 MiB known bytes
    uint8_t *rows; uint32_t *idx, *sink;
    hipMalloc(&rows, n_rows * 64); hipMemset(rows, 1, n_rows * 64);
    hipMalloc(&idx, n * 4); hipMalloc(&sink, 4);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    for (uint32_t i = 0; i < n; ++i) {  // distinct pseudo-random rows: stride walk with a random offset inside each stride
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (uint32_t)((uint64_t)i * (n_rows / n) + (x % (n_rows / n)));
    }
    for (uint32_t i = n - 1; i > 0; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint32_t j = x % (i + 1); std::swap(h[i], h[j]); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gather_rows, dim3((n * 3 + 255) / 256), dim3(256), 0, 0, rows, idx, n, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const uint4 *)rows, (uint64_t)(1ull << 30) / 16, sink);  // 1 GiB coalesced
    hipDeviceSynchronize();
    hipLaunchKernelGGL(gather_rows_nt, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, rows, idx, n, sink);
    hipDeviceSynchronize();
    // does a miss on one 64-B half fill the whole 128-B line?  touch half 0 of 8192 lines, then half 1
    hipLaunchKernelGGL(touch_half, dim3(32), dim3(256), 0, 0, rows, idx, 8192u, 0u, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(touch_half, dim3(32), dim3(256), 0, 0, rows, idx, 8192u, 1u, sink);
    hipDeviceSynchronize();
    printf("gather_rows known bytes: %llu (+ %u index bytes); stream_read known bytes: %llu\n", (unsigned long long)n * 64, n * 4, 1ull << 30);
    return 0;
}
