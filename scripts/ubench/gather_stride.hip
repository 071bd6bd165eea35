// Random gather of 36-byte records out of a large table: does the record stride / alignment
// change the cost?  (stride 9 floats = packed, 12 = 48 B, 16 = one aligned 64-B sector, 32 = 128 B)
// build: hipcc --offload-arch=gfx950 -O2 -o gather_stride gather_stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_gather(const float *__restrict__ table, const unsigned *__restrict__ idx, float *__restrict__ out,
                         size_t n, int stride)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = table + (size_t)idx[i] * stride;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) s += r[k];
    out[i] = s;
}
int main()
{
    const size_t T = 10'000'000, N = 12'000'000;
    std::vector<unsigned> h(N);
    std::mt19937 rng(7);
    for (auto &v : h) v = rng() % T;
    unsigned *idx; float *table, *out;
    CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&out, N * 4)); CK(hipMalloc(&table, T * 32 * 4));
    CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(table, 0, T * 32 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int stride : {9, 12, 16, 32}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            for (int it = 0; it < 5; ++it)
                hipLaunchKernelGGL(k_gather, dim3((N + 255) / 256), dim3(256), 0, 0, table, idx, out, N, stride);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("stride %2d floats: %.3f ms per 12M gathers (%.1f ns per 1000, %.0f useful GB/s)\n", stride, ms / 5,
                            ms / 5 * 1e6 / N * 1000, N * 36.0 / (ms / 5 * 1e-3) / 1e9);
        }
    }
    return 0;
}
