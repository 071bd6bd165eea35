// Microbenchmark: a synthetic frame with the launch shape of T-Rex 1024^2 on 16-pixel tiles — 4096
// workgroups of 256 threads, ~30 % of them "covered" (a dependent load, then a chain of dependent
// FMAs per wavefront and LDS traffic, then 7 KB of stores), the rest "empty" (a dependent load,
// then 7 KB of stores) — on 1..4 streams, to see what the machine does with that shape apart from
// the renderer's code.  Knobs: VALU instructions per covered wavefront, the dependent load on/off,
// covered tiles in one band of rows (like a model in the middle of the frame) or scattered,
// registers per thread, LDS per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 frame_shape.hip -o frame_shape
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS, int REGS>
__global__ __launch_bounds__(256) void frame(const unsigned* __restrict__ count, float* z, float* c, float* n,
                                             int W, int ntx, int valu_iters, int dep_load, float val) {
  __shared__ float pad[LDS / 4 + 64];
  const int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx;
  unsigned cnt = dep_load ? count[tile] : (unsigned)(blockIdx.x * 2654435761u >> 31);   // (dep_load = 0: no load)
  if (!dep_load) cnt = count[0] == 12345u ? 1u : ((ty >= 18 && ty < 46 && tx >= 12 && tx < 52) ? 1u : 0u);
  float acc[REGS];
#pragma unroll
  for (int r = 0; r < REGS; ++r) acc[r] = val + r;
  if (cnt) {
    // covered: dependent FMA chain + LDS round trips (a sweep)
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int r = 0; r < REGS; ++r) acc[r] = acc[r] * 1.0001f + 0.5f;
      if ((i & 31) == 31) { pad[threadIdx.x & 63] = acc[0]; __syncthreads(); acc[0] += pad[(threadIdx.x + 1) & 63]; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < REGS; ++r) s += acc[r];
  const float4 v = make_float4(s, s, s, s);
  const size_t p0 = (size_t)ty * 16 * W + (size_t)tx * 16;
  const int t = threadIdx.x;
  if (t < 64) *reinterpret_cast<float4*>(z + p0 + (size_t)(t >> 2) * W + (t & 3) * 4) = v;
  else { const int k = t - 64, r = k / 12; *reinterpret_cast<float4*>(c + (p0 + (size_t)r * W) * 3 + (k - r * 12) * 4) = v; }
  if (t < 192) { const int r = t / 12; *reinterpret_cast<float4*>(n + (p0 + (size_t)r * W) * 3 + (t - r * 12) * 4) = v; }
}

int main(int argc, char** argv) {
  const int W = 1024, H = 1024, ntx = 64, nt = 4096;
  std::vector<unsigned> h(nt, 0);
  int ncov = 0;
  for (int ty = 18; ty < 46; ++ty) for (int tx = 12; tx < 52; ++tx) { h[ty * ntx + tx] = 1; ++ncov; }   // a 40 x 28 block: 1120 covered tiles
  unsigned* dcount[4]; float *z[4], *c[4], *n[4]; hipStream_t st[4];
  for (int k = 0; k < 4; ++k) {
    CK(hipMalloc(&dcount[k], nt * 4)); CK(hipMemcpy(dcount[k], h.data(), nt * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&z[k], (size_t)W * H * 4)); CK(hipMalloc(&c[k], (size_t)W * H * 12)); CK(hipMalloc(&n[k], (size_t)W * H * 12));
    CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
  }
  printf("%d covered tiles of %d\n", ncov, nt);
  auto run = [&](auto kern, const char* name, int NS, int iters, int dep) {
    const int reps = 400;
    auto go = [&](int r) { for (int i = 0; i < r; ++i) { const int k = i % NS;
      hipLaunchKernelGGL(kern, dim3(nt), dim3(256), 0, st[k], dcount[k], z[k], c[k], n[k], W, ntx, iters, dep, 1.0f); } };
    go(40); (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    go(reps); (void)hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-22s streams %d  fma-iterations %4d (x REGS instr per wavefront)  dependent load %d: %6.2f us per frame\n", name, NS, iters, dep, us);
  };
  for (int NS : {1, 4}) {
    for (int dep : {1, 0}) {
      run(frame<13312, 8>, "lds 13K regs 8", NS, 0, dep);
      run(frame<13312, 8>, "lds 13K regs 8", NS, 75, dep);      // 600 VALU per covered wavefront
      run(frame<13312, 8>, "lds 13K regs 8", NS, 150, dep);     // 1200
      run(frame<13312, 8>, "lds 13K regs 8", NS, 300, dep);     // 2400
    }
    run(frame<13312, 32>, "lds 13K regs 32", NS, 19, 1);        // same 600 instructions, 4 independent chains x 8
    run(frame<13312, 32>, "lds 13K regs 32", NS, 75, 1);
    run(frame<0, 8>, "lds 0 regs 8", NS, 75, 1);
  }
  return 0;
}
