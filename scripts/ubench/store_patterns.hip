// Microbenchmark: how fast can one workgroup-per-tile kernel write the three framebuffer
// planes (z 4 B/px, colour 12 B/px, normal 12 B/px) of an HxW frame, by store pattern?
//   v0  thread per pixel: dword + 2 x dwordx3            (the r01-v1 resolve)
//   v1  thread per 4 px in a row: float4 + 3 + 3 float4  (registers only)
//   v2  wave stages a row in LDS, then contiguous float4 stores
//   v3  linear grid-stride float4 clear (reference ceiling, no tiling)
// build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TS>
__global__ __launch_bounds__(256) void v0(float* z, float* c, float* n, int W, int ntx, float val) {
  int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx;
  for (int p = threadIdx.x; p < TS * TS; p += 256) {
    int x = tx * TS + (p % TS), y = ty * TS + (p / TS);
    size_t pix = (size_t)y * W + x;
    z[pix] = val;
    c[pix * 3] = val; c[pix * 3 + 1] = val; c[pix * 3 + 2] = val;
    n[pix * 3] = val; n[pix * 3 + 1] = val; n[pix * 3 + 2] = val;
  }
}
template <int TS>
__global__ __launch_bounds__(256) void v1(float* z, float* c, float* n, int W, int ntx, float val) {
  int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx;
  constexpr int QPR = TS / 4;  // quads per row
  float4 v = make_float4(val, val, val, val);
  for (int q = threadIdx.x; q < TS * QPR; q += 256) {
    int x = tx * TS + (q % QPR) * 4, y = ty * TS + (q / QPR);
    size_t pix = (size_t)y * W + x;
    *reinterpret_cast<float4*>(z + pix) = v;
    float4* cp = reinterpret_cast<float4*>(c + pix * 3);
    cp[0] = v; cp[1] = v; cp[2] = v;
    float4* np = reinterpret_cast<float4*>(n + pix * 3);
    np[0] = v; np[1] = v; np[2] = v;
  }
}
// each wave owns rows; per row: lanes write their pixel's 7 floats to LDS, then the wave
// stores the row's z (TS/4 float4), colour and normal (3*TS/4 float4 each) contiguously
template <int TS>
__global__ __launch_bounds__(256) void v2(float* z, float* c, float* n, int W, int ntx, float val) {
  __shared__ __attribute__((aligned(16))) float stage[4][64 * 7];
  int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx;
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* s = stage[wave];
  constexpr int PXW = 64;              // pixels per wave pass
  constexpr int ROWS = PXW / TS > 0 ? PXW / TS : 1;   // rows per pass when TS < 64
  for (int p0 = wave * PXW; p0 < TS * TS; p0 += 4 * PXW) {
    int p = p0 + lane;
    // compute (here: constant) and stage: z at [0,64), colour at [64, 256), normal at [256, 448)
    s[lane] = val;
    s[64 + lane * 3] = val; s[64 + lane * 3 + 1] = val; s[64 + lane * 3 + 2] = val;
    s[256 + lane * 3] = val; s[256 + lane * 3 + 1] = val; s[256 + lane * 3 + 2] = val;
    __builtin_amdgcn_wave_barrier();
    // rows covered by this pass
    int row0 = p0 / TS;
    // z: PXW floats = PXW/4 float4, split over ROWS rows of TS/4 float4
    if (lane < PXW / 4) {
      int r = lane / (TS / 4), k = lane % (TS / 4);
      size_t pix = (size_t)(ty * TS + row0 + r) * W + tx * TS;
      reinterpret_cast<float4*>(z + pix)[k] = reinterpret_cast<float4*>(s)[lane];
    }
    if (lane < PXW * 3 / 4) {
      int r = lane / (TS * 3 / 4), k = lane % (TS * 3 / 4);
      size_t pix = (size_t)(ty * TS + row0 + r) * W + tx * TS;
      reinterpret_cast<float4*>(c + pix * 3)[k] = reinterpret_cast<float4*>(s + 64)[lane];
      reinterpret_cast<float4*>(n + pix * 3)[k] = reinterpret_cast<float4*>(s + 256)[lane];
    }
    __builtin_amdgcn_wave_barrier();
    (void)p; (void)ROWS;
  }
}
__global__ __launch_bounds__(256) void v3(float4* z, float4* c, float4* n, size_t npix4, float val) {
  float4 v = make_float4(val, val, val, val);
  size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix4; i += stride) z[i] = v;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix4 * 3; i += stride) { c[i] = v; n[i] = v; }
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  for (int H : {1024, 4096, 8192}) {
    int W = H; size_t np = (size_t)H * W;
    float *z, *c, *n;
    CK(hipMalloc(&z, np * 4)); CK(hipMalloc(&c, np * 12)); CK(hipMalloc(&n, np * 12));
    double bytes = np * 28.0;
    int reps = H == 1024 ? 200 : 20;
#define RUN(name, TS, K) { int ntx = W / TS, nt = ntx * (H / TS); \
    float ms = timeit([&] { hipLaunchKernelGGL((K<TS>), dim3(nt), dim3(256), 0, 0, z, c, n, W, ntx, 1.0f); }, reps); \
    printf("%4d^2 %-4s TS=%2d  %8.1f us  %7.1f GB/s\n", H, name, TS, ms * 1e3, bytes / ms / 1e6); }
    RUN("v0", 16, v0) RUN("v0", 32, v0) RUN("v0", 64, v0)
    RUN("v1", 16, v1) RUN("v1", 32, v1) RUN("v1", 64, v1)
    RUN("v2", 16, v2) RUN("v2", 32, v2) RUN("v2", 64, v2)
    for (int g : {1024, 4096, 16384}) {
      float ms = timeit([&] { hipLaunchKernelGGL(v3, dim3(g), dim3(256), 0, 0, (float4*)z, (float4*)c, (float4*)n, np / 4, 1.0f); }, reps);
      printf("%4d^2 v3 grid=%5d %8.1f us  %7.1f GB/s\n", H, g, ms * 1e3, bytes / ms / 1e6);
    }
    hipFree(z); hipFree(c); hipFree(n);
  }
  return 0;
}
