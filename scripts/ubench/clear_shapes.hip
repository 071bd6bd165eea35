// Microbenchmark: what bounds a fused clear of a 1024 x 1024 frame (28 MB: z 4 B/px + colour and
// normal 12 B/px each) done tile by tile — bytes, partial lines, or the number of workgroups?
//   rect<BW,BH,NT>: one workgroup of NT threads clears a BW x BH pixel rectangle with float4 stores.
// Every variant writes exactly the same bytes.  Optional LDS reservation per workgroup mimics the
// raster kernel's footprint (15.4 KB => 8 workgroups per CU at most).
// build: hipcc --offload-arch=gfx950 -O3 clear_shapes.hip -o clear_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BW, int BH, int NT, int LDS>
__global__ __launch_bounds__(NT) void rect(float* z, float* c, float* n, int W, int nbx, float val) {
  __shared__ float pad[LDS / 4 + 1];
  if (LDS && val == 123.0f) pad[threadIdx.x] = val;     // keeps the allocation
  const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
  const size_t p0 = (size_t)by * BH * W + (size_t)bx * BW;
  const float4 v = make_float4(val, val, val, val);
  constexpr int ZQ = BW / 4, CQ = 3 * BW / 4;
  for (int i = threadIdx.x; i < BH * ZQ; i += NT) {
    const int r = i / ZQ;
    *reinterpret_cast<float4*>(z + p0 + (size_t)r * W + (i - r * ZQ) * 4) = v;
  }
  for (int i = threadIdx.x; i < BH * CQ; i += NT) {
    const int r = i / CQ;
    const size_t off = (p0 + (size_t)r * W) * 3 + (i - r * CQ) * 4;
    *reinterpret_cast<float4*>(c + off) = v;
    *reinterpret_cast<float4*>(n + off) = v;
  }
  if (LDS && val == 123.0f) z[0] = pad[(threadIdx.x + 1) % NT];
}

template <typename F> float timeit(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / reps;
}

int main() {
  const int W = 1024, H = 1024;
  float *z, *c, *n;
  CK(hipMalloc(&z, (size_t)W * H * 4)); CK(hipMalloc(&c, (size_t)W * H * 12)); CK(hipMalloc(&n, (size_t)W * H * 12));
  const double bytes = 28.0 * W * H;
#define RUN(BW, BH, NT, LDS) { \
    const int nbx = W / BW, nb = nbx * (H / BH); \
    float us = timeit([&] { hipLaunchKernelGGL((rect<BW, BH, NT, LDS>), dim3(nb), dim3(NT), 0, 0, z, c, n, W, nbx, 1.0f); }, 200); \
    printf("rect %3d x %-3d %4d threads  lds %5d  %5d workgroups: %6.2f us back to back = %5.2f TB/s\n", BW, BH, NT, LDS, nb, us, bytes / us / 1e6); }
  RUN(16, 16, 256, 0)  RUN(16, 16, 256, 15360)  RUN(16, 16, 64, 0)  RUN(16, 16, 128, 0)
  RUN(32, 16, 256, 0)  RUN(32, 16, 256, 15360)  RUN(32, 8, 256, 0)
  RUN(32, 32, 256, 0)  RUN(32, 32, 256, 15360)  RUN(64, 16, 256, 0)  RUN(64, 32, 256, 0)  RUN(64, 64, 256, 0)
  RUN(128, 16, 256, 0) RUN(1024, 1, 256, 0) RUN(1024, 4, 256, 0)  RUN(1024, 4, 1024, 0)
  // the same clear into NS separate frames on NS streams, launches round robin: frames in flight
  for (int NS : {1, 2, 4}) {
    hipStream_t st[4]; float *zz[4], *cc[4], *nn[4];
    for (int k = 0; k < NS; ++k) {
      CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
      CK(hipMalloc(&zz[k], (size_t)W * H * 4)); CK(hipMalloc(&cc[k], (size_t)W * H * 12)); CK(hipMalloc(&nn[k], (size_t)W * H * 12));
    }
    const int reps = 400;
    auto go = [&](int r) { for (int i = 0; i < r; ++i) { const int k = i % NS;
      hipLaunchKernelGGL((rect<16, 16, 256, 15360>), dim3(4096), dim3(256), 0, st[k], zz[k], cc[k], nn[k], W, 64, 1.0f); } };
    go(20); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto t0 = std::chrono::steady_clock::now();
    go(reps); CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%d stream(s), 16 x 16 tiles, 15 KB lds: %6.2f us per 28 MB frame = %5.2f TB/s (host clock, %d launches)\n", NS, us, bytes / us / 1e6, reps);
  }
  return 0;
}
