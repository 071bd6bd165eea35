// Host cost of enqueueing work on MI355X / ROCm: kernel launches vs a captured 2-kernel graph.
// build: hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { int a[8]; };
__global__ void k_small(const float *p0, const float *p1, const float *p2, const unsigned *p3, unsigned *p4,
                        const unsigned *p5, unsigned cap, float *z, float *c, float *n, int *w, Args g, int dbg)
{
    if (dbg == 12345 && threadIdx.x == 0) z[blockIdx.x] = p0[0] + p1[0] + p2[0] + p3[0] + p4[0] + p5[0] + cap + c[0] + n[0] + w[0] + g.a[3];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    float *buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    Args g{}; const int K = 20000;
    auto launch = [&](hipStream_t st, int grid) {
        hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st, buf, buf, buf, (unsigned *)buf, (unsigned *)buf,
                           (unsigned *)buf, 0u, buf, buf, buf, (int *)buf, g, 0);
    };
    for (int grid : {1, 4096}) {
        for (int ns : {1, 3}) {
            for (int i = 0; i < 200; ++i) launch(s[i % ns], grid);
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int i = 0; i < K; ++i) launch(s[i % ns], grid);
            double t1 = now();
            CK(hipDeviceSynchronize());
            double t2 = now();
            printf("launch grid=%4d streams=%d: issue %.2f us, total %.2f us per launch\n", grid, ns, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K);
        }
    }
    // graph of two kernels (54 + 4096 workgroups), one exec per stream
    hipGraphExec_t ge[3];
    for (int k = 0; k < 3; ++k) {
        hipGraph_t gr;
        CK(hipStreamBeginCapture(s[k], hipStreamCaptureModeThreadLocal));
        launch(s[k], 54); launch(s[k], 4096);
        CK(hipStreamEndCapture(s[k], &gr));
        CK(hipGraphInstantiate(&ge[k], gr, nullptr, nullptr, 0));
        CK(hipGraphDestroy(gr));
    }
    for (int ns : {1, 3}) {
        for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge[i % ns], s[i % ns]));
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < K; ++i) CK(hipGraphLaunch(ge[i % ns], s[i % ns]));
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("graph(2 kernels) streams=%d: issue %.2f us, total %.2f us per graph\n", ns, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K);
    }
    // the same two kernels as plain launches
    for (int ns : {1, 3}) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < K; ++i) { launch(s[i % ns], 54); launch(s[i % ns], 4096); }
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("2 launches streams=%d: issue %.2f us, total %.2f us per pair\n", ns, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K);
    }
    return 0;
}
