// Host cost of one pipelined frame through the C ABI alone (no Python): random small triangles
// shaped like the T-Rex workload (13 814 triangles, 1024 x 1024), swap chain of `depth`.
// build: hipcc -O2 -I include -o frame_issue scripts/ubench/frame_issue.cpp -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "crender_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "cython3dmodelrenderer_amd/libcrender_hip.so";
    const int depth = argc > 2 ? std::atoi(argv[2]) : 3;
    void *h = dlopen(path, RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
#define SYM(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(h, #name)); if (!p_##name) { printf("missing %s\n", #name); return 1; }
    SYM(crender_projection_matrix) SYM(crender_plan_workspace_bytes) SYM(crender_plan_create)
    SYM(crender_pipeline_create) SYM(crender_pipeline_frame) SYM(crender_pipeline_join) SYM(crender_last_error)
    const int H = 1024, W = 1024; const int64_t T = 13814;
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> tri(T * 9), col(T * 9, 200.f), nrm(T * 9);
    for (int64_t t = 0; t < T; ++t) {
        const float cz = 2.5f + 0.5f * u(rng), cx = 0.6f * u(rng), cy = 0.9f * u(rng);
        for (int v = 0; v < 3; ++v) {
            tri[t * 9 + v * 3 + 0] = cx + 0.02f * u(rng); tri[t * 9 + v * 3 + 1] = cy + 0.02f * u(rng);
            tri[t * 9 + v * 3 + 2] = cz + 0.02f * u(rng);
            nrm[t * 9 + v * 3 + 0] = 0; nrm[t * 9 + v * 3 + 1] = 0; nrm[t * 9 + v * 3 + 2] = (t & 1) ? 1.f : -1.f;
        }
    }
    float *d_tri, *d_col, *d_nrm;
    CK(hipMalloc(&d_tri, T * 36)); CK(hipMalloc(&d_col, T * 36)); CK(hipMalloc(&d_nrm, T * 36));
    CK(hipMemcpy(d_tri, tri.data(), T * 36, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), T * 36, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nrm, nrm.data(), T * 36, hipMemcpyHostToDevice));
    float P[16]; p_crender_projection_matrix(45, 0.1, 1000, H, W, P);
    crender_plan *plans[8]; float *z[8], *c[8], *n[8];
    const size_t ws = p_crender_plan_workspace_bytes(H, W, 0, H, T, 0, 0);
    for (int k = 0; k < depth; ++k) {
        void *w; CK(hipMalloc(&w, ws));
        if (p_crender_plan_create(&plans[k], H, W, 0, H, T, 0, 0, w, ws, nullptr)) { printf("%s\n", p_crender_last_error()); return 1; }
        CK(hipMalloc(&z[k], (size_t)H * W * 4)); CK(hipMalloc(&c[k], (size_t)H * W * 12)); CK(hipMalloc(&n[k], (size_t)H * W * 12));
    }
    crender_pipeline *pipe;
    if (p_crender_pipeline_create(&pipe, plans, depth)) { printf("%s\n", p_crender_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    auto frame = [&](int i) { const int k = i % depth; return p_crender_pipeline_frame(pipe, d_tri, d_col, d_nrm, T, P, z[k], c[k], n[k], nullptr, CRENDER_FUSED_CLEAR, nullptr); };
    for (int i = 0; i < 30 * depth; ++i) if (frame(i)) { printf("%s\n", p_crender_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    for (int K : {200, 3000}) {
        double t0 = now();
        for (int i = 0; i < K; ++i) frame(i);
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("depth %d, %5d frames: issue %.2f us/frame, total %.2f us/frame\n", depth, K, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K);
    }
    return 0;
}
