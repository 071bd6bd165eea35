// Vector-instruction issue rate of one SIMD with 1..8 wavefronts resident, for the plain and the
// PACKED f32 forms (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 do two f32 operations per lane).
// The question behind it: is a packed op worth two plain ones in a VALU-bound kernel (k_raster on
// the 10 M-triangle frame issues a vector instruction on 95 % of its SIMD cycles)?
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// 16 independent instructions per trip
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int trips, float seed)
{
    float a[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float c = seed * 0.999f;
    const v2f c2 = {c, c * 1.001f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if constexpr (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
            if constexpr (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if constexpr (MODE == 5) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            if constexpr (MODE == 7) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
            if constexpr (MODE == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (MODE == 9) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(c2));
            if constexpr (MODE == 12) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 13) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 14) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 15) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 16) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 17) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            if constexpr (MODE == 18) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 11) asm volatile("v_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %3" : "+v"(a[i]), "+v"(p[i]) : "v"(c), "v"(c2));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char *name, float *out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int trips = 16384;
    for (int wps : {2, 4, 8}) {
        // 256 CUs, 4 SIMDs each; a 256-thread workgroup = one wavefront per SIMD
        const int blocks = 256 * wps;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, trips, 1.0f);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double per_simd = (double)trips * 16 * (MODE == 11 ? 2 : 1) * wps;    // wavefront instructions per SIMD
        printf("%-28s waves/SIMD %d: %8.3f ms  %6.1f wavefront instructions per us per SIMD  (%.2f cycles each at 2.4 GHz)\n",
               name, wps, best, per_simd / (best * 1e3), best * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4096));
    run<0>("v_mul_f32 (e32)", out);
    run<13>("v_mul_f32_e64", out);
    run<12>("v_add_f32_e32", out);
    run<15>("v_sub_f32_e32", out);
    run<14>("v_fmac_f32_e32", out);
    run<16>("v_add_u32_e32", out);
    run<17>("v_cndmask_b32_e32", out);
    run<18>("v_mad_u32_u24 (VOP3)", out);
    run<1>("v_pk_mul_f32", out);
    run<2>("v_fma_f32", out);
    run<3>("v_pk_fma_f32", out);
    run<4>("v_pk_add_f32", out);
    run<10>("v_pk_add_f32 neg", out);
    run<5>("v_min3_f32", out);
    run<6>("v_cmp_lt_f32", out);
    run<7>("v_cvt_f32_i32", out);
    run<8>("v_rcp_f32", out);
    run<9>("v_div_fixup_f32", out);
    run<11>("v_mul_f32 + v_pk_mul_f32", out);
    return 0;
}
