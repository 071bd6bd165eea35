// What does a SELECT cost on gfx950?  scripts/ubench/valu_rate.hip measured v_cndmask_b32_e32 at ~23 cycles
// per wavefront instruction (a v_mul_f32: 2.5) and nobody followed it up.  This one asks which part of that is
// the instruction and which the harness: the e32 form on vcc with and without the s_nop the compiler puts
// behind an asm statement that clobbers vcc, the e64 form on an SGPR pair, a select the COMPILER emits from
// C code, and the instructions one would replace a select with (v_bfi_b32, v_and_or_b32, v_max / v_min,
// a multiplication by 0 / 1).
// build: hipcc --offload-arch=gfx950 -O2 -o select_cost select_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int trips, float seed)
{
    float a[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; u[i] = threadIdx.x * 17u + i; }
    const float c = seed * 0.999f;
    const unsigned cu = (unsigned)seed * 3u + 1u;
    // a lane mask in an SGPR pair and in vcc, set ONCE
    unsigned long long mask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    unsigned long long mask2 = __builtin_amdgcn_ballot_w64((threadIdx.x & 2) != 0);
    if constexpr (MODE == 0 || MODE == 1 || MODE == 9 || MODE == 26) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
    if constexpr (MODE == 20) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[0]), "v"(c + 100.f) : "vcc");      // vcc from the VALU, once
    // (the loop's counter lives in SCC between its compare and its branch: nothing in here may write SCC —
    //  an s_and_b64 in the body made this loop endless)
    for (int t = 0; t < trips; ++t) {
        if constexpr (MODE == 21) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");                               // the SALU writes vcc every trip
        if constexpr (MODE == 22) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[0]), "v"(c + 100.f) : "vcc");  // the VALU writes vcc every trip
        if constexpr (MODE == 0) {
            // one asm block: sixteen v_cndmask on vcc back to back, no s_nop between them
            asm volatile(
                "v_cndmask_b32_e32 %0, %0, %16, vcc\n\tv_cndmask_b32_e32 %1, %1, %16, vcc\n\t"
                "v_cndmask_b32_e32 %2, %2, %16, vcc\n\tv_cndmask_b32_e32 %3, %3, %16, vcc\n\t"
                "v_cndmask_b32_e32 %4, %4, %16, vcc\n\tv_cndmask_b32_e32 %5, %5, %16, vcc\n\t"
                "v_cndmask_b32_e32 %6, %6, %16, vcc\n\tv_cndmask_b32_e32 %7, %7, %16, vcc\n\t"
                "v_cndmask_b32_e32 %8, %8, %16, vcc\n\tv_cndmask_b32_e32 %9, %9, %16, vcc\n\t"
                "v_cndmask_b32_e32 %10, %10, %16, vcc\n\tv_cndmask_b32_e32 %11, %11, %16, vcc\n\t"
                "v_cndmask_b32_e32 %12, %12, %16, vcc\n\tv_cndmask_b32_e32 %13, %13, %16, vcc\n\t"
                "v_cndmask_b32_e32 %14, %14, %16, vcc\n\tv_cndmask_b32_e32 %15, %15, %16, vcc"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                  "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                : "v"(c));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MODE == 1) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");   // valu_rate's form
            if constexpr (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(mask));
            if constexpr (MODE == 3) a[i] = (a[i] > c) ? a[i] * c : a[i] + c;          // what the compiler makes of a select
            if constexpr (MODE == 4) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(cu));
            if constexpr (MODE == 5) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(cu));
            if constexpr (MODE == 6) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 7) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
            if constexpr (MODE == 8) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 9) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(cu) : "vcc");
            if constexpr (MODE == 10) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(c));
            if constexpr (MODE == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if constexpr (MODE == 12) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(cu));
            if constexpr (MODE == 13) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2" : : "v"(a[i]), "s"(mask), "v"(c));
            if constexpr (MODE == 14) { asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(c) : "vcc"); }
            if constexpr (MODE == 20 || MODE == 21 || MODE == 22) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            if constexpr (MODE == 23) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c), "s"(mask) : "vcc");
            if constexpr (MODE == 24) asm volatile("s_mov_b64 vcc, %2\n\ts_nop 3\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c), "s"(mask) : "vcc");
            if constexpr (MODE == 25) asm volatile("s_mov_b64 %1, %3\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[i]), "=&s"(mask2) : "v"(c), "s"(mask));
            if constexpr (MODE == 26) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            // how far from the v_cmp that wrote vcc may a VOP2 v_cndmask sit?  (groups: one v_cmp, fillers, selects)
            if constexpr (MODE == 31) { if ((i & 1) == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, %0, %3, vcc\n\tv_cndmask_b32_e32 %1, %1, %3, vcc" : "+v"(a[i]), "+v"(a[i + 1]) : "v"(a[(i + 2) & 15]), "v"(c) : "vcc"); }
            if constexpr (MODE == 32) { if ((i & 3) == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %4, %5\n\tv_cndmask_b32_e32 %0, %0, %5, vcc\n\tv_cndmask_b32_e32 %1, %1, %5, vcc\n\tv_cndmask_b32_e32 %2, %2, %5, vcc\n\tv_cndmask_b32_e32 %3, %3, %5, vcc" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(a[(i + 4) & 15]), "v"(c) : "vcc"); }
            if constexpr (MODE == 33) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_mul_f32_e32 %1, %1, %3\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 5) & 15]) : "v"(a[(i + 1) & 15]), "v"(c) : "vcc");
            if constexpr (MODE == 34) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_mul_f32_e32 %1, %1, %3\n\tv_mul_f32_e32 %4, %4, %3\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 5) & 15]) , "+v"(a[(i + 9) & 15]): "v"(a[(i + 1) & 15]), "v"(c) : "vcc");
            if constexpr (MODE == 35) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_mul_f32_e32 %1, %1, %3\n\tv_mul_f32_e32 %4, %4, %3\n\tv_mul_f32_e32 %1, %1, %3\n\tv_mul_f32_e32 %4, %4, %3\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 5) & 15]) , "+v"(a[(i + 9) & 15]): "v"(a[(i + 1) & 15]), "v"(c) : "vcc");
            if constexpr (MODE == 36) asm volatile("v_cmp_lt_f32_e64 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(c) : "vcc");
            if constexpr (MODE == 37) { if ((i & 3) == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %4, %5\n\tv_cndmask_b32_e64 %0, %0, %5, vcc\n\tv_cndmask_b32_e64 %1, %1, %5, vcc\n\tv_cndmask_b32_e64 %2, %2, %5, vcc\n\tv_cndmask_b32_e64 %3, %3, %5, vcc" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(a[(i + 4) & 15]), "v"(c) : "vcc"); }
            if constexpr (MODE == 38) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %2, %0, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(c) : "vcc");   // the other operand order
            if constexpr (MODE == 39) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(c) : "vcc");    // an inline constant
            if constexpr (MODE == 15) u[i] = (u[i] < cu + i) ? u[i] + 3u : u[i] ^ cu;     // integer select from C
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char *name, float *out, int per_trip = 16)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int trips = 8192;
    for (int wps : {2, 8}) {
        const int blocks = 256 * wps;       // a 256-thread workgroup = one wavefront per SIMD
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, trips, 1.0f);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double per_simd = (double)trips * per_trip * wps;
        printf("%-52s waves/SIMD %d: %8.3f ms  %.2f cycles per counted instruction at 2.4 GHz\n",
               name, wps, best, best * 1e-3 * 2.4e9 / per_simd);
        fflush(stdout);
    }
    return 0;
}

// select_cost [mode ...]: only the named modes (one process per risky mode, under its own timeout)
static bool wanted(int argc, char **argv, int mode)
{
    if (argc < 2) return true;
    for (int i = 1; i < argc; ++i) if (atoi(argv[i]) == mode) return true;
    return false;
}
#define RUN(mode, ...) do { if (wanted(argc, argv, mode)) run<mode>(__VA_ARGS__); } while (0)

int main(int argc, char **argv)
{
    float *out; CK(hipMalloc(&out, 4096));
    RUN(8, "v_mul_f32_e32 (yardstick)", out);
    RUN(0, "v_cndmask_b32_e32 vcc, 16 in ONE asm block", out);
    RUN(1, "v_cndmask_b32_e32 vcc, one asm each (+ s_nop)", out);
    RUN(2, "v_cndmask_b32_e64 on an SGPR pair", out);
    RUN(26, "v_cndmask_b32_e64 naming vcc (SALU wrote it, once)", out);
    RUN(20, "v_cndmask_e32 vcc; a v_cmp wrote vcc ONCE before the loop", out);
    RUN(21, "v_cndmask_e32 vcc; s_mov_b64 vcc every 16", out);
    RUN(22, "v_cndmask_e32 vcc; v_cmp vcc every 16", out);
    RUN(23, "s_mov_b64 vcc + v_cndmask_e32 vcc pairs (count: 16)", out);
    RUN(24, "s_mov_b64 vcc + s_nop 3 + v_cndmask_e32 vcc (count: 16)", out);
    RUN(25, "s_mov_b64 sgpr pair + v_cndmask_e64 pairs (count: 16)", out);
    RUN(3, "C: a = a > c ? a * c : a + c  (count: 16 selects)", out);
    RUN(15, "C: u = u < k ? u + 3 : u ^ c  (count: 16 selects)", out);
    RUN(14, "v_cmp_lt_f32 vcc + v_cndmask vcc pairs (count: 32)", out, 32);
    RUN(31, "groups: v_cmp vcc + 2 v_cndmask_e32 (cycles per GROUP)", out, 8);
    RUN(32, "groups: v_cmp vcc + 4 v_cndmask_e32 (cycles per GROUP)", out, 4);
    RUN(37, "groups: v_cmp vcc + 4 v_cndmask_e64 naming vcc (per GROUP)", out, 4);
    RUN(33, "groups: v_cmp vcc, 1 v_mul, v_cndmask_e32 (per GROUP)", out);
    RUN(34, "groups: v_cmp vcc, 2 v_mul, v_cndmask_e32 (per GROUP)", out);
    RUN(35, "groups: v_cmp vcc, 4 v_mul, v_cndmask_e32 (per GROUP)", out);
    RUN(36, "groups: v_cmp_e64 naming vcc + v_cndmask_e32 (per GROUP)", out);
    RUN(38, "groups: v_cmp vcc + v_cndmask_e32, operands swapped (per GROUP)", out);
    RUN(39, "groups: v_cmp vcc + v_cndmask_e32 with constant 0 (per GROUP)", out);
    RUN(13, "v_cmp_lt_f32_e64 into an SGPR pair", out);
    RUN(9, "v_addc_co_u32 (vcc in and out)", out);
    RUN(4, "v_bfi_b32", out);
    RUN(5, "v_and_or_b32", out);
    RUN(6, "v_max_f32_e32", out);
    RUN(7, "v_min_u32_e32", out);
    RUN(12, "v_lshl_add_u32", out);
    RUN(10, "v_mov_b32_e32", out);
    RUN(11, "v_mov_b32_dpp row_shr:1", out);
    return 0;
}
