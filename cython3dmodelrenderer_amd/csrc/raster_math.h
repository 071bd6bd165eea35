// raster_math.h — per-vertex / per-sample arithmetic of the rasterizer (device side).
//
// Every function states the reference lines whose float32 operation order it keeps
// (".pyx" = crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx,
//  "mu.pyx" = crender/cy/pixel_buffer_filler/math_utils.pyx of the reference).
// The translation unit MUST be built with -ffp-contract=off, correctly rounded f32
// division and f32 denormals enabled: each operator below is one IEEE binary32
// operation, exactly as in the reference's gcc -O2 x86-64 build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CR_DEV __device__ __forceinline__

namespace crender {

// Projection constants handed to kernels by value (a3, .pyx:83-90 and .pyx:109).
struct ProjConst {
    float p[16];   // row-major 4x4
    float xs, ys;  // (float)(w / 2.0), (float)(h / 2.0)
};

// K1 body for one vertex, .pyx:116-130.  The reference runs in place, so column j
// is formed from the already overwritten columns < j; v[] is updated the same way.
CR_DEV void project_vertex(const ProjConst &P, float v[3])
{
    const float z = v[2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        v[j] = v[0] * P.p[0 * 4 + j] + v[1] * P.p[1 * 4 + j] + v[2] * P.p[2 * 4 + j] + P.p[3 * 4 + j];
    v[0] = v[0] / z;
    v[1] = v[1] / z;
    v[2] = v[2] / z;
    v[0] = v[0] + 1.0f;
    v[1] = v[1] + 1.0f;
    v[0] = v[0] * P.xs;
    v[1] = v[1] * P.ys;
}

// .pyx:202: (n0z + n1z + n2z) / 3 >= 0.0 with a float sum and a DOUBLE division by
// 3.0 (what Cython emits).  A double quotient of a float by 3 is zero only if the
// float is, and keeps its sign, so the test equals `sum >= 0` (true for -0, false for
// NaN) and the division is not needed.
CR_DEV bool backface(float n0z, float n1z, float n2z)
{
    const float s = n0z + n1z + n2z;
    return s >= 0.0f;
}

// <int>ceil(x) of the reference's x86-64 build (.pyx:165-166): the double ceil of a
// float is the float ceil; cvttsd2si returns INT_MIN for anything outside int range.
CR_DEV int ceil_to_int(float x)
{
    const float c = ceilf(x);
    if (!(c >= -2147483648.0f && c < 2147483648.0f)) return (int)0x80000000;
    return (int)c;
}

CR_DEV int clipi(int a, int lo, int hi)  // mu.pxd:8-13
{
    return a < lo ? lo : (a > hi ? hi : a);
}

// a6, .pyx:132-175: pixel box [xl, xr) x [yt, yb) of a projected triangle.
CR_DEV void pixel_box(float x0, float y0, float x1, float y1, float x2, float y2,
                      int w, int h, int &xl, int &xr, int &yt, int &yb)
{
    float fxl = (float)w, fxr = 0.0f, fyt = (float)h, fyb = 0.0f;
    if (x0 < fxl) fxl = x0;
    if (x0 > fxr) fxr = x0;
    if (y0 < fyt) fyt = y0;
    if (y0 > fyb) fyb = y0;
    if (x1 < fxl) fxl = x1;
    if (x1 > fxr) fxr = x1;
    if (y1 < fyt) fyt = y1;
    if (y1 > fyb) fyb = y1;
    if (x2 < fxl) fxl = x2;
    if (x2 > fxr) fxr = x2;
    if (y2 < fyt) fyt = y2;
    if (y2 > fyb) fyb = y2;
    xl = clipi(ceil_to_int(fxl), 0, w);
    xr = clipi(ceil_to_int(fxr), 0, w);
    yt = clipi(ceil_to_int(fyt), 0, h);
    yb = clipi(ceil_to_int(fyb), 0, h);
}

// The six xy coordinates + three depths of a projected triangle.
struct TriXYZ {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

// a9, mu.pyx:8-34: barycentrics of integer pixel (X, Y).  The nine edge constants are
// functions of the triangle only; they are written inline so the compiler hoists them
// out of sample loops.  Three correctly rounded divisions, no reciprocal.
CR_DEV void barycentric(const TriXYZ &t, int X, int Y, float &b1, float &b2, float &b3)
{
    const float l01 = t.x1 - t.x2, l02 = t.y1 - t.y2;
    const float l03 = l01 * (t.y0 - t.y2) - l02 * (t.x0 - t.x2);
    const float l11 = t.x2 - t.x0, l12 = t.y2 - t.y0;
    const float l13 = l11 * (t.y1 - t.y0) - l12 * (t.x1 - t.x0);
    const float l21 = t.x0 - t.x1, l22 = t.y0 - t.y1;
    const float l23 = l21 * (t.y2 - t.y1) - l22 * (t.x2 - t.x1);
    const float fx = (float)X, fy = (float)Y;
    b1 = (l01 * (fy - t.y2) - l02 * (fx - t.x2)) / l03;
    b2 = (l11 * (fy - t.y0) - l12 * (fx - t.x0)) / l13;
    b3 = (l21 * (fy - t.y1) - l22 * (fx - t.x1)) / l23;
}

// .pyx:219 / 226-231: attribute interpolation, a0*b1 + a1*b2 + a2*b3 left to right.
CR_DEV float interp(float a0, float a1, float a2, float b1, float b2, float b3)
{
    return a0 * b1 + a1 * b2 + a2 * b3;
}

// ---- depth keys ------------------------------------------------------------------
// The reference's serial loop leaves, per pixel, the minimum-z fragment, ties to the
// highest triangle index, a fragment equal to the prior buffer value overwriting it
// (write on `not new_z > z`, .pyx:223).  That is the minimum of the 64-bit keys
//     (order-preserving map of z) << 32 | (0xFFFFFFFE - triangle)   for fragments
//     (order-preserving map of z) << 32 |  0xFFFFFFFF               for the prior value
// so the whole z-order reduces to an order-independent unsigned 64-bit atomic min.
constexpr uint32_t KEY_LOW_PRIOR = 0xFFFFFFFFu;

CR_DEV uint32_t zord(float z)  // z is not NaN; -0 and +0 compare equal in the reference
{
    if (z == 0.0f) z = 0.0f;
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

CR_DEV uint32_t zord_prior(float z)  // prior buffer content: a NaN loses to everything
{
    return (z != z) ? 0xFFFFFFFFu : zord(z);
}

CR_DEV unsigned long long make_key(uint32_t zo, uint32_t low)
{
    return ((unsigned long long)zo << 32) | low;
}

CR_DEV unsigned long long fragment_key(float z, uint32_t tri)
{
    return make_key(zord(z), 0xFFFFFFFEu - tri);
}

// Coverage test of one sample: returns true and the key if pixel (X, Y) yields a
// fragment (.pyx:215-221: all barycentrics non-negative — NaN passes — and z not NaN).
CR_DEV bool fragment(const TriXYZ &t, uint32_t tri, int X, int Y, unsigned long long &key)
{
    float b1, b2, b3;
    barycentric(t, X, Y, b1, b2, b3);
    if (b1 < 0.0f || b2 < 0.0f || b3 < 0.0f) return false;
    const float z = interp(t.z0, t.z1, t.z2, b1, b2, b3);
    if (z != z) return false;  // == not (-1 <= z or z <= 1), .pyx:220
    key = fragment_key(z, tri);
    return true;
}

// ---- per-triangle setup for sample loops ------------------------------------------------
// The sweeps evaluate many pixels of one triangle, so everything that depends on the
// triangle only is computed once: the nine edge constants of mu.pyx:11-21, and two exact
// shortcuts for the three divisions of mu.pyx:34.
//
// (1) Sign rejection.  b_k = num_k / den_k is certainly < 0 when num_k and den_k are finite
//     with opposite signs, |num_k| >= 2^-60 and |den_k| <= 2^60 (the quotient is then at
//     least 2^-120 in magnitude: it cannot round to -0).  rej_k below is +-1 with den_k's
//     sign when den_k qualifies, else 0 (never reject); the test is num_k * rej_k < -2^-60.
//     Samples where the test fails on all edges go through the division as before.
//
// (2) Division by a per-triangle denominator.  hipcc expands a correctly rounded f32 `/` to
//       d' = div_scale(d), n' = div_scale(n), r = rcp(d'), r += fma(-d', r, 1) * r,
//       q = n' * r, q += fma(-d', q, n') * r, res = div_fmas(fma(-d', q, n'), r, q),
//       div_fixup(res, d, n).
//     When |d| and |n| both lie in [2^-40, 2^40] neither div_scale changes its operand and
//     div_fmas is a plain fma, so the reciprocal refinement depends on d alone and is hoisted;
//     the per-sample tail (1 mul, 4 fma; div_fixup is an identity inside the window) performs the
//     very same operations and rounds identically.  Outside the window (and for n == 0) the full `/` is used.
//     tests/test_hip_parity_gpu.py::test_fast_division_is_bit_exact checks the tail against
//     `/` on 2^27 operand pairs spanning the whole window.
constexpr float kRejTiny = 8.67361738e-19f;     // 2^-60
constexpr float kRejHuge = 1.15292150e+18f;     // 2^60
constexpr float kDivLo = 9.09494702e-13f;       // 2^-40
constexpr float kDivHi = 1.09951163e+12f;       // 2^40

CR_DEV bool in_div_window(float a)
{
    const float m = fabsf(a);
    return m >= kDivLo && m <= kDivHi;   // false for NaN, inf, 0
}

CR_DEV float refined_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// n / d for n, d inside the window, r = refined_rcp(d): the tail of hipcc's own expansion.
CR_DEV float div_tail(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    // (hipcc's expansion ends with v_div_fixup_f32(res, d, n): for finite non-zero operands whose
    // quotient neither overflows nor underflows — every pair inside the window — it returns res
    // with the sign of n / d, which res has already; left out, the exhaustive comparison with `/`
    // of test_fast_division_is_bit_exact still holds bit for bit)
    return __builtin_fmaf(e, r, q);
}

struct TriSetup {
    float x0, y0, x1, y1, x2, y2, z0, z1, z2;
    float l01, l02, l03, l11, l12, l13, l21, l22, l23;
    float r1, r2, r3;        // refined reciprocals of l03, l13, l23 (valid if fast)
    float rej1, rej2, rej3;  // +-1 / 0, see (1)
    bool fast;               // all three denominators inside the division window
};

CR_DEV float rej_sign(float den)
{
    const float m = fabsf(den);
    if (!(m > 0.0f && m <= kRejHuge)) return 0.0f;   // 0, inf, NaN, too large: never reject
    return den > 0.0f ? 1.0f : -1.0f;
}

// want_fast: prepare shortcut (2); not worth its ~20 instructions for a record that is
// swept for only a few blocks.
CR_DEV TriSetup make_setup(const TriXYZ &t, bool want_fast)
{
    TriSetup s;
    s.x0 = t.x0; s.y0 = t.y0; s.x1 = t.x1; s.y1 = t.y1; s.x2 = t.x2; s.y2 = t.y2;
    s.z0 = t.z0; s.z1 = t.z1; s.z2 = t.z2;
    s.l01 = t.x1 - t.x2; s.l02 = t.y1 - t.y2;
    s.l03 = s.l01 * (t.y0 - t.y2) - s.l02 * (t.x0 - t.x2);
    s.l11 = t.x2 - t.x0; s.l12 = t.y2 - t.y0;
    s.l13 = s.l11 * (t.y1 - t.y0) - s.l12 * (t.x1 - t.x0);
    s.l21 = t.x0 - t.x1; s.l22 = t.y0 - t.y1;
    s.l23 = s.l21 * (t.y2 - t.y1) - s.l22 * (t.x2 - t.x1);
    s.fast = want_fast && in_div_window(s.l03) && in_div_window(s.l13) && in_div_window(s.l23);
    s.r1 = s.r2 = s.r3 = 0.0f;
    if (s.fast) {
        s.r1 = refined_rcp(s.l03); s.r2 = refined_rcp(s.l13); s.r3 = refined_rcp(s.l23);
    }
    s.rej1 = rej_sign(s.l03); s.rej2 = rej_sign(s.l13); s.rej3 = rej_sign(s.l23);
    return s;
}

// Numerators of the three barycentrics at pixel (X, Y), mu.pyx:34 before the division.
CR_DEV void numerators(const TriSetup &s, int X, int Y, float &n1, float &n2, float &n3)
{
    const float fx = (float)X, fy = (float)Y;
    n1 = s.l01 * (fy - s.y2) - s.l02 * (fx - s.x2);
    n2 = s.l11 * (fy - s.y0) - s.l12 * (fx - s.x0);
    n3 = s.l21 * (fy - s.y1) - s.l22 * (fx - s.x1);
}

// True if some barycentric is certainly negative (shortcut (1)); false decides nothing.
CR_DEV bool surely_outside(const TriSetup &s, float n1, float n2, float n3)
{
    return (n1 * s.rej1 < -kRejTiny) || (n2 * s.rej2 < -kRejTiny) || (n3 * s.rej3 < -kRejTiny);
}

// The three quotients; `allow_fast` lets the caller switch shortcut (2) off (debug knob).
CR_DEV void quotients(const TriSetup &s, float n1, float n2, float n3, bool allow_fast,
                      float &b1, float &b2, float &b3)
{
    const float hi = fmaxf(fmaxf(fabsf(n1), fabsf(n2)), fabsf(n3));
    const float lo = fminf(fminf(fabsf(n1), fabsf(n2)), fabsf(n3));
    if (allow_fast && s.fast && lo >= kDivLo && hi <= kDivHi) {
        b1 = div_tail(n1, s.l03, s.r1);
        b2 = div_tail(n2, s.l13, s.r2);
        b3 = div_tail(n3, s.l23, s.r3);
    } else {
        b1 = n1 / s.l03;
        b2 = n2 / s.l13;
        b3 = n3 / s.l23;
    }
}

// Fragment test given the numerators (.pyx:215-221).
CR_DEV bool fragment_from(const TriSetup &s, uint32_t tri, float n1, float n2, float n3,
                          bool allow_fast, unsigned long long &key)
{
    float b1, b2, b3;
    quotients(s, n1, n2, n3, allow_fast, b1, b2, b3);
    if (b1 < 0.0f || b2 < 0.0f || b3 < 0.0f) return false;
    const float z = interp(s.z0, s.z1, s.z2, b1, b2, b3);
    if (z != z) return false;
    key = fragment_key(z, tri);
    return true;
}

// Load nine consecutive floats (one triangle of a [T][3][3] array).
CR_DEV void load9(const float *__restrict__ p, float v[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = p[i];
}

// (experiment, development builds: the same through non-temporal loads — the lines are used once)
CR_DEV void load9_nt(const float *__restrict__ p, float v[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = __builtin_nontemporal_load(p + i);
}

CR_DEV TriXYZ load_tri(const float *__restrict__ p)
{
    float v[9];
    load9(p, v);
    return TriXYZ{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
}

// f1 fused into the store (CRENDER_FUSED_GURO): guro_illumination.py:20-27 on one pixel, in numpy's
// float32 operation order — s = ((0 + n0*l0) + n1*l1) + n2*l2, m = sqrt((n0*n0 + n1*n1) + n2*n2),
// f = clip(s / (m + 1e-6), 0, 1) (a NaN stays a NaN), colour *= f.  The normal is stored as it is.
struct Light {
    float l0, l1, l2;
    int on;
};
CR_DEV float guro_factor(const Light &L, float n0, float n1, float n2)
{
    // (numpy's add.reduce over a short axis starts from +0: the three -0 products of a zero normal
    // under a light along -z sum to +0, not -0 — the sign of the background's shaded colour)
    const float s = ((0.0f + n0 * L.l0) + n1 * L.l1) + n2 * L.l2;
    const float m = sqrtf((n0 * n0 + n1 * n1) + n2 * n2);
    float f = s / (m + 1e-6f);
    f = f < 0.0f ? 0.0f : f;        // (np.clip keeps a NaN, and a -0)
    f = f > 1.0f ? 1.0f : f;
    return f;
}

// Element `idx` of an array whose base is the same for the whole wavefront.  With a 32-bit index
// type the address is base + a 32-BIT BYTE OFFSET (the caller guarantees idx * 4 < 2^32): the
// compiler then keeps the base in scalar registers and the per-lane part in one vector register,
// instead of 64-bit vector arithmetic per access (v_mad_u64_u32, v_lshl_add_u64).  Worth 1 % of
// bunny 4096^2's vector instructions and of its raster launch, where the vector pipes are the bound.
template <typename I>
CR_DEV float *elem(float *base, I idx)
{
    if constexpr (sizeof(I) == 4) return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (uint32_t)(idx * 4u));
    else return base + idx;
}
template <typename I>
CR_DEV const float *elem(const float *base, I idx)
{
    if constexpr (sizeof(I) == 4) return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint32_t)(idx * 4u));
    else return base + idx;
}

// z, colour and normal of a fragment from its barycentrics and the triangle's attributes
// (.pyx:219, 226-242), with the optional fused illumination.
template <typename I>
CR_DEV void store_fragment(float z, const float c[9], const float n[9], float b1, float b2, float b3,
                           const Light &L, I pix, float *__restrict__ zb, float *__restrict__ cb,
                           float *__restrict__ nb, bool nt = false)
{
    if (nt) {       // (experiment, development builds: non-temporal stores)
        float c0 = interp(c[0], c[3], c[6], b1, b2, b3);
        float c1 = interp(c[1], c[4], c[7], b1, b2, b3);
        float c2 = interp(c[2], c[5], c[8], b1, b2, b3);
        const float n0 = interp(n[0], n[3], n[6], b1, b2, b3);
        const float n1 = interp(n[1], n[4], n[7], b1, b2, b3);
        const float n2 = interp(n[2], n[5], n[8], b1, b2, b3);
        if (L.on) {
            const float f = guro_factor(L, n0, n1, n2);
            c0 *= f; c1 *= f; c2 *= f;
        }
        float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
        __builtin_nontemporal_store(z, elem(zb, pix));
        __builtin_nontemporal_store(c0, cp); __builtin_nontemporal_store(c1, cp + 1); __builtin_nontemporal_store(c2, cp + 2);
        __builtin_nontemporal_store(n0, np_); __builtin_nontemporal_store(n1, np_ + 1); __builtin_nontemporal_store(n2, np_ + 2);
        return;
    }
    *elem(zb, pix) = z;
    float c0 = interp(c[0], c[3], c[6], b1, b2, b3);
    float c1 = interp(c[1], c[4], c[7], b1, b2, b3);
    float c2 = interp(c[2], c[5], c[8], b1, b2, b3);
    const float n0 = interp(n[0], n[3], n[6], b1, b2, b3);
    const float n1 = interp(n[1], n[4], n[7], b1, b2, b3);
    const float n2 = interp(n[2], n[5], n[8], b1, b2, b3);
    if (L.on) {
        const float f = guro_factor(L, n0, n1, n2);
        c0 *= f; c1 *= f; c2 *= f;
    }
    float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
    cp[0] = c0; cp[1] = c1; cp[2] = c2;
    np_[0] = n0; np_[1] = n1; np_[2] = n2;
}

// Recompute the winning fragment of pixel (X, Y) and store z, colour, normal
// (.pyx:219, 226-242).  Same device functions as the coverage pass, so z is the very
// value the key was built from.
template <typename I>
CR_DEV void shade_and_store(const float *__restrict__ proj, const float *__restrict__ col,
                            const float *__restrict__ nrm, uint32_t tri, int X, int Y,
                            I pix, float *__restrict__ zb, float *__restrict__ cb,
                            float *__restrict__ nb, const Light &L = Light{0.f, 0.f, 0.f, 0}, int nt = 0)
{
    const TriXYZ t = load_tri(elem(proj, (I)((I)tri * 9)));
    float c[9], n[9];
    if (nt & 1) {
        load9_nt(elem(col, (I)((I)tri * 9)), c);
        load9_nt(elem(nrm, (I)((I)tri * 9)), n);
    } else {
        load9(elem(col, (I)((I)tri * 9)), c);
        load9(elem(nrm, (I)((I)tri * 9)), n);
    }
    float b1, b2, b3;
    barycentric(t, X, Y, b1, b2, b3);
    store_fragment(interp(t.z0, t.z1, t.z2, b1, b2, b3), c, n, b1, b2, b3, L, pix, zb, cb, nb, (nt & 2) != 0);
}

}  // namespace crender
