// model_ops.hip — the rows either side of the hot path on gfx950: Guro
// illumination (f1), the device-resident Model's transforms and normals (f2), presentation as a
// uint8 image (f3), texture-to-vertex colours (f4), and the sort keys of the tile-coherent
// triangle order.  Each entry point is argument checks + one or two launches.
#include "common.h"

using namespace crender_detail;

namespace {

// ---- f1: Guro illumination, guro_illumination.py:20-27 ----------------------------
// numpy evaluates, in float32: s = sum_k(n_k * l_k); m = sqrt(sum_k(n_k * n_k));
// c = clip(s / (m + 1e-6f), 0, 1); colour *= c.  A 3-element float32 add.reduce over the
// last axis runs left to right, (a0 + a1) + a2 (checked against numpy 2.2 in
// tests/test_host_cpu.py::test_numpy_three_element_sum_order).
__global__ __launch_bounds__(kThreads) void k_guro(float *__restrict__ cb, const float *__restrict__ nb,
                                                   float l0, float l1, float l2,
                                                   size_t first_pix, size_t npix)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npix; i += stride) {
        const size_t pix = first_pix + i;
        const float n0 = nb[pix * 3], n1 = nb[pix * 3 + 1], n2 = nb[pix * 3 + 2];
        const float s = ((0.0f + n0 * l0) + n1 * l1) + n2 * l2;     // (the reduction starts from +0: see guro_factor)
        const float m = sqrtf((n0 * n0 + n1 * n1) + n2 * n2);
        float c = s / (m + 1e-6f);
        c = c < 0.0f ? 0.0f : c;  // np.clip keeps a NaN a NaN
        c = c > 1.0f ? 1.0f : c;
        cb[pix * 3] *= c;
        cb[pix * 3 + 1] *= c;
        cb[pix * 3 + 2] *= c;
    }
}

// ---- f3: presentation, run.py:26 — image[::-1].astype('uint8') ----------------------
// float32 -> uint8 as numpy's C cast does it on x86-64: truncate toward zero to int32
// (cvttss2si: NaN / out of range -> INT_MIN), keep the low byte.  Rows are flipped.
__global__ __launch_bounds__(kThreads) void k_present_u8(const float *__restrict__ cb,
                                                         unsigned char *__restrict__ out, int H,
                                                         int W, int flip)
{
    const size_t row_elems = (size_t)W * 3, n = (size_t)H * row_elems;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const size_t y = i / row_elems, r = i - y * row_elems;
        const float v = cb[(flip ? (size_t)(H - 1) - y : y) * row_elems + r];
        int iv = (int)0x80000000;
        if (v > -2147483904.0f && v < 2147483648.0f) iv = (int)v;
        out[i] = (unsigned char)(iv & 0xFF);
    }
}

// ---- f2: model transforms on resident vertex buffers (reference: cy/data_structures/model.py:153-236)
// shift / scale / mean vertex / max span / the *_by_triangles gathers, each in numpy's own float32
// (or, where numpy promotes, float64) operation order, so that a device-resident model hands the
// filler the very arrays the reference's Model would (tests: bit for bit against the host Model).
// rotate and the vertex-normal computation (below) go through numpy's BLAS on the host: the device
// spells out what that BLAS computes for 3-vectors (dot3_f32) and agrees with the host Model bit for
// bit on every mesh tested; what is NOT the reference's property is the BLAS build itself.
__global__ __launch_bounds__(kThreads) void k_model_shift(float *__restrict__ v, size_t n, double s0, double s1,
                                                          double s2, int in_double)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % 3);
        const double s = c == 0 ? s0 : c == 1 ? s1 : s2;
        // vertices + shift: float32 + float32 array stays float32; a Python list or a float64
        // array promotes the sum to float64, rounded once when the Model stores float32
        v[i] = in_double ? (float)((double)v[i] + s) : v[i] + (float)s;
    }
}

// vtx -= mean; vtx *= coef; vtx += mean, three float32 passes in place (model.py:222-228)
__global__ __launch_bounds__(kThreads) void k_model_scale(float *__restrict__ v, size_t n,
                                                          const float *__restrict__ mean3, float coef, int keep)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        float x = v[i];
        if (keep) {
            const float m = mean3[i % 3];
            x = x - m;
            x = x * coef;
            x = x + m;
        } else {
            x = x * coef;
        }
        v[i] = x;
    }
}

// vertices.mean(axis=0): numpy adds the rows one after another into a float32 accumulator per
// component — the first row is the start value, there is no pairwise summation along a strided
// axis — and divides by the intp count in float64, stored as float32.  One lane per component; the
// chain of additions is inherently serial (4 ns each), the LOADS are not: 32 rows are requested
// together before their 32 additions (one dependent load per row was 0.3 us per vertex: seconds
// for a model of millions of vertices).
__global__ void k_model_mean(const float *__restrict__ v, int64_t V, float *__restrict__ mean3)
{
    const int c = threadIdx.x;
    if (c >= 3) return;
    constexpr int kAhead = 32;
    float acc = v[c];
    int64_t i = 1;
    for (; i + kAhead <= V; i += kAhead) {
        float x[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) x[k] = v[(i + k) * 3 + c];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) acc = acc + x[k];
    }
    for (; i < V; ++i) acc = acc + v[i * 3 + c];
    mean3[c] = (float)((double)acc / (double)V);
}

// max over vertices of ||v - mean|| with numpy's float32 norm: sqrt((dx*dx + dy*dy) + dz*dz).
// Non-negative floats order like their bit patterns: an unsigned atomic max is exact.
__global__ __launch_bounds__(kThreads) void k_model_max_span(const float *__restrict__ v, int64_t V,
                                                             const float *__restrict__ mean3,
                                                             uint32_t *__restrict__ out_bits)
{
    const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2];
    float best = 0.0f;
    bool nan = false;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < V; i += stride) {
        const float dx = v[i * 3] - m0, dy = v[i * 3 + 1] - m1, dz = v[i * 3 + 2] - m2;
        const float r = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (r != r) nan = true;          // np.max propagates a NaN
        else if (r > best) best = r;
    }
    atomicMax(out_bits, nan ? 0x7FC00000u : __float_as_uint(best));
}

// out[t][k][:] = attr[index[t][k]][:]  (vertices[faces], normals[faces_n], colours[faces_t])
__global__ __launch_bounds__(kThreads) void k_model_gather(const float *__restrict__ attr,
                                                           const int32_t *__restrict__ index,
                                                           float *__restrict__ out, int64_t n_corners)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_corners; i += stride) {
        const int64_t j = index[i];
        out[i * 3] = attr[j * 3]; out[i * 3 + 1] = attr[j * 3 + 1]; out[i * 3 + 2] = attr[j * 3 + 2];
    }
}

// Model.rotate (model.py:238-256): new_vertices = np.matmul(vertices, mat_rot.T) — float32 vertices
// times a float64 matrix, so numpy promotes: every coordinate is a three-term float64 dot product,
// rounded once when _update_vertices_and_normals stores float32.  The matrix comes from the host
// (three 2x2 blocks composed with numpy in float64, as the reference does).  The summation order
// inside numpy's matmul belongs to its BLAS build; in float64 it survives the rounding to float32
// only when the sum sits within ~1e-16 of a float32 rounding boundary.
__global__ __launch_bounds__(kThreads) void k_model_rotate(float *__restrict__ v, int64_t V, double r00, double r01,
                                                           double r02, double r10, double r11, double r12,
                                                           double r20, double r21, double r22)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < V; i += stride) {
        const double x = (double)v[i * 3], y = (double)v[i * 3 + 1], z = (double)v[i * 3 + 2];
        v[i * 3] = (float)((x * r00 + y * r01) + z * r02);
        v[i * 3 + 1] = (float)((x * r10 + y * r11) + z * r12);
        v[i * 3 + 2] = (float)((x * r20 + y * r21) + z * r22);
    }
}

// Model._compute_normals_by_vertex (model.py:175-208), step 1: the unit normal of every face,
// n = -cross(v1 - v0, v1 - v2) in float32 (numpy's cross: each product rounded, then the
// difference), divided by its norm unless that is 0.  The norm is np.linalg.norm = sqrt(n . n):
// np.dot of two float32 3-vectors as numpy computes it: its BLAS (OpenBLAS sdot, kernel/x86_64/sdot.c)
// forms the products in float32 and adds them up in a DOUBLE accumulator, rounding to float32 once at
// the end — 200 000 random and near-parallel vector pairs agree with this spelling bit for bit, where the
// plain float32 sum agrees on 81 % (tests/test_host_cpu.py::test_numpy_dot_of_3_vectors).  np.linalg.norm
// of a float32 vector is the float32 square root of that dot.
CR_DEV float dot3_f32(const float a[3], const float b[3])
{
    const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
    return (float)(((double)p0 + (double)p1) + (double)p2);
}
CR_DEV void unit3_f32(float n[3])
{
    const float len = sqrtf(dot3_f32(n, n));
    if (len == 0.0f) return;
    n[0] = n[0] / len; n[1] = n[1] / len; n[2] = n[2] / len;
}
__global__ __launch_bounds__(kThreads) void k_model_face_normals(const float *__restrict__ v,
                                                                 const int32_t *__restrict__ faces, int64_t T,
                                                                 float *__restrict__ fn)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < T; t += stride) {
        const float *p0 = v + (int64_t)faces[t * 3] * 3, *p1 = v + (int64_t)faces[t * 3 + 1] * 3,
                    *p2 = v + (int64_t)faces[t * 3 + 2] * 3;
        const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        const float b[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        float n[3] = {-(a[1] * b[2] - a[2] * b[1]), -(a[2] * b[0] - a[0] * b[2]), -(a[0] * b[1] - a[1] * b[0])};
        unit3_f32(n);
        fn[t * 3] = n[0]; fn[t * 3 + 1] = n[1]; fn[t * 3 + 2] = n[2];
    }
}
// Step 2: one thread per vertex walks the faces that name it, in face order (offs / occ: a CSR made
// once at upload, one entry per (face, corner) occurrence), collects a face normal unless an
// already collected one has a dot product >= 1 with it (`taken`: one byte per occurrence), and
// stores the normalised mean of the collected ones — numpy's mean over axis 0: rows added one after
// another in float32, divided by the count (float64 division, as for vertices.mean) — or zeros for
// a vertex no face names.
__global__ __launch_bounds__(kThreads) void k_model_vertex_normals(const float *__restrict__ fn,
                                                                   const int32_t *__restrict__ offs,
                                                                   const int32_t *__restrict__ occ,
                                                                   unsigned char *__restrict__ taken, int64_t V,
                                                                   float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < V; i += stride) {
        const int32_t a = offs[i], b = offs[i + 1];
        float acc[3] = {0.0f, 0.0f, 0.0f};
        int m = 0;
        for (int32_t j = a; j < b; ++j) {
            const float *nj = fn + (int64_t)occ[j] * 3;
            const float n[3] = {nj[0], nj[1], nj[2]};
            bool fresh = true;
            for (int32_t k = a; k < j; ++k) {
                if (!taken[k]) continue;
                const float *nk = fn + (int64_t)occ[k] * 3;
                const float e[3] = {nk[0], nk[1], nk[2]};
                if (dot3_f32(e, n) >= 1.0f) fresh = false;         // (NaN: not a duplicate)
            }
            taken[j] = fresh ? 1 : 0;
            if (fresh) {
                acc[0] = acc[0] + n[0]; acc[1] = acc[1] + n[1]; acc[2] = acc[2] + n[2];
                ++m;
            }
        }
        float r[3] = {0.0f, 0.0f, 0.0f};
        if (m > 0) {
            r[0] = (float)((double)acc[0] / (double)m);
            r[1] = (float)((double)acc[1] / (double)m);
            r[2] = (float)((double)acc[2] / (double)m);
            unit3_f32(r);
        }
        out[i * 3] = r[0]; out[i * 3 + 1] = r[1]; out[i * 3 + 2] = r[2];
    }
}

// Row f4's device part — model.py:143-151: one colour per texture coordinate, the nearest texel
// with the v axis flipped, as float32:
//   row = clip(int32((1 - v) * h), 0, h - 1), column = clip(int32(u * w), 0, w - 1)
// in numpy's float32 arithmetic; the cast is the host's truncating conversion, which yields
// INT_MIN (so, after the clip, texel 0) for a NaN and for anything outside int32.
CR_DEV int host_f32_to_i32(float f)
{
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
}
__global__ __launch_bounds__(kThreads) void k_model_texture_colors(const float *__restrict__ uv, int uv_cols,
                                                                   int64_t n, const unsigned char *__restrict__ tex,
                                                                   int th, int tw, float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float u = uv[i * uv_cols], v = uv[i * uv_cols + 1];
        int row = host_f32_to_i32((1.0f - v) * (float)th);
        int colm = host_f32_to_i32(u * (float)tw);
        row = row < 0 ? 0 : (row > th - 1 ? th - 1 : row);
        colm = colm < 0 ? 0 : (colm > tw - 1 ? tw - 1 : colm);
        const unsigned char *t = tex + ((size_t)row * tw + colm) * 3;
        out[i * 3] = (float)t[0]; out[i * 3 + 1] = (float)t[1]; out[i * 3 + 2] = (float)t[2];
    }
}

// Sort key of a triangle for the tile-coherent order: Morton code of the 4-pixel CELL its projected
// centroid falls in — hence of its 8-, 16-, 32- and 64-pixel tile too (a Morton code's prefixes are the
// coarser cells') — an ordering heuristic only: nothing exact depends on it.  Cells finer than the
// raster tile keep neighbours in the frame neighbours in memory WITHIN a tile as well: the records a
// tile takes from its neighbours' clusters (the sixth of its list whose centroid lies across the border)
// sit in the few cells along that border instead of anywhere in the cluster, and the winners of a
// wavefront's two pixel rows in a few runs of records.  10 M small triangles at 4096^2, same box:
// cells of 32 / 16 / 8 / 4 / 2 pixels -> 2 175 / 2 057 / 2 002 / 1 948 / 1 938 MB fetched per raster
// launch, 0.602 / 0.581 / 0.563 / 0.560 / 0.558 ms (profiles/r05/ab_order_cell_synth10m.txt).
__global__ __launch_bounds__(kThreads) void k_tile_order_keys(const float *__restrict__ tri, int64_t T,
                                                              ProjConst P, int W, int H,
                                                              uint32_t *__restrict__ keys, int cell_shift)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < T; t += stride) {
        const float *v = tri + t * 9;
        float c[3] = {(v[0] + v[3] + v[6]) * (1.0f / 3.0f), (v[1] + v[4] + v[7]) * (1.0f / 3.0f),
                      (v[2] + v[5] + v[8]) * (1.0f / 3.0f)};
        project_vertex(P, c);
        int x = (int)c[0], y = (int)c[1];
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        if (!(c[0] == c[0]) || !(c[1] == c[1])) x = y = 0;
        uint32_t a = (uint32_t)x >> cell_shift, b = (uint32_t)y >> cell_shift, m = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) m |= ((a >> k) & 1u) << (2 * k) | ((b >> k) & 1u) << (2 * k + 1);
        keys[t] = m;
    }
}

}  // namespace

extern "C" {

int crender_present_u8(const float *d_color, unsigned char *d_out, int H, int W, int flip_rows,
                       void *stream)
{
    if (!d_color || !d_out || H <= 0 || W <= 0)
        return fail(CRENDER_EINVAL, "crender_present_u8: bad argument");
    hipLaunchKernelGGL(k_present_u8, dim3(grid_for((size_t)H * W * 3, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_color, d_out, H, W, flip_rows);
    CR_LAUNCH_CHECK("k_present_u8");
    return CRENDER_OK;
}

int crender_guro_illumination(float *d_color, const float *d_normal, const float *light3, int H, int W,
                              int y0, int y1, void *stream)
{
    if (!d_color || !d_normal || !light3 || H <= 0 || W <= 0 || y0 < 0 || y1 > H || y0 >= y1)
        return fail(CRENDER_EINVAL, "crender_guro_illumination: bad argument");
    const size_t first = (size_t)y0 * W, npix = (size_t)(y1 - y0) * W;
    hipLaunchKernelGGL(k_guro, dim3(grid_for(npix, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_color, d_normal, light3[0], light3[1],
                       light3[2], first, npix);
    CR_LAUNCH_CHECK("k_guro");
    return CRENDER_OK;
}

int crender_tile_order_keys(const float *d_tri, int64_t T, const float *P16, int w, int h, uint32_t *d_keys,
                            void *stream)
{
    if (T < 0 || !P16 || w <= 0 || h <= 0 || (T > 0 && (!d_tri || !d_keys)))
        return fail(CRENDER_EINVAL, "crender_tile_order_keys: bad argument");
    if (T == 0) return CRENDER_OK;
    int cell_shift = 2;      // 4-pixel cells
#ifdef CRENDER_DEV_KNOBS
    if (std::getenv("CRENDER_ORDER_CELL_SHIFT")) cell_shift = std::atoi(std::getenv("CRENDER_ORDER_CELL_SHIFT"));
#endif
    hipLaunchKernelGGL(k_tile_order_keys, dim3(grid_for((size_t)T, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_tri, T, make_proj(P16, w, h), w, h, d_keys, cell_shift);
    CR_LAUNCH_CHECK("k_tile_order_keys");
    return CRENDER_OK;
}

// ---- f2: model transforms (cy/data_structures/model.py:153-236) -----------------------------------
int crender_model_shift(float *d_vertices, int64_t V, const double *shift3, int shift_is_float32, void *stream)
{
    if (V < 0 || !shift3 || (V > 0 && !d_vertices)) return fail(CRENDER_EINVAL, "crender_model_shift: bad argument");
    if (V == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_model_shift, dim3(grid_for((size_t)V * 3, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_vertices, (size_t)V * 3, shift3[0], shift3[1],
                       shift3[2], shift_is_float32 ? 0 : 1);
    CR_LAUNCH_CHECK("k_model_shift");
    return CRENDER_OK;
}

int crender_model_scale(float *d_vertices, int64_t V, const float *d_mean3, float coef, int keep_position,
                        void *stream)
{
    if (V < 0 || (V > 0 && !d_vertices) || (keep_position && !d_mean3))
        return fail(CRENDER_EINVAL, "crender_model_scale: bad argument");
    if (V == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_model_scale, dim3(grid_for((size_t)V * 3, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_vertices, (size_t)V * 3, d_mean3, coef, keep_position);
    CR_LAUNCH_CHECK("k_model_scale");
    return CRENDER_OK;
}

int crender_model_stats(const float *d_vertices, int64_t V, float *d_mean3, float *d_max_span, void *stream)
{
    if (V <= 0 || !d_vertices || !d_mean3 || !d_max_span)
        return fail(CRENDER_EINVAL, "crender_model_stats: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_model_mean, dim3(1), dim3(64), 0, s, d_vertices, V, d_mean3);
    CR_LAUNCH_CHECK("k_model_mean");
    CR_HIP(hipMemsetAsync(d_max_span, 0, sizeof(float), s));
    hipLaunchKernelGGL(k_model_max_span, dim3(grid_for((size_t)V, 1024)), dim3(kThreads), 0, s, d_vertices, V,
                       d_mean3, reinterpret_cast<uint32_t *>(d_max_span));
    CR_LAUNCH_CHECK("k_model_max_span");
    return CRENDER_OK;
}

int crender_model_texture_colors(const float *d_uv, int uv_cols, int64_t n, const unsigned char *d_texture,
                                 int th, int tw, float *d_out, void *stream)
{
    if (n < 0 || uv_cols < 2 || th <= 0 || tw <= 0 || th > (1 << 24) || tw > (1 << 24) ||
        (n > 0 && (!d_uv || !d_texture || !d_out)))
        return fail(CRENDER_EINVAL, "crender_model_texture_colors: bad argument");
    if (n == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_model_texture_colors, dim3(grid_for((size_t)n, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_uv, uv_cols, n, d_texture, th, tw, d_out);
    CR_LAUNCH_CHECK("k_model_texture_colors");
    return CRENDER_OK;
}

int crender_model_rotate(float *d_vertices, int64_t V, const double *R9, void *stream)
{
    if (V < 0 || !R9 || (V > 0 && !d_vertices)) return fail(CRENDER_EINVAL, "crender_model_rotate: bad argument");
    if (V == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_model_rotate, dim3(grid_for((size_t)V, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_vertices, V, R9[0], R9[1], R9[2], R9[3], R9[4], R9[5],
                       R9[6], R9[7], R9[8]);
    CR_LAUNCH_CHECK("k_model_rotate");
    return CRENDER_OK;
}

int crender_model_vertex_normals(const float *d_vertices, int64_t V, const int32_t *d_faces, int64_t T,
                                 const int32_t *d_offsets, const int32_t *d_occurrences, float *d_face_normals,
                                 unsigned char *d_taken, float *d_normals, void *stream)
{
    if (V < 0 || T < 0 || (V > 0 && (!d_vertices || !d_offsets || !d_normals)) ||
        (T > 0 && (!d_faces || !d_occurrences || !d_face_normals || !d_taken)))
        return fail(CRENDER_EINVAL, "crender_model_vertex_normals: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (T > 0) {
        hipLaunchKernelGGL(k_model_face_normals, dim3(grid_for((size_t)T, 4096)), dim3(kThreads), 0, s, d_vertices,
                           d_faces, T, d_face_normals);
        CR_LAUNCH_CHECK("k_model_face_normals");
    }
    if (V > 0) {
        hipLaunchKernelGGL(k_model_vertex_normals, dim3(grid_for((size_t)V, 4096)), dim3(kThreads), 0, s,
                           d_face_normals, d_offsets, d_occurrences, d_taken, V, d_normals);
        CR_LAUNCH_CHECK("k_model_vertex_normals");
    }
    return CRENDER_OK;
}

int crender_model_gather(const float *d_attr, const int32_t *d_index, float *d_out, int64_t T, void *stream)
{
    if (T < 0 || (T > 0 && (!d_attr || !d_index || !d_out)))
        return fail(CRENDER_EINVAL, "crender_model_gather: bad argument");
    if (T == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_model_gather, dim3(grid_for((size_t)T * 3, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_attr, d_index, d_out, T * 3);
    CR_LAUNCH_CHECK("k_model_gather");
    return CRENDER_OK;
}

}  // extern "C"
