/*
 * crender_oracle.c — CPU restatement of the reference's Version-C rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed CPU
 * baseline ("port") for the MI355X HIP path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product package
 * (cython3dmodelrenderer_amd/) never imports, links or executes anything here.
 *
 * Reference followed (paths relative to /root/reference):
 *   crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx  (".pyx")
 *   crender/cy/pixel_buffer_filler/math_utils.pyx / math_utils.pxd   ("mu.pyx", "mu.pxd")
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - oracle_bar() is checked against the reference's own math_utils.pyx compiled
 *     from where it lies (oracle/build_ref.sh -> oracle/_ref/), bit for bit.
 *   - the whole pipeline is checked against the reference's committed render
 *     output/T-Rex.png and against the reference-run buffer hashes and work counts
 *     that SURVEY.md section 8c / 8a-a10 record for cube@256, T-Rex@256, T-Rex@1024.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 * FMA contraction MUST stay off: every float op below is a separately rounded
 * IEEE binary32 operation in the reference (gcc -O2, x86-64 SSE, no -march).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <omp.h>

#define ORACLE_API __attribute__((visibility("default")))

/* Work counters, same meaning as SURVEY.md section 8a row a10's probe columns. */
typedef struct {
    int64_t culled;        /* back-face culled triangles                  .pyx:202 */
    int64_t empty;         /* triangles with an empty pixel box           .pyx:209 */
    int64_t drawn;         /* triangles that reach the pixel loops                 */
    int64_t bbox_samples;  /* (x, y) iterations of the two pixel loops    .pyx:213 */
    int64_t inside;        /* samples passing the barycentric sign test   .pyx:216 */
    int64_t writes;        /* samples passing the NaN and depth tests     .pyx:223 */
} oracle_stats;

/* ---- a2/a3: scalars and the projection matrix ------------------------------
 * .pyx:54-59: fov is stored to a C float, f = 1/tan(fov/2/180*pi) is evaluated in
 * double from that float and stored to a C float; a = h/w (Python true division)
 * stored to a C float.
 * .pyx:83-90: q = z_far/(z_far - z_near) in C float arithmetic; f/a in C float
 * arithmetic; -z_near*q is a double product of two floats (exact) rounded once to
 * float32 by np.array(dtype='float32') == the correctly rounded float product.
 * Row-major P[4][4]. */
ORACLE_API void oracle_projection_matrix(double fov, double z_near, double z_far,
                                         int h, int w, float *P)
{
    float fovf = (float)fov;
    float f = (float)(1.0 / tan((double)fovf / 2 / 180 * M_PI));
    float zn = (float)z_near, zf = (float)z_far;
    float a = (float)((double)h / (double)w);
    float q = zf / (zf - zn);
    memset(P, 0, 16 * sizeof(float));
    P[0 * 4 + 0] = f / a;
    P[1 * 4 + 1] = f;
    P[2 * 4 + 2] = q;
    P[2 * 4 + 3] = 1.0f;
    P[3 * 4 + 2] = (float)((double)(-zn) * (double)q);
}

/* ---- a5 / K1: projection, .pyx:106-130 --------------------------------------
 * render_model passes the same array as source and destination (.pyx:99), so
 * column j reads the already overwritten columns < j.  That is reproduced by
 * working on a local copy of the vertex that is updated column by column.
 * x_scale = w/2.0, y_scale = h/2.0 are doubles stored to C floats (.pyx:109). */
static inline void project_vertex(const float *P, float xs, float ys, float *v)
{
    float z = v[2];
    for (int j = 0; j < 3; ++j)
        v[j] = v[0] * P[0 * 4 + j] + v[1] * P[1 * 4 + j] + v[2] * P[2 * 4 + j] + P[3 * 4 + j];
    v[0] = v[0] / z;
    v[1] = v[1] / z;
    v[2] = v[2] / z;
    v[0] = v[0] + 1.0f;
    v[1] = v[1] + 1.0f;
    v[0] = v[0] * xs;
    v[1] = v[1] * ys;
}

ORACLE_API void oracle_project(const float *tri_in, float *tri_out, int64_t T,
                               const float *P, int w, int h, int n_threads)
{
    float xs = (float)((double)w / 2.0), ys = (float)((double)h / 2.0);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t t = 0; t < T; ++t) {
        for (int i = 0; i < 3; ++i) {
            float v[3];
            memcpy(v, tri_in + t * 9 + i * 3, sizeof v);
            project_vertex(P, xs, ys, v);
            memcpy(tri_out + t * 9 + i * 3, v, sizeof v);
        }
    }
}

/* ---- a7: clip, mu.pxd:8-13 --------------------------------------------------- */
static inline int clipi(int a, int lo, int hi)
{
    if (a < lo) return lo;
    if (a > hi) return hi;
    return a;
}

/* <int>ceil(x) as the reference's x86-64 build evaluates it (.pyx:165-166):
 * cvttsd2si of a double outside int range returns INT_MIN.  NaN cannot reach
 * here (the min/max scan below never selects a NaN). */
static inline int ceil_to_int(float x)
{
    double c = ceil((double)x);
    if (!(c >= -2147483648.0 && c <= 2147483647.0)) return INT_MIN;
    return (int)c;
}

/* ---- a6: pixel box of a projected triangle, .pyx:132-175 ----------------------
 * out = {x_left, x_right, y_top, y_bot}; pixel range is [left, right) x [top, bot). */
ORACLE_API void oracle_bbox(const float *tri, int w, int h, int *out)
{
    float xl = (float)w, xr = 0.0f, yt = (float)h, yb = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float x = tri[i * 3], y = tri[i * 3 + 1];
        if (x < xl) xl = x;
        if (x > xr) xr = x;
        if (y < yt) yt = y;
        if (y > yb) yb = y;
    }
    out[0] = clipi(ceil_to_int(xl), 0, w);
    out[1] = clipi(ceil_to_int(xr), 0, w);
    out[2] = clipi(ceil_to_int(yt), 0, h);
    out[3] = clipi(ceil_to_int(yb), 0, h);
}

/* ---- a9: barycentric coordinates of integer pixel (x, y), mu.pyx:8-34 ---------
 * The nine edge constants are recomputed per pixel, exactly as the reference does,
 * and every division is a correctly rounded float division (cdivision(True): a
 * zero denominator silently yields +-inf / NaN). */
ORACLE_API void oracle_bar(const float *tri, int x, int y, float *b)
{
    float x0 = tri[0], y0 = tri[1];
    float x1 = tri[3], y1 = tri[4];
    float x2 = tri[6], y2 = tri[7];
    float l01 = x1 - x2, l02 = y1 - y2;
    float l03 = l01 * (y0 - y2) - l02 * (x0 - x2);
    float l11 = x2 - x0, l12 = y2 - y0;
    float l13 = l11 * (y1 - y0) - l12 * (x1 - x0);
    float l21 = x0 - x1, l22 = y0 - y1;
    float l23 = l21 * (y2 - y1) - l22 * (x2 - x1);
    float X = (float)x, Y = (float)y;
    b[0] = (l01 * (Y - y2) - l02 * (X - x2)) / l03;
    b[1] = (l11 * (Y - y0) - l12 * (X - x0)) / l13;
    b[2] = (l21 * (Y - y1) - l22 * (X - x1)) / l23;
}

/* One fragment: everything between the barycentrics and the store, .pyx:215-242.
 * Returns 0 = outside, 1 = inside but rejected (NaN or behind), 2 = wins. */
static inline int shade_sample(const float *tri, const float *col, const float *nrm,
                               int x, int y, float zcur, float *out7)
{
    float b[3];
    oracle_bar(tri, x, y, b);
    if (b[0] < 0.0f || b[1] < 0.0f || b[2] < 0.0f) return 0;          /* .pyx:216 */
    float new_z = tri[2] * b[0] + tri[5] * b[1] + tri[8] * b[2];      /* .pyx:219 */
    if (!(-1.0 <= new_z || new_z <= 1.0)) return 1;                   /* .pyx:220, NaN only */
    if (new_z > zcur) return 1;                                       /* .pyx:223, write on <= */
    out7[0] = new_z;
    for (int k = 0; k < 3; ++k)                                       /* .pyx:229-231 */
        out7[1 + k] = col[k] * b[0] + col[3 + k] * b[1] + col[6 + k] * b[2];
    for (int k = 0; k < 3; ++k)                                       /* .pyx:226-228 */
        out7[4 + k] = nrm[k] * b[0] + nrm[3 + k] * b[1] + nrm[6 + k] * b[2];
    return 2;
}

/* .pyx:202: (n0z + n1z + n2z) is a float sum; "/ 3" is emitted by Cython as a
 * double division by 3.0; the comparison is in double. */
static inline int backface(const float *nrm)
{
    float s = nrm[2] + nrm[5] + nrm[8];
    return ((double)s / 3.0) >= 0.0;
}

/* ---- a10 / K2, serial semantics (the deterministic 1-thread reference order) ---
 * Rows are restricted to the strip [y0, y1) (multi-GPU row strips; the full frame
 * is y0 = 0, y1 = H).  winner (optional, int32[H*W]) receives the index of the
 * triangle whose fragment is stored last at each pixel; untouched where nothing
 * is written.  st (optional) receives the work counters. */
ORACLE_API void oracle_raster_serial(const float *tri, const float *col, const float *nrm,
                                     int64_t T, float *zbuf, float *cbuf, float *nbuf,
                                     int H, int W, int y0, int y1,
                                     int32_t *winner, oracle_stats *st)
{
    oracle_stats s = {0, 0, 0, 0, 0, 0};
    for (int64_t t = 0; t < T; ++t) {
        const float *tr = tri + t * 9, *cl = col + t * 9, *nr = nrm + t * 9;
        if (backface(nr)) { s.culled++; continue; }
        int bb[4];
        oracle_bbox(tr, W, H, bb);
        if (bb[0] - bb[1] == 0 || bb[2] - bb[3] == 0) { s.empty++; continue; }  /* .pyx:209 */
        s.drawn++;
        int ya = bb[2] > y0 ? bb[2] : y0, yb = bb[3] < y1 ? bb[3] : y1;
        for (int x = bb[0]; x < bb[1]; ++x) {                        /* x outer, .pyx:213 */
            for (int y = ya; y < yb; ++y) {                          /* y inner, .pyx:214 */
                float o[7];
                size_t p = (size_t)y * W + x;
                s.bbox_samples++;
                int r = shade_sample(tr, cl, nr, x, y, zbuf[p], o);
                if (r == 0) continue;
                s.inside++;
                if (r == 1) continue;
                s.writes++;
                zbuf[p] = o[0];
                memcpy(cbuf + p * 3, o + 1, 3 * sizeof(float));
                memcpy(nbuf + p * 3, o + 4, 3 * sizeof(float));
                if (winner) winner[p] = (int32_t)t;
            }
        }
    }
    if (st) *st = s;
}

/* ---- a10 / K2, Version-C shape: the timed CPU baseline ------------------------
 * Same arithmetic; the reference's parallel structure: one dynamic-scheduled
 * OpenMP loop over triangles (chunk 1), the depth test outside the lock, one
 * omp_lock_t per pixel around the store (.pyx:196-244).  With n_threads > 1 it
 * has the reference's own benign-looking race; n_threads = 1 is deterministic.
 * lock_grid is allocated and initialised by oracle_locks_create (the reference
 * does it in __cinit__, .pyx:73-77, outside the timed path). */
ORACLE_API void *oracle_locks_create(int H, int W)
{
    size_t n = (size_t)H * W;
    omp_lock_t *g = (omp_lock_t *)malloc(n * sizeof(omp_lock_t));
    if (!g) return NULL;
    for (size_t i = 0; i < n; ++i) omp_init_lock(&g[i]);
    return g;
}

ORACLE_API void oracle_locks_destroy(void *locks, int H, int W)
{
    omp_lock_t *g = (omp_lock_t *)locks;
    if (!g) return;
    size_t n = (size_t)H * W;
    for (size_t i = 0; i < n; ++i) omp_destroy_lock(&g[i]);
    free(g);
}

ORACLE_API void oracle_raster_omp(const float *tri, const float *col, const float *nrm,
                                  int64_t T, float *zbuf, float *cbuf, float *nbuf,
                                  int H, int W, int y0, int y1,
                                  void *locks, int n_threads)
{
    omp_lock_t *grid = (omp_lock_t *)locks;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for (int64_t t = 0; t < T; ++t) {
        const float *tr = tri + t * 9, *cl = col + t * 9, *nr = nrm + t * 9;
        if (backface(nr)) continue;
        int bb[4];
        oracle_bbox(tr, W, H, bb);
        if (bb[0] - bb[1] == 0 || bb[2] - bb[3] == 0) continue;
        int ya = bb[2] > y0 ? bb[2] : y0, yb = bb[3] < y1 ? bb[3] : y1;
        for (int x = bb[0]; x < bb[1]; ++x) {
            for (int y = ya; y < yb; ++y) {
                float o[7];
                size_t p = (size_t)y * W + x;
                /* unlocked read of the depth buffer, as .pyx:223 */
                float zcur = *(volatile float *)&zbuf[p];
                if (shade_sample(tr, cl, nr, x, y, zcur, o) != 2) continue;
                omp_set_lock(&grid[p]);
                zbuf[p] = o[0];
                memcpy(cbuf + p * 3, o + 1, 3 * sizeof(float));
                memcpy(nbuf + p * 3, o + 4, 3 * sizeof(float));
                omp_unset_lock(&grid[p]);
            }
        }
    }
}

/* ---- a4: render_model = copy + K1 in place + K2, .pyx:92-104 -------------------
 * scratch must hold T*9 floats (the reference's .copy() of the vertex array). */
ORACLE_API void oracle_render_model(const float *tri, const float *col, const float *nrm,
                                    int64_t T, const float *P,
                                    float *zbuf, float *cbuf, float *nbuf, int H, int W,
                                    float *scratch, void *locks, int n_threads)
{
    oracle_project(tri, scratch, T, P, W, H, n_threads);
    if (locks)
        oracle_raster_omp(scratch, col, nrm, T, zbuf, cbuf, nbuf, H, W, 0, H, locks, n_threads);
    else
        oracle_raster_serial(scratch, col, nrm, T, zbuf, cbuf, nbuf, H, W, 0, H, NULL, NULL);
}

/* Buffer initialisation of __cinit__, .pyx:65-67: z = 1e6, colour = normal = 0. */
ORACLE_API void oracle_clear(float *zbuf, float *cbuf, float *nbuf, int H, int W, int n_threads)
{
    size_t n = (size_t)H * W;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (size_t i = 0; i < n; ++i) {
        zbuf[i] = 1e6f;
        cbuf[3 * i] = cbuf[3 * i + 1] = cbuf[3 * i + 2] = 0.0f;
        nbuf[3 * i] = nbuf[3 * i + 1] = nbuf[3 * i + 2] = 0.0f;
    }
}

/* ---- next row f1: GuroIllumination.draw_illumination, guro_illumination.py:20-27 -----------
 * The reference's four numpy statements on float32 buffers, one pixel at a time:
 *   scalar_product = np.sum(n * light, axis=-1)        -> ((0 + n0*l0) + n1*l1) + n2*l2   (numpy's
 *   norm = np.linalg.norm(n, axis=-1)                  -> sqrt((n0*n0 + n1*n1) + n2*n2)  3-element
 *   shadow = clip(scalar_product / (norm + 1e-6), 0, 1)                            add.reduce order,
 *   colour *= shadow                                                               pinned against numpy in
 * tests/test_oracle_cpu.py::test_oracle_guro_matches_numpy_statements). */
ORACLE_API void oracle_guro(float *color, const float *normal, const float *light3, int64_t npix)
{
    const float l0 = light3[0], l1 = light3[1], l2 = light3[2];
    for (int64_t i = 0; i < npix; ++i) {
        const float n0 = normal[3 * i], n1 = normal[3 * i + 1], n2 = normal[3 * i + 2];
        /* numpy's add.reduce over a short axis starts from +0: three -0 products (a zero normal
         * under a light along -z) sum to +0, not -0 */
        const float s = ((0.0f + n0 * l0) + n1 * l1) + n2 * l2;
        const float m = sqrtf((n0 * n0 + n1 * n1) + n2 * n2);
        float f = s / (m + 1e-6f);
        f = f < 0.0f ? 0.0f : f;        /* np.clip keeps a NaN (and a -0) */
        f = f > 1.0f ? 1.0f : f;
        color[3 * i] *= f;
        color[3 * i + 1] *= f;
        color[3 * i + 2] *= f;
    }
}
