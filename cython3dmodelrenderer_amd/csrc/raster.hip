// raster.hip — K2 on gfx950 (MI355X / CDNA4): the tile rasterizer, the swap chain's one-launch frame
// kernel and the independent global-atomic implementation; their host side.
//
// Frame = K1 (projection) + K2 (rasterization) of the reference's
// AdvancedPixelBufferFiller.render_model (.pyx:92-244), restructured for the GPU:
//
//   k_setup   one thread per triangle: [project,] back-face cull, pixel box, tile range, and
//             the binning.  Scenes of up to 65536 triangles append their indices straight
//             into fixed-capacity per-tile lists ("direct bins": the frame is two launches);
//             larger scenes count list lengths here and go through
//   k_scan    one workgroup: exclusive scan of the tile list lengths, and
//   k_fill    writes triangle indices into the scanned lists (block-private LDS cursors,
//             one global atomic per (block, touched tile) to reserve list space).
//   k_raster  one workgroup per screen tile: a 64-bit (z, index) key per pixel lives in
//             LDS; the tile's list is swept in 4x4-pixel blocks, flattened and split evenly
//             over sixteen 16-lane groups, with LDS atomic-min; then every pixel recomputes
//             its winning fragment and stores z / colour / normal once (the clear is fused).
//
// No HBM atomics on the framebuffer and every framebuffer byte is written once per frame.
// Build flags (see _build.py): -ffp-contract=off, correctly rounded division, denormals on —
// float parity with the reference depends on them.
//
// Measurement knobs exist only in a development build (-DCRENDER_DEV_KNOBS, scripts/dev_build.sh):
// there CRENDER_DEBUG (environment, read once) is a bit mask; the product library is compiled
// without them (every `dbg & bit` below folds to 0):
//   1 no coverage work, 2 no shading (both produce WRONG images: ablation timing only),
//   4 block-histogram count / fill passes on every scan-path frame, 8 XCD-banded tile map, 16 never use direct bins, 32 no sign rejection, 64 no hoisted
//   reciprocal, 128 small-record block sweep for every batch (64-pixel tiles), 256 invert the
//   scatter-dispatch rule, 512 no row rotation of the tile map, 4096 coarse pass keeps every
//   block, 8192 large-record sweep for every batch (32/64-pixel tiles), bits 16..23 = n + 1:
//   pixel-parallel path of 16-pixel tiles for batches <= n records (n = 0 disables it; default
//   kPixelPathRecords), 16384 frames of a swap chain are treated as lone frames (ordered dispatch
//   and split tiles although they overlap), 32768 no pixel-owner sweep on 32-pixel tiles.
#include "plan.h"

using namespace crender_detail;

namespace {

// ---- tile rasterizer --------------------------------------------------------------
// Alternative block -> tile map (debug knob 8): one contiguous band of tiles per XCD.
// Measured SLOWER than the identity map on every workload (r01: T-Rex 8192^2 0.60 vs 0.42 ms):
// the covered tiles cluster, so banding piles the work onto a few XCDs.  The identity map
// deals neighbouring tiles round-robin over the XCDs and is the default.
CR_DEV int xcd_band_tile(int b, int n)
{
    const int per = n >> 3, rem = n & 7;
    const int xcd = b & 7, k = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + k;
}

// Slot of tile-local pixel (dx, dy) in the LDS key plane.  On 32-pixel tiles a row of the plane (32 keys of
// 8 bytes) is exactly one sweep of the 64 LDS banks, so lanes that work on the same columns of different
// rows — the 16 lanes of a 4x4 block, the rows of one wide box, the lanes of a run-wise sweep that sit a
// few items apart — would all meet in the same banks.  Each row's PAIRS of keys are therefore permuted by
// the row number (x ^ 2 (y mod 16)): the same columns of up to 16 consecutive rows lie in 16 different
// bank groups.
template <int TS>
CR_DEV int key_slot(int dx, int dy)
{
#ifndef CRENDER_NO_KEY_SWIZZLE
    if constexpr (TS == 32) return dy * 32 + (dx ^ ((dy << 1) & 30));
#endif
    return dy * TS + dx;
}

// Lower an LDS depth key (order-independent: the final key is the minimum over all fragments).
CR_DEV void lds_key_min(unsigned long long *slot, unsigned long long k)
{
    // No pre-read of the key: a non-returning ds_min_u64 does not stall the wavefront, whereas
    // "load, compare, then maybe atomic" puts two dependent LDS round trips on every trip's
    // critical path (T-Rex 1024^2 raster 24.2 -> 23.6 us).
    __hip_atomic_fetch_min(slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Two kernels render a 32-pixel plan's tiles, BOTH exact on every tile (the parity tests force each one
// through every scene): the general one with every path in it, and
//   owners  every batch goes to the pixel owners (frames of large triangles: bunny 4096^2, T-Rex 8192^2);
//           no key plane, no prefix sums: 19.5 KB of LDS and 7 wavefronts per SIMD instead of 6.
// Which one a frame gets is a hint about speed only (run_raster_pass: the size class the previous frames'
// tiles reported, or crender_plan_set_raster_path).  (A third kernel for frames of SMALL records — the
// run-wise sweep alone, over exact per-row spans — was built and measured in round 6 and is not here: the
// spans cost what they save, profiles/r06/ab_row_spans.txt.)
enum { kPathGeneral = 0, kPathOwners = 1 };
constexpr int kStatSlots = 16;

// One batch of (tile, triangle) work in LDS, struct-of-arrays, slot = thread index.
// A record's pixel box (clipped to the tile) is cut into work items numbered row-major:
// 4x4-pixel blocks on 32/64-pixel tiles, single pixels on 16-pixel tiles; blk_scan holds the
// wave-local exclusive prefix of the item counts.
struct WorkQueue {
    float x0[kThreads], y0[kThreads], z0[kThreads];
    float x1[kThreads], y1[kThreads], z1[kThreads];
    float x2[kThreads], y2[kThreads], z2[kThreads];
    uint32_t tri[kThreads];
    // the record's pixel box clipped to the tile, TILE-LOCAL and packed: x0 | y0 << 6 | w << 12 |
    // h << 19 (w = 0: no work).  One word instead of two: with the small-record batches' private
    // arrays below the 32-pixel kernel stays at six workgroups per CU (27 136 bytes of LDS each).
    uint32_t box[kThreads];
    uint32_t wave_blocks[kThreads / 64];
    // a batch is swept one way or the other: the two sweeps' private arrays share their memory
    union {
        struct {
            unsigned long long mask[kThreads];  // large-record batches: blocks that survive the cull
            uint32_t blk_scan[kThreads];        // exclusive prefix of the records' block counts within the wavefront
        } big;
        struct {                            // small-record batches of 32-pixel tiles: what depends on the
            float l03[kThreads], l13[kThreads], l23[kThreads];   // triangle alone and is not one operation
            float r1[kThreads], r2[kThreads], r3[kThreads];      // away — the denominators of mu.pyx:11-21 and
            uint32_t px_scan[kThreads];                          // their refined reciprocals (r1 = 0: none);
        } pre;                                                   // exclusive prefix of the records' item counts
    };
    // 32-pixel tiles count a batch both ways (blocks above, pixels in pre.px_scan) and pick the sweep after
    uint32_t wave_px[kThreads / 64];
};

// 16-pixel tiles keep a batch's records array-of-structures with everything that depends on the
// triangle alone worked out ONCE by the record's thread: the nine edge constants of mu.pyx:11-21
// and the refined reciprocals of the three denominators (raster_math.h (2)).  A sample then costs
// six 16-byte LDS reads and ~100 vector instructions instead of twelve 4-byte reads and ~140
// (T-Rex 1024^2's raster launch is bound by the vector pipes: 3.6 M instructions x 4 cycles over
// 1024 SIMDs).  96 bytes per record, 128 records per batch (12 KB: eight workgroups per CU).
struct __attribute__((aligned(16))) Rec16 {
    float x0, y0, x1, y1;
    float x2, y2, z0, z1;
    float z2; uint32_t low, box_xy, box_wh;      // low word of the record's depth keys; box_wh bit 31:
                                                 // denominators inside the division window
    float l01, l02, l11, l12;
    float l21, l22, l03, l13;
    float l23, r1, r2, r3;
};
static_assert(sizeof(Rec16) == 96, "six 16-byte pieces");
constexpr int kBatch16 = 128;
constexpr uint32_t kRecFast = 0x80000000u;

CR_DEV void put_rec16(Rec16 *dst, const TriXYZ &t, uint32_t low, uint32_t box_xy, uint32_t box_wh)
{
    const TriSetup s = make_setup(t, true);
    float4 *d = reinterpret_cast<float4 *>(dst);
    d[0] = make_float4(t.x0, t.y0, t.x1, t.y1);
    d[1] = make_float4(t.x2, t.y2, t.z0, t.z1);
    d[2] = make_float4(t.z2, __uint_as_float(low), __uint_as_float(box_xy),
                       __uint_as_float(box_wh | (s.fast ? kRecFast : 0u)));
    d[3] = make_float4(s.l01, s.l02, s.l11, s.l12);
    d[4] = make_float4(s.l21, s.l22, s.l03, s.l13);
    d[5] = make_float4(s.l23, s.r1, s.r2, s.r3);
}

// A record's sample at pixel (X, Y): same operations as fragment() — the numerators of mu.pyx:34
// from the stored edge constants, then the three correctly rounded quotients.
struct Rec16Regs {
    float4 a, b, c, d, e, f;
};
CR_DEV Rec16Regs load_rec16(const Rec16 *r)
{
    const float4 *p = reinterpret_cast<const float4 *>(r);
    return Rec16Regs{p[0], p[1], p[2], p[3], p[4], p[5]};
}
CR_DEV bool fragment16(const Rec16Regs &R, int X, int Y, unsigned long long &key)
{
    TriSetup s;
    s.x0 = R.a.x; s.y0 = R.a.y; s.x1 = R.a.z; s.y1 = R.a.w;
    s.x2 = R.b.x; s.y2 = R.b.y; s.z0 = R.b.z; s.z1 = R.b.w;
    s.z2 = R.c.x;
    s.l01 = R.d.x; s.l02 = R.d.y; s.l11 = R.d.z; s.l12 = R.d.w;
    s.l21 = R.e.x; s.l22 = R.e.y; s.l03 = R.e.z; s.l13 = R.e.w;
    s.l23 = R.f.x; s.r1 = R.f.y; s.r2 = R.f.z; s.r3 = R.f.w;
    s.rej1 = s.rej2 = s.rej3 = 0.0f;
    s.fast = (__float_as_uint(R.c.w) & kRecFast) != 0;
    float n1, n2, n3;
    numerators(s, X, Y, n1, n2, n3);
    float b1, b2, b3;
    quotients(s, n1, n2, n3, true, b1, b2, b3);
    if (b1 < 0.0f || b2 < 0.0f || b3 < 0.0f) return false;     // .pyx:215-216 (NaN passes)
    const float z = interp(s.z0, s.z1, s.z2, b1, b2, b3);
    if (z != z) return false;                                  // .pyx:220
    key = make_key(zord(z), __float_as_uint(R.c.y));
    return true;
}

// The winner's z, colour and normal with the barycentrics taken from its LDS record (edge
// constants and reciprocals are there already) and colour / normal gathered by triangle index:
// the operations of shade_and_store on the same inputs (.pyx:219, 226-242), without its gather
// of the projected vertices and its nine edge constants.
template <typename I>
CR_DEV void shade16_store(const Rec16Regs &R, const float *__restrict__ col, const float *__restrict__ nrm,
                          uint32_t tri, int X, int Y, I pix,
                          float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb, const Light &Lt)
{
    float c[9], n[9];
    load9(elem(col, (I)((I)tri * 9)), c);
    load9(elem(nrm, (I)((I)tri * 9)), n);
    TriSetup s;
    s.x0 = R.a.x; s.y0 = R.a.y; s.x1 = R.a.z; s.y1 = R.a.w;
    s.x2 = R.b.x; s.y2 = R.b.y; s.z0 = R.b.z; s.z1 = R.b.w;
    s.z2 = R.c.x;
    s.l01 = R.d.x; s.l02 = R.d.y; s.l11 = R.d.z; s.l12 = R.d.w;
    s.l21 = R.e.x; s.l22 = R.e.y; s.l03 = R.e.z; s.l13 = R.e.w;
    s.l23 = R.f.x; s.r1 = R.f.y; s.r2 = R.f.z; s.r3 = R.f.w;
    s.rej1 = s.rej2 = s.rej3 = 0.0f;
    s.fast = (__float_as_uint(R.c.w) & kRecFast) != 0;
    float n1, n2, n3, b1, b2, b3;
    numerators(s, X, Y, n1, n2, n3);
    quotients(s, n1, n2, n3, true, b1, b2, b3);
    store_fragment(interp(s.z0, s.z1, s.z2, b1, b2, b3), c, n, b1, b2, b3, Lt, pix, zb, cb, nb);
}

// The record a 16-lane group is sweeping.  T = TriXYZ (small records: the edge constants are
// hoisted by the compiler) or TriSetup (large records: with the two division shortcuts).
template <typename T>
struct Work {
    T s;
    uint32_t id;
    int bx0, by0, bx1, by1;  // clipped pixel box [bx0, bx1) x [by0, by1)
    int nbx, nblk;           // 4x4 blocks across, in total
};

CR_DEV int box_w(uint32_t wh) { return (int)(wh & 0xFFFF); }
CR_DEV int box_h(uint32_t wh) { return (int)(wh >> 16); }
// WorkQueue::box: tile-local packed box <-> (xy, wh) in frame coordinates (xy = x0 | y0 << 16, wh = w | h << 16)
CR_DEV uint32_t pack_box(uint32_t xy, uint32_t wh, int X0, int Y0)
{
    if (wh == 0) return 0u;
    return (uint32_t)((int)(xy & 0xFFFF) - X0) | ((uint32_t)((int)(xy >> 16) - Y0) << 6) | ((wh & 0x7Fu) << 12) |
           ((wh >> 16) << 19);
}
CR_DEV uint32_t packed_wh(uint32_t b) { return ((b >> 12) & 0x7Fu) | (((b >> 19) & 0x7Fu) << 16); }
// bits 26..31 of the word: how many records WITHOUT work follow this one in its wavefront's 64 slots (the
// run-wise sweep steps over them; a frame rendered alone clips every list to a quadrant of its tile, and
// three records in four have no work there)
CR_DEV uint32_t packed_skip(uint32_t b) { return b >> 26; }
CR_DEV uint32_t packed_xy(uint32_t b, int X0, int Y0)
{
    return (uint32_t)(X0 + (int)(b & 0x3Fu)) | ((uint32_t)(Y0 + (int)((b >> 6) & 0x3Fu)) << 16);
}

CR_DEV int blocks_of(uint32_t box_wh)
{
    return ((box_w(box_wh) + 3) >> 2) * ((box_h(box_wh) + 3) >> 2);
}

template <typename T>
CR_DEV Work<T> load_work(const WorkQueue &q, int r, int X0, int Y0)
{
    Work<T> w;
    w.id = q.tri[r];
    const uint32_t xy = packed_xy(q.box[r], X0, Y0), wh = packed_wh(q.box[r]);
    w.bx0 = xy & 0xFFFF;
    w.by0 = xy >> 16;
    const int bw = box_w(wh), bh = box_h(wh);
    w.bx1 = w.bx0 + bw;
    w.by1 = w.by0 + bh;
    w.nbx = (bw + 3) >> 2;
    w.nblk = w.nbx * ((bh + 3) >> 2);
    const TriXYZ t{q.x0[r], q.y0[r], q.z0[r], q.x1[r], q.y1[r], q.z1[r], q.x2[r], q.y2[r], q.z2[r]};
    if constexpr (sizeof(T) == sizeof(TriXYZ)) w.s = t;
    else w.s = make_setup(t, w.nblk >= 16);
    return w;
}

// Flattened block index -> (wavefront, slot): the record holding block p of the batch.
// (`scan` = the queue's wave-local exclusive prefix of the counts in question, `wo` the
// exclusive prefix of the wavefronts' totals)
CR_DEV int find_record(const uint32_t *scan, const uint32_t *wo, int p, uint32_t &first_block)
{
    int w = 0;
#pragma unroll
    for (int v = 1; v < kThreads / 64; ++v)
        if ((uint32_t)p >= wo[v]) w = v;
    const uint32_t pl = (uint32_t)p - wo[w];
    int lo = w * 64, n = 64;   // last slot in [lo, lo + 64) with scan <= pl
    while (n > 1) {
        const int half = n >> 1;
        if (scan[lo + half] <= pl) lo += half;
        n -= half;
    }
    first_block = pl - scan[lo];
    return lo;
}

// True if no pixel of the rectangle [xa, xb] x [ya, yb] can hold a fragment of the triangle: one
// edge is "surely outside" (raster_math.h (1)) at the corner where its numerator is largest — the
// numerators are monotone in X and in Y (every rounding step is), so every pixel of the rectangle
// then fails that edge.  A NaN fails the test (keeps the rectangle); exact, never a guess.
CR_DEV bool rect_surely_missed(const TriSetup &s, int xa, int xb, int ya, int yb)
{
    const float fxa = (float)xa, fxb = (float)xb, fya = (float)ya, fyb = (float)yb;
    auto worst = [&](float l1, float l2, float ya_, float xb_, float rej) {
        const float fy = (l1 * rej >= 0.0f) ? fyb : fya;
        const float fx = (l2 * rej >= 0.0f) ? fxa : fxb;
        return (l1 * (fy - ya_) - l2 * (fx - xb_)) * rej;
    };
    return worst(s.l01, s.l02, s.y2, s.x2, s.rej1) < -kRejTiny ||
           worst(s.l11, s.l12, s.y0, s.x0, s.rej2) < -kRejTiny ||
           worst(s.l21, s.l22, s.y1, s.x1, s.rej3) < -kRejTiny;
}

// Coarse pass of a large-record batch: one lane per dense 4x4 block; surviving blocks are
// recorded in q.big.mask.
CR_DEV void coarse_cull(WorkQueue &q, const uint32_t *wo, int total, int tid, int X0, int Y0,
                                                      bool keep_all)
{
    for (int p = tid; p < total; p += kThreads) {
        uint32_t first;
        const int r = find_record(q.big.blk_scan, wo, p, first);
        const TriSetup s = make_setup(TriXYZ{q.x0[r], q.y0[r], q.z0[r], q.x1[r], q.y1[r], q.z1[r],
                                             q.x2[r], q.y2[r], q.z2[r]}, false);
        const uint32_t xy = packed_xy(q.box[r], X0, Y0), wh = packed_wh(q.box[r]);
        const int nbx = (int)((wh & 0xFFFF) + 3) >> 2;
        const int b = (int)first;
        const int by = (int)(((float)b + 0.5f) * (1.0f / (float)nbx)), bx = b - by * nbx;
        const int xa = (int)(xy & 0xFFFF) + (bx << 2), ya = (int)(xy >> 16) + (by << 2);
        // t_k = num_k * rej_k (>= -2^-60 unless surely outside) grows with Y when l_k1 * rej_k > 0
        // and with X when l_k2 * rej_k < 0: evaluate each edge at the corner where it is largest
        const float fxa = (float)xa, fxb = (float)(xa + 3), fya = (float)ya, fyb = (float)(ya + 3);
        auto worst = [&](float l1, float l2, float ya_, float xb_, float rej) {
            const float fy = (l1 * rej >= 0.0f) ? fyb : fya;
            const float fx = (l2 * rej >= 0.0f) ? fxa : fxb;
            return (l1 * (fy - ya_) - l2 * (fx - xb_)) * rej;
        };
        const bool o1 = worst(s.l01, s.l02, s.y2, s.x2, s.rej1) < -kRejTiny;
        const bool o2 = worst(s.l11, s.l12, s.y0, s.x0, s.rej2) < -kRejTiny;
        const bool o3 = worst(s.l21, s.l22, s.y1, s.x1, s.rej3) < -kRejTiny;
        if (keep_all || !(o1 || o2 || o3)) atomicOr(&q.big.mask[r], 1ull << b);
    }
}

#ifdef CRENDER_STAMPS
// Diagnostic build only: per-workgroup phase timestamps (s_memrealtime, 100 MHz, one clock for
// the whole device — s_memtime has a base per XCD / clock domain) written to a buffer of their
// own that no kernel reads.  16 words per workgroup of the raster grid: t_start, t_ready, t_swept,
// t_end, list length, t_loads, t_queue, XCC id, tile, quadrant + 1.
__device__ unsigned long long *g_stamps = nullptr;
#define CR_STAMP(slot)                                                             \
    do {                                                                           \
        if (g_stamps && threadIdx.x == 0) g_stamps[stamp_base + (slot)] = wall_clock64(); \
    } while (0)
#else
#define CR_STAMP(slot) do { } while (0)
#endif

// Background of a tile rectangle (fused clear): z = 1e6, colour = normal = 0, winner = -1.
// Full-width rows of 16-byte aligned planes go out as float4 stores (a 16-pixel tile is 448 of
// them, two per thread, against seven dword stores per pixel); anything else pixel by pixel.
// Every address is a uniform base (the rectangle's first pixel) plus a 32-bit per-thread offset:
// the empty tiles are three quarters of a 1024^2 frame's workgroups and their instruction count
// is part of the launch's (64-bit per-thread address arithmetic tripled it).
// The background goes out write-through (sc1): a plain store allocates its line in the XCD's L2 and
// 28 MB of them per 1024^2 frame push the lists and records the covered tiles are about to read out
// of it; write-through stores cost the same and leave the L2 alone (T-Rex 1024^2: one frame alone
// 22.8 -> 21.7 us, a launch that only clears 6.8 -> 6.2 us per frame in flight).
typedef float cr_v4f __attribute__((ext_vector_type(4)));
CR_DEV void st4(float *p, const float4 &v)
{
    const cr_v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
}
template <int TS>
CR_DEV void clear_rect(float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
                       int32_t *__restrict__ win, int W, int X0, int Y0, int X1, int Y1, bool vec, int tid)
{
    const size_t p0 = (size_t)Y0 * W + X0;
    float *z0 = zb + p0, *c0 = cb + p0 * 3, *n0 = nb + p0 * 3;
    int32_t *w0 = win ? win + p0 : nullptr;
    const uint32_t t = (uint32_t)tid, uW = (uint32_t)W;
    // (the constants are made here, opaquely: hoisted to the top of the kernel they would hold
    // registers across the whole sweep)
    float z1 = 1e6f, o1 = 0.0f;
    asm volatile("" : "+v"(z1), "+v"(o1));
    const float4 zv = make_float4(z1, z1, z1, z1), ov = make_float4(o1, o1, o1, o1);
    const int4 wv = make_int4(-1, -1, -1, -1);
    if (vec && X1 - X0 == TS) {
        constexpr uint32_t ZQ = TS / 4, CQ = 3 * TS / 4;          // float4 per row: z, colour / normal
        const uint32_t rows = (uint32_t)(Y1 - Y0);
        if (TS == 16 && rows == 16) {
            // thread t: piece t (z plane for t < 64, else colour piece t - 64) and piece t + 256
            // (normal piece t, t < 192)
            const uint32_t r1 = t / CQ, off1 = r1 * uW * 3 + (t - r1 * CQ) * 4;
            if (t < 64) {
                const uint32_t off0 = (t >> 2) * uW + (t & 3) * 4;
                st4(z0 + off0, zv);
                if (w0) *reinterpret_cast<int4 *>(w0 + off0) = wv;
            } else {
                const uint32_t k = t - 64, r0 = k / CQ;
                st4(c0 + r0 * uW * 3 + (k - r0 * CQ) * 4, ov);
            }
            if (t < 192) st4(n0 + off1, ov);
            return;
        }
        for (uint32_t i = t; i < rows * ZQ; i += kThreads) {
            const uint32_t r = i / ZQ, off = r * uW + (i - r * ZQ) * 4;
            st4(z0 + off, zv);
            if (w0) *reinterpret_cast<int4 *>(w0 + off) = wv;
        }
        for (uint32_t i = t; i < rows * CQ; i += kThreads) {
            const uint32_t r = i / CQ, off = r * uW * 3 + (i - r * CQ) * 4;
            st4(c0 + off, ov);
            st4(n0 + off, ov);
        }
        return;
    }
    const uint32_t w = (uint32_t)(X1 - X0), n = w * (uint32_t)(Y1 - Y0);
    for (uint32_t p = t; p < n; p += kThreads) {
        const uint32_t dy = p / w, off = dy * uW + (p - dy * w);
        z0[off] = z1;
        c0[off * 3] = o1; c0[off * 3 + 1] = o1; c0[off * 3 + 2] = o1;
        n0[off * 3] = o1; n0[off * 3 + 1] = o1; n0[off * 3 + 2] = o1;
        if (w0) w0[off] = -1;
    }
}

// Per-frame view of the plan's tile lists, of the heavy-tile hand-off and of the dispatch-order
// hint (k_raster side).
struct TileLists {
    const uint32_t *offs;       // scan path: list offsets into `entries`; null = direct bins
    const uint32_t *count;      // direct bins: list lengths of THIS frame (never written here)
    uint32_t *count_next;       // the other parity's counters: zeroed here for the next frame
    const uint32_t *entries;    // scan path: triangle indices — or (`pairs`: a triangle order is set and the
    int pairs;                  // lists were written by k_fill_wave<true>) (position, caller's index) pairs
    const float4 *bins;         // direct bins: [ntiles][capacity] entries (BinEntry, three pieces each)
    uint32_t capacity;
    uint32_t T;                 // triangle count: list entries >= T (stale workspace) are ignored
    // The triangle arrays may be a PERMUTATION of the caller's (crender_plan_set_triangle_order:
    // tile-coherent order, so that list entries and winners gather near-streams).  Depth keys and
    // the winner plane speak the caller's indices: orig_of[position] for the key, pos_of[index]
    // back to the arrays.  Both null: the arrays are in the caller's order.
    const uint32_t *orig_of, *pos_of;
    // heavy tiles (see register_heavy): null / 0 when the launch has no helper workgroups
    uint32_t *heavy_flag, *heavy_slots, *heavy_ctr_next;
    const uint32_t *heavy_ctr;  // how many tiles THIS frame's binning pass registered (it ran in an earlier launch)
    int nhelp;                  // 3 * (helper triples of this frame) helper workgroups
    uint32_t quad_at;           // lists from here on are split in quadrants, shorter registered ones in halves
    // dispatch order (see build_order): null when the launch is not ordered
    int addr32;                  // framebuffer and attribute byte offsets fit 32 bits (see elem())
    const uint32_t *order, *hint, *hint_bad;
    uint32_t *order_next, *hint_next, *hint_bad_next;
    unsigned char *grouped_next;
    int vec_clear;              // planes 16-byte aligned and W % 4 == 0
    Light light;                // CRENDER_FUSED_GURO: illumination applied as pixels are stored
    // this frame's bin-usage record in the plan's pinned host memory (crender_plan_poll_bin_usage):
    // {frame number, hdr[0], hdr[1], hdr[4]} {large tiles, small tiles, kernel, frame number}, two 16-byte stores
    // by the launch's LAST main workgroup (in raster order a corner tile, in an ordered launch a group of empty
    // tiles: nobody's critical path)
    const uint32_t *hdr;
    uint32_t *usage;
    uint32_t usage_seq;
    // Size class of this frame's covered tiles, counted by the tiles' own workgroups (32-pixel plans): per
    // covered tile one non-returning atomic into one of kStatSlots (large, small) pairs — is the first
    // wavefront's share of the list, on average, records of 16 blocks and more (the pixel owners' kind) or
    // smaller ones (the run-wise sweep's)?  The launch's record carries the sums of the PREVIOUS launch on
    // the plan (complete by then: launches of a plan follow each other on a stream) and zeroes them; the
    // host picks the next launches' kernel by them (raster_path_hint).  A hint about speed only.
    uint32_t *stats, *stats_prev;
    uint32_t path;              // which kernel this launch is (kPath*): goes into the record
};

// One record of a tile's list: projected vertices, triangle index AS THE CALLER KNOWS IT (what depth
// keys and the winner plane speak), pixel box.  false = a stale index (beyond the frame's triangle
// count): no work.
CR_DEV bool load_record(const TileLists &L, const float *__restrict__ proj, const Geom &G, uint32_t idx,
                        uint32_t &id, TriXYZ &t, uint32_t &ebx, uint32_t &eby)
{
    if (L.pairs || L.offs) {
        uint32_t at;                 // position in the arrays
        if (L.pairs) {
            const uint2 e = reinterpret_cast<const uint2 *>(L.entries)[idx];
            at = e.x; id = e.y;
            if (at >= L.T) return false;
        } else {
            at = id = L.entries[idx];
            if (at >= L.T) return false;
            if (L.orig_of) id = L.orig_of[at];
        }
        t = load_tri(proj + (size_t)at * 9);
        int xl, xr, yt, yb;
        pixel_box(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2, G.W, G.H, xl, xr, yt, yb);
        ebx = (uint32_t)xl | ((uint32_t)xr << 16);
        eby = (uint32_t)yt | ((uint32_t)yb << 16);
        return true;
    }
    const float4 *e = L.bins + (size_t)idx * 3;          // (BinEntry: 32- and 64-pixel tiles)
    const float4 e0 = e[0], e1 = e[1], e2 = e[2];
    t = TriXYZ{e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x};
    id = __float_as_uint(e2.y);
    ebx = __float_as_uint(e2.z);
    eby = __float_as_uint(e2.w);
    if (id >= L.T) return false;
    if (L.orig_of) id = L.orig_of[id];
    return true;
}

// One covered tile's size class into the frame's statistics (see TileLists::stats): called by ONE lane, as the
// LAST memory operation of its wavefront.  (In the middle of the tile's chain — where the class is known — the
// atomic sat in front of every later load in the wavefront's in-order memory counter: a device-scope atomic on
// a word 78 workgroups share takes microseconds to come back, and the lone T-Rex 1024^2 launch took 18.4 us
// instead of 15.3; profiles/r06/ab_stats_atomics.txt.)
CR_DEV void count_tile_class(const TileLists &L, int cls)
{
    if (L.stats && cls >= 0) atomicAdd(&L.stats[(((blockIdx.x >> 3) & (kStatSlots - 1)) << 1) + (uint32_t)cls], 1u);
}
// 0: the first wavefront's share of the list is, on average, records of 16 blocks and more; 1: smaller ones
CR_DEV int tile_class_of(uint32_t incl_blocks, uint32_t nrec)
{
    const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)incl_blocks, 63);
    const uint32_t n0 = nrec < 64u ? nrec : 64u;
    return tot0 >= 16u * n0 ? 0 : 1;
}

// ---- dispatch order from the previous frame's coverage -------------------------------------
// The dispatcher starts workgroups strictly in grid order and a workgroup slot is held until its
// stores are acknowledged, so in raster order the covered tiles of T-Rex 1024^2 (the middle rows
// of the frame) started 1-3 us into the launch, behind a full chip of background tiles whose
// 28 MB of clears also doubled the latency of every load the covered tiles then issued
// (in-kernel stamps, profiles/r02).  Consecutive frames cover almost the same tiles, so each
// raster launch leaves an ORDER for the next launch on the same plan: the tiles it found covered
// first (longest lists first), one workgroup each, then the empty tiles in groups of kGroup per
// workgroup, which are cleared without a look at their lists.  The order is only a hint about speed — it is always a permutation of the tiles
// and every workgroup reads the actual list length of each tile it is handed, rasterizing it
// if it is not empty after all — so a stale order (another model, a first frame) costs time,
// never pixels.  Built by the launch's first workgroup from this frame's counters, which no
// workgroup writes; read by the next launch (ping-pong buffers).
constexpr int kGroup = 8;      // empty 16-pixel tiles cleared per workgroup of the order's last section
constexpr int group_tiles(int ts) { return ts >= 32 ? kGroup / 4 : kGroup; }   // (the same 56 KB of 32-pixel tiles)
CR_DEV void build_order(const uint32_t *__restrict__ count, int ntx, int nty,
                        uint32_t *__restrict__ order_next, unsigned char *__restrict__ grouped_next,
                        uint32_t *__restrict__ hint_next, uint32_t *scr, int group)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = ntx * nty;
    // Step 1: each tile's class into LDS, one byte per tile (coalesced loads of the counters, all
    // in flight).  Then the stable partition over contiguous chunks of the class bytes.  The
    // launch cannot end before this workgroup does: it has to stay a few microseconds (a version
    // that also looked at every empty tile's eight neighbours, to give tiles next to the model a
    // workgroup of their own, took as long as the whole launch).
    // classes: 0 heavy, 1 medium, 2 light lists (a workgroup each); 4 empty (cleared in groups)
    unsigned char *cls = reinterpret_cast<unsigned char *>(scr + 32);
    (void)nty;
    for (int i = tid; i < ntiles; i += kThreads) {
        const uint32_t c = count[i];
        cls[i] = (unsigned char)(c >= kHeavyAt ? 0 : c >= 8u ? 1 : c ? 2 : 4);
    }
    __syncthreads();
    auto cls_of = [&](int i) { return (int)cls[i]; };
    // contiguous chunk of tiles per thread: the partition is stable, so each class keeps raster
    // order (tiles that are cleared together stay neighbours in memory: scattered, the clears of
    // T-Rex 1024^2 alone took 14 us instead of 7)
    const int chunk = (ntiles + kThreads - 1) / kThreads;
    const int i0 = tid * chunk, i1 = i0 + chunk < ntiles ? i0 + chunk : ntiles;
    uint32_t n[5] = {0, 0, 0, 0, 0};
    for (int i = i0; i < i1; ++i) {
        const int k = cls_of(i);
#pragma unroll
        for (int c = 0; c < 5; ++c) n[c] += (k == c);
    }
    uint32_t incl[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        incl[c] = wave_incl_sum(n[c]);
        if (lane == 63) scr[wave * 5 + c] = incl[c];
    }
    __syncthreads();
    uint32_t off[5], tot[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const uint32_t t = scr[w * 5 + c];
            if (w < wave) before += t;
            total += t;
        }
        tot[c] = total;
        off[c] = before + incl[c] - n[c];
    }
    uint32_t basec = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) { off[c] += basec; basec += tot[c]; }
    for (int i = i0; i < i1; ++i) {
        const int k = cls_of(i);
        uint32_t pos = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (k == c) pos = off[c]++;
        order_next[pos] = (uint32_t)i;
        grouped_next[i] = k == 4;
    }
    if (tid == 0) {
        const uint32_t ncov = tot[0] + tot[1] + tot[2];
        hint_next[1] = ncov + tot[3];                       // tiles with a workgroup of their own
        hint_next[2] = (tot[4] + group - 1) / group;        // workgroups that clear `group` tiles each
        hint_next[3] = tot[0];                              // of them, the first so many hold the long lists (class 0)
        hint_next[0] = ncov ? 1u : 0u;     // an empty frame says nothing about the next one
    }
}

// ---- pixel-owner sweep (32-pixel tiles whose whole list is ONE batch of large records) -------------
// bunny 4096^2 and T-Rex 8192^2 are a few thousand triangles of thousands of pixels each: a tile
// holds a handful of records that each cover much of it.  The block sweep above computes every
// covered pixel's barycentrics twice (once for the depth key in LDS, once more in the resolve) and
// pays an LDS atomic per fragment.  Here the tile's 1024 pixels are OWNED: thread t holds four pixels
// of row t >> 3 — x = (t & 7) + 8 j, so that pixel j of a wavefront's 64 lanes is the j-th 8 x 8 block
// of its band of eight rows — with their running minimum key AND the winning fragment's
// barycentrics in registers; the wavefront walks the records in a uniform loop, passing over
// records whose triangle certainly misses its band (one word per record, worked out once by the
// record's thread: the block cull's corner test, raster_math.h (1), on box ∩ band), and the
// divisions of pixel j are skipped when none of the block's 64 pixels is a candidate — a triangle
// that touches part of a band costs the blocks it touches (with a thread's pixels side by side,
// every one of the four passes found SOME lane live: 466 k division passes per bunny frame
// instead of 333 k; raster 149 -> 133 us).  The resolve only interpolates: no LDS atomics, no
// second set of divisions.  Same device functions, same keys, same tie rule as the sweeps above:
// the planes are bit-identical.
constexpr uint32_t kOwnFast = 1u << 4;     // flags word of a record: bits 0..3 bands, 4 window, 5..10 signs
template <bool CLEAR, typename I, typename Q>
CR_DEV void owner_tile(const Q &q, const float *pre, int nrec, const float *__restrict__ col, const float *__restrict__ nrm,
                       const uint32_t *__restrict__ pos_of, const Light &Lt,
                       float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
                       int32_t *__restrict__ win, int W, int X0, int Y0, int X1, int Y1)
{
    const int tid = threadIdx.x;
    // a thread's four pixels lie 8 apart: pixel j of the wavefront's lanes is the j-th 8x8 block of its band
    constexpr int XS = 8;
    const int Xs = X0 + (tid & 7), Y = Y0 + (tid >> 3);
    const bool row_in = Y < Y1;
    const I pix0 = (I)((I)Y * (I)W + (I)Xs);
    unsigned long long best[4];
    float w1[4], w2[4], w3[4];          // the winner's barycentrics
    uint32_t slots = 0;                 // the winner's record slot, one byte per pixel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        best[j] = make_key(zord(1e6f), KEY_LOW_PRIOR);
        if (!CLEAR && row_in && Xs + XS * j < X1) best[j] = make_key(zord_prior(*elem(zb, (I)(pix0 + XS * j))), KEY_LOW_PRIOR);
        w1[j] = w2[j] = w3[j] = 0.0f;
    }
    const uint32_t my_band = 1u << (tid >> 6);
    for (int r = 0; r < nrec; ++r) {
        const float4 p0 = *reinterpret_cast<const float4 *>(pre + 8 * r);
        const uint32_t flags = __float_as_uint(p0.w);
        // the triangle cannot touch this wavefront's rows (no work, box or triangle elsewhere): uniform
        if (!(flags & my_band)) continue;
        const uint32_t wh = packed_wh(q.box[r]);
        const uint32_t xy = packed_xy(q.box[r], X0, Y0);
        const int bx0 = (int)(xy & 0xFFFF), by0 = (int)(xy >> 16);
        const int bx1 = bx0 + box_w(wh), by1 = by0 + box_h(wh);
        TriSetup st;
        {   // the record's setup: differences anew (one operation each), the rest as its thread left it
            const float4 p1 = *reinterpret_cast<const float4 *>(pre + 8 * r + 4);
            st.x0 = q.x0[r]; st.y0 = q.y0[r]; st.z0 = q.z0[r];
            st.x1 = q.x1[r]; st.y1 = q.y1[r]; st.z1 = q.z1[r];
            st.x2 = q.x2[r]; st.y2 = q.y2[r]; st.z2 = q.z2[r];
            st.l01 = st.x1 - st.x2; st.l02 = st.y1 - st.y2;
            st.l11 = st.x2 - st.x0; st.l12 = st.y2 - st.y0;
            st.l21 = st.x0 - st.x1; st.l22 = st.y0 - st.y1;
            st.l03 = p0.x; st.l13 = p0.y; st.l23 = p0.z; st.fast = (flags & kOwnFast) != 0;
            st.r1 = p1.x; st.r2 = p1.y; st.r3 = p1.z;
            auto sign_of = [](uint32_t two_bits) { return two_bits == 1u ? 1.0f : two_bits == 2u ? -1.0f : 0.0f; };
            st.rej1 = sign_of((flags >> 5) & 3u); st.rej2 = sign_of((flags >> 7) & 3u); st.rej3 = sign_of((flags >> 9) & 3u);
        }
        const uint32_t low = 0xFFFFFFFEu - q.tri[r];
        const bool rows_ok = Y >= by0 && Y < by1;
        // numerators() with the row's share of each edge worked out once for the four x-neighbours
        // (the same operations in the same order: mu.pyx:34 before the division)
        const float fy = (float)Y;
        const float ry1 = st.l01 * (fy - st.y2), ry2 = st.l11 * (fy - st.y0), ry3 = st.l21 * (fy - st.y1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (X0 + XS * j >= bx1 || X0 + XS * j + XS <= bx0) continue;     // the box misses block j (uniform)
            const int x = Xs + XS * j;
            const float fx = (float)x;
            const float n1 = ry1 - st.l02 * (fx - st.x2), n2 = ry2 - st.l12 * (fx - st.x0), n3 = ry3 - st.l22 * (fx - st.x1);
            const bool live = rows_ok && (unsigned)(x - bx0) < (unsigned)(bx1 - bx0) && !surely_outside(st, n1, n2, n3);
            if (wave_any(live)) {                                 // wavefront-uniform
                if (live) {
                    float b1, b2, b3;
                    quotients(st, n1, n2, n3, true, b1, b2, b3);
                    if (!(b1 < 0.0f || b2 < 0.0f || b3 < 0.0f)) {          // .pyx:215-216 (NaN passes)
                        const float z = interp(st.z0, st.z1, st.z2, b1, b2, b3);
                        if (z == z) {                                      // .pyx:220
                            const unsigned long long k = make_key(zord(z), low);
                            if (k < best[j]) {
                                best[j] = k;
                                w1[j] = b1; w2[j] = b2; w3[j] = b3;
                                slots = (slots & ~(0xFFu << (8 * j))) | ((uint32_t)r << (8 * j));
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- resolve: interpolate the winners and store (.pyx:219, 226-242)
    if (!row_in || Xs >= X1) return;
    float zv[4], cv[12], nv[12];
    int32_t iv[4];
    bool have[4];
    uint32_t prev = 0xFFFFFFFFu;
    float c[9], n[9], z0 = 0.f, z1 = 0.f, z2 = 0.f;
    uint32_t id = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        have[j] = (uint32_t)best[j] != KEY_LOW_PRIOR;
        zv[j] = 1e6f; iv[j] = -1;
        cv[3 * j] = cv[3 * j + 1] = cv[3 * j + 2] = 0.0f;
        nv[3 * j] = nv[3 * j + 1] = nv[3 * j + 2] = 0.0f;
        if (have[j]) {
            const uint32_t sl = (slots >> (8 * j)) & 0xFFu;
            if (sl != prev) {            // (a thread's four pixels mostly share their winner)
                prev = sl;
                id = q.tri[sl];
                z0 = q.z0[sl]; z1 = q.z1[sl]; z2 = q.z2[sl];
                const uint32_t at = pos_of ? pos_of[id] : id;
                load9(elem(col, (I)((I)at * 9)), c);
                load9(elem(nrm, (I)((I)at * 9)), n);
            }
            const float b1 = w1[j], b2 = w2[j], b3 = w3[j];
            zv[j] = interp(z0, z1, z2, b1, b2, b3);
            float c0 = interp(c[0], c[3], c[6], b1, b2, b3);
            float c1 = interp(c[1], c[4], c[7], b1, b2, b3);
            float c2 = interp(c[2], c[5], c[8], b1, b2, b3);
            const float n0 = interp(n[0], n[3], n[6], b1, b2, b3);
            const float n1 = interp(n[1], n[4], n[7], b1, b2, b3);
            const float n2 = interp(n[2], n[5], n[8], b1, b2, b3);
            if (Lt.on) {
                const float f = guro_factor(Lt, n0, n1, n2);
                c0 *= f; c1 *= f; c2 *= f;
            }
            cv[3 * j] = c0; cv[3 * j + 1] = c1; cv[3 * j + 2] = c2;
            nv[3 * j] = n0; nv[3 * j + 1] = n1; nv[3 * j + 2] = n2;
            iv[j] = (int32_t)id;
        }
    }
    float *zp = elem(zb, pix0), *cp = elem(cb, (I)(pix0 * 3)), *np_ = elem(nb, (I)(pix0 * 3));
    int32_t *wp = win ? reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix0)) : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (Xs + XS * j >= X1 || !(CLEAR || have[j])) continue;
        const int o = XS * j;
        zp[o] = zv[j];
        cp[3 * o] = cv[3 * j]; cp[3 * o + 1] = cv[3 * j + 1]; cp[3 * o + 2] = cv[3 * j + 2];
        np_[3 * o] = nv[3 * j]; np_[3 * o + 1] = nv[3 * j + 1]; np_[3 * o + 2] = nv[3 * j + 2];
        if (wp) wp[o] = iv[j];
    }
}

// ---- a tile's workgroup ---------------------------------------------------------------------------
// What every path of a tile's workgroup works with: the frame's arrays, this workgroup's rectangle
// (a whole tile, or one half / quadrant of a heavy one), its list, its LDS.  The paths below are
// functions of their own — which tile (pick_tile), the two small-record sweeps (sweep_items, sweep_runs32),
// the pixel path of 16-pixel tiles, the pixel owners (owner_path32 -> owner_tile), the culled block sweep,
// the two walks of 64-pixel tiles, the resolve — all inlined into raster_body, which queues the batches.
template <int TS>
struct Tile {
    const float *proj, *col, *nrm;
    const TileLists &L;
    float *zb, *cb, *nb;
    int32_t *win;
    const Geom &G;
    unsigned long long *key;     // LDS: the key plane (the pixel owners' per-record words instead)
    unsigned char *qraw;         // LDS: the batch queue (Rec16 records on 16-pixel tiles, else a WorkQueue)
    int dbg;                     // CRENDER_DEBUG of a development build, 0 in the product
    int X0, Y0, X1, Y1;          // the rectangle
    int rw, quad;                // its width in the key plane's terms; -1 = whole tile, 0..3 = part of a heavy one
    uint32_t beg, end;           // the tile's list
#ifdef CRENDER_STAMPS
    size_t stamp_base;
#endif
};
#ifdef CRENDER_DEV_KNOBS
#define CR_TILE_DBG(c) [[maybe_unused]] const int dbg = (c).dbg
#else
#define CR_TILE_DBG(c) [[maybe_unused]] constexpr int dbg = 0
#endif
#ifdef CRENDER_STAMPS
#define CR_TILE_STAMPS(c) [[maybe_unused]] const size_t stamp_base = (c).stamp_base
#else
#define CR_TILE_STAMPS(c) do { } while (0)
#endif
// (the names the bodies below were written with)
#define CR_TILE_LOCALS(c)                                                                                        \
    [[maybe_unused]] const float *const proj = (c).proj, *const col = (c).col, *const nrm = (c).nrm;             \
    [[maybe_unused]] const TileLists &L = (c).L;                                                                 \
    [[maybe_unused]] float *const zb = (c).zb, *const cb = (c).cb, *const nb = (c).nb;                           \
    [[maybe_unused]] int32_t *const win = (c).win;                                                               \
    [[maybe_unused]] const Geom &G = (c).G;                                                                      \
    [[maybe_unused]] unsigned long long *const key = (c).key;                                                    \
    [[maybe_unused]] unsigned char *const qraw = (c).qraw;                                                       \
    [[maybe_unused]] WorkQueue &q = *reinterpret_cast<WorkQueue *>((c).qraw);                  /* TS != 16 */    \
    [[maybe_unused]] Rec16 *const recs = reinterpret_cast<Rec16 *>((c).qraw);                  /* TS == 16 */    \
    [[maybe_unused]] uint32_t *const scan16 = reinterpret_cast<uint32_t *>((c).qraw + sizeof(Rec16) * kBatch16); \
    [[maybe_unused]] uint32_t *const wave16 = scan16 + kThreads;                                                 \
    [[maybe_unused]] const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                              \
    [[maybe_unused]] const int X0 = (c).X0, Y0 = (c).Y0, X1 = (c).X1, Y1 = (c).Y1, rw = (c).rw, quad = (c).quad; \
    [[maybe_unused]] const uint32_t beg = (c).beg, end = (c).end;                                                \
    CR_TILE_DBG(c);                                                                                              \
    CR_TILE_STAMPS(c)

// the batch: records array-of-structures on 16-pixel tiles (Rec16), else the WorkQueue
template <int TS>
constexpr size_t raster_queue_bytes()
{
    return TS == 16 ? sizeof(Rec16) * kBatch16 + sizeof(uint32_t) * (kThreads + 8) : sizeof(WorkQueue);
}
// (the pixel owners' kernel, kPathOwners, sizes its own: path_queue_bytes below)

// Which tile workgroup `b` of a raster launch takes, and which part of it; false: the workgroup is done
// (it built the dispatch order, found its helper slot empty, or cleared its group of empty tiles).
template <int TS, bool CLEAR>
CR_DEV bool pick_tile(const Tile<TS> &c, int b, int &b_out, int &tile, int &quad, bool &helper_out)
{
    const TileLists &L = c.L;
    const Geom &G = c.G;
    float *const zb = c.zb, *const cb = c.cb, *const nb = c.nb;
    int32_t *const win = c.win;
    unsigned char *const qraw = c.qraw;
    const int tid = threadIdx.x;
    CR_TILE_DBG(c);
    // ---- which tile, and which part of it --------------------------------------------------
    // grid = [order builder, if ordered][3 * hmax helpers][ntiles main workgroups, one tile each]; an ordered
    // launch reads it as [builder][covered tiles, longest lists first][helpers][groups of empty tiles]
    if (L.order_next) {
        if (b == 0) {
            build_order(L.count, G.ntx, G.nty, L.order_next, L.grouped_next, L.hint_next, reinterpret_cast<uint32_t *>(qraw), group_tiles(TS));
            return false;
        }
        b -= 1;
    }
    // An ORDERED launch with helpers starts the own workgroups of the tiles with LONG lists first (class 0 of
    // the order: 32 records or more — the launch ends when the slowest of them does), then the helpers that
    // take the other parts of the heavy tiles, then the rest of the order.  With the helpers leading the
    // grid (the unordered layout) a heavy tile's own workgroup started 2-3 us into the launch, behind up to
    // 1 536 helper slots (in-kernel stamps, profiles/r05: lone k_frame<32,true> 14.3 -> 12.9 us stamped);
    // with ALL covered tiles ahead of the helpers the parts of the heaviest tiles started behind the light ones.
    // (On 16-pixel plans the helpers stay in front: there they are few, and the one more dependent load in
    // front of their slot word cost the parts of the heaviest tiles more than the earlier start of the own
    // workgroups gained: k_raster<16,true> 14.2 -> 15.4 us.)
    // Of the 3 * hmax helper slots only the first 3 * (tiles registered) hold work — T-Rex 1024^2: 936 of
    // 1 536 — and an empty slot's workgroup still lives the 1.5 us its dependent loads take, in front of the
    // own workgroups of the short lists, which then started 3-4 us into the launch, into the flood of the empty
    // tiles' clears, and ended it (stamps, profiles/r05).  The ordered launch counts the slots in use (the
    // binning pass's registration counter) and sends the unused ones to the END of the grid, where they leave
    // without a look at their slot word: it is zero, nobody registered there.
    int lead = 0;                                               // own workgroups ahead of the helpers
    int used = L.nhelp;                                         // helper slots in front of the other own workgroups
#ifndef CRENDER_HELPERS_FIRST
    if constexpr (TS == 32) {
        const bool ordered_now = L.nhelp > 0 && L.order && L.hint[0] && !*L.hint_bad;
        if (ordered_now) {
            lead = (int)L.hint[3];
#ifndef CRENDER_ALL_HELPER_SLOTS
            const uint32_t reg = *L.heavy_ctr, hmax = (uint32_t)L.nhelp / 3u;
            used = 3 * (int)(reg < hmax ? reg : hmax);
#endif
        }
    }
#endif
    if (b >= G.ntiles + used) return false;                     // (an unused helper slot)
    const bool helper = b >= lead && b < lead + used;
    quad = -1;                   // -1 = the whole tile, 0..3 = one part of a heavy tile (half or quadrant)
    if (helper) {
        // part 1..3 of the heavy tile registered in this workgroup's slot, if any
        b -= lead;                                       // (its slot)
        const uint32_t v = L.heavy_slots[b];
        if (v == 0) return false;                        // (same word for every thread: uniform)
        tile = (int)v - 1;
        quad = 1 + b % 3;
    } else {
        const int m = b < lead ? b : b - used;
        if (m == 0 && tid == 0) {
            if (L.heavy_ctr_next) *L.heavy_ctr_next = 0;
            *L.hint_bad_next = 0;
        }
        if (m == G.ntiles - 1 && tid == 0) {
            // (the binning pass that wrote these words ran in an earlier launch of the stream; so did the
            // launch that counted into stats_prev)
            uint32_t nl = 0, ns = 0;
            if (L.stats_prev) {
                uint4 v[kStatSlots / 2];
#pragma unroll
                for (int i = 0; i < kStatSlots / 2; ++i) v[i] = reinterpret_cast<const uint4 *>(L.stats_prev)[i];
#pragma unroll
                for (int i = 0; i < kStatSlots / 2; ++i) {
                    nl += v[i].x + v[i].z; ns += v[i].y + v[i].w;
                    reinterpret_cast<uint4 *>(L.stats_prev)[i] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            // two aligned 16-byte stores; the sequence word leads the first and TRAILS the second: the
            // reader takes the record only when both are this frame's
            uint4 *rec = reinterpret_cast<uint4 *>(L.usage);
            rec[1] = make_uint4(nl, ns, L.path, L.usage_seq);
            rec[0] = make_uint4(L.usage_seq, L.hdr[0], L.hdr[1], L.hdr[4]);
        }
        if (L.order && L.hint[0] && !*L.hint_bad) {
            const int ns = (int)L.hint[1], ng = (int)L.hint[2];
            if (m < ns) {
                tile = (int)L.order[m];
            } else {
                // the order's last section: up to kGroup empty tiles per workgroup, cleared with two
                // float4 stores per thread and tile (no list to look at: the binning pass vouches
                // for their emptiness, see first_entry_of)
                if (m >= ns + ng) return false;
                constexpr int kG = group_tiles(TS);
                const int first = ns + (m - ns) * kG;
                const int ntl = G.ntiles - first < kG ? G.ntiles - first : kG;
                uint32_t tl[kG];
#pragma unroll
                for (int j = 0; j < kG; ++j) tl[j] = j < ntl ? L.order[first + j] : 0u;
#pragma unroll
                for (int j = 0; j < kG; ++j) {
                    if (j >= ntl) break;
                    const uint32_t tu = tl[j];
                    const int gy = G.ntx_magic ? (int)__umulhi(tu, G.ntx_magic) : (int)tu / G.ntx;
                    const int gx = (int)tu - gy * G.ntx;
                    const int x0 = gx * TS, y0 = G.y0 + gy * TS;
                    if (tid == 0) L.count_next[tu] = 0;
                    if (CLEAR)
                        clear_rect<TS>(zb, cb, nb, win, G.W, x0, y0, (x0 + TS < G.W) ? x0 + TS : G.W,
                                       (y0 + TS < G.y1) ? y0 + TS : G.y1, L.vec_clear != 0, tid);
                }
                return false;
            }
        } else {
            tile = (dbg & 8) ? xcd_band_tile(m, G.ntiles) : m;
            // Tile-coherent triangle order (lists of (position, index) pairs: millions of small triangles):
            // neighbouring tiles read the same records along their common border — a sixth of a list —
            // and workgroup m runs on XCD m % 8, each XCD with an L2 of its own: dealt round-robin, the
            // neighbours sit on eight different XCDs and every shared line is fetched once per XCD that
            // wants it.  Instead blocks of 16 x 16 tiles are dealt round-robin to the XCDs and each
            // block's tiles are walked one after another on its XCD (192 tiles in flight there: all
            // neighbours): 10 M small triangles 1 949 -> 1 642 MB fetched per launch, the launch itself
            // +1 % (blocks of 2 / 4 / 8 tiles: 1 738 / 1 700 / 1 667 MB, +3.5 / +2 / +2.5 %;
            // profiles/r05/ab_block_map_synth10m.txt).  Needs whole blocks, eight at a time.
            constexpr uint32_t SL = 4, SB = 1u << SL;
            const bool super_map = (L.pairs != 0) != ((dbg & (1 << 27)) != 0) && (G.ntx & (SB - 1)) == 0 &&
                                   (G.nty & (SB - 1)) == 0 && (((uint32_t)G.ntiles >> (2 * SL)) & 7u) == 0;
            if (super_map) {
                const uint32_t xcd = (uint32_t)m & 7u, j = (uint32_t)m >> 3;
                const uint32_t sidx = (j >> (2 * SL)) * 8u + xcd, local = j & (SB * SB - 1u);
                const uint32_t SX = (uint32_t)G.ntx >> SL;
                const uint32_t sy = sidx / SX, sx = sidx - sy * SX;
                tile = (int)((sy * SB + (local >> SL)) * (uint32_t)G.ntx + sx * SB + (local & (SB - 1u)));
            }
            // Large grids: scatter the dispatch order (block b -> tile b * stride mod ntiles) so
            // that a band of covered tiles is spread over the whole launch instead of arriving
            // together (T-Rex 8192^2: 0.446 -> 0.402 ms).  Small grids are faster in raster order
            // (T-Rex 1024^2: 24.7 vs 29.1 us), so the scatter starts at 32768 tiles.
            if (!super_map && (G.ntiles >= 32768) != ((dbg & 256) != 0)) {
                // (b * stride) mod ntiles, the product below 2^48: quotient from a double multiply
                // (exact product, at most one off after rounding), remainder fixed up
                const unsigned long long P = (unsigned long long)m * (unsigned)G.tile_stride;
                const unsigned long long qd = (unsigned long long)((double)P * G.inv_ntiles);
                long long r = (long long)(P - qd * (unsigned)G.ntiles);
                if (r < 0) r += G.ntiles;
                if (r >= G.ntiles) r -= G.ntiles;
                tile = (int)r;
            }
            // Workgroup b runs on XCD b % 8 and, there, on shader engine (b / 8) % 4, and the
            // dispatcher places workgroups strictly in order.  With a tile row that is a multiple
            // of 32 tiles a tile COLUMN would always meet the same (XCD, engine) pair: the pairs
            // that own the columns under the model fill up with long-lived workgroups and stall
            // the whole dispatch while a third of the chip's workgroup slots stand free.  Rotating
            // row ty by 9 * ty columns walks every pair through every column (T-Rex 1024^2 raster
            // 24.0 -> 21.6 us; the larger frames gain 0-2 %).
            if (!(dbg & 512) && !super_map) {
                const int ty = G.ntx_magic ? (int)__umulhi((uint32_t)tile, G.ntx_magic) : tile / G.ntx;
                const int t = tile - ty * G.ntx + 9 * ty;
                const int tx = G.ntx_magic ? t - (int)__umulhi((uint32_t)t, G.ntx_magic) * G.ntx : t % G.ntx;
                tile = ty * G.ntx + tx;
            }
        }
    }
    b_out = b;
    helper_out = helper;
    return true;
}

// depth keys of the rectangle: the prior buffer value (or the cleared value) per pixel
template <int TS, bool CLEAR>
CR_DEV void init_keys(const Tile<TS> &c)
{
    CR_TILE_LOCALS(c);
    const unsigned long long key_clear = make_key(zord(1e6f), KEY_LOW_PRIOR);
    for (int p = tid; p < TS * TS; p += kThreads) {
        unsigned long long k = key_clear;
        if (!CLEAR) {
            const int x = X0 + (p % TS), y = Y0 + (p / TS);
            if (x < X1 && y < Y1) k = make_key(zord_prior(zb[(size_t)y * G.W + x]), KEY_LOW_PRIOR);
        }
        key[key_slot<TS>(p % TS, p / TS)] = k;
    }
}

// Short batch on a 16-pixel tile: one PIXEL per thread, every thread walks the
// records (LDS broadcast reads), the running minimum stays in a register — no
// block scan, no record search, no LDS atomics.  A wavefront (4 rows of the tile)
// skips a record whose box misses its rows.
CR_DEV void pixel_path16(const Tile<16> &c, uint32_t left, bool first, const TriXYZ &cur_t, uint32_t key_low,
                         uint32_t box_xy, uint32_t box_wh)
{
    [[maybe_unused]] constexpr int TS = 16;
    CR_TILE_LOCALS(c);
    if (!first) __syncthreads();
    if (tid < (int)left) put_rec16(&recs[tid], cur_t, key_low, box_xy, box_wh);
    __syncthreads();
#ifdef CRENDER_STAMPS
    if (first) CR_STAMP(6);
#endif
    const int px = X0 + (tid & 15), py = Y0 + (tid >> 4);
    unsigned long long best = key[tid];
    for (uint32_t r = 0; r < left; ++r) {
        const uint32_t wh = recs[r].box_wh & ~kRecFast;
        if (wh == 0) continue;
        const uint32_t xy = recs[r].box_xy;
        const int bx0 = (int)(xy & 0xFFFF), by0 = (int)(xy >> 16);
        const bool in = px >= bx0 && px < bx0 + (int)(wh & 0xFFFF) &&
                        py >= by0 && py < by0 + (int)(wh >> 16);
        if (!wave_any(in)) continue;
        const Rec16Regs R = load_rec16(&recs[r]);
        unsigned long long k;
        if (in && fragment16(R, px, py, k) && k < best) best = k;
    }
    key[tid] = best;
}

// Per-pixel sweep: every pixel of every clipped box is one work item; thread t takes
// items t, t + 256, ...  All lanes work on a sample that lies in its box (a 4x4 block
// of a small box is mostly empty: 71 % of T-Rex 1024^2's block lanes were inside their
// box, 40-50 % on its busiest tiles, 35 % for the 10 M small triangles), and there is
// no per-group record walk.  The item's record comes from a two-level search of the
// prefix sums (most batches fit the first wavefront's 64 slots: then no wavefront
// selection and only log2 of the record count steps).
template <int TS>
CR_DEV void sweep_items(const Tile<TS> &c, const uint32_t *scan, const uint32_t *wo_, int total_, int nrec)
{
    CR_TILE_LOCALS(c);
    const int first_n = nrec <= 1 ? 1 : (nrec > 64 ? 64 : 1 << (32 - __clz(nrec - 1)));
    for (int e = tid; e < total_; e += kThreads) {
        uint32_t i;
        int r;
        if (nrec <= 64) {
            int lo = 0;
            for (int n = first_n; n > 1;) {     // last slot with scan <= e
                const int half = n >> 1;
                if (scan[lo + half] <= (uint32_t)e) lo += half;
                n -= half;
            }
            r = lo;
            i = (uint32_t)e - scan[lo];
        } else {
            r = find_record(scan, wo_, e, i);
        }
        if constexpr (TS == 16) {
            const Rec16Regs R = load_rec16(&recs[r]);
            const uint32_t xy = __float_as_uint(R.c.z);
            const int bw = box_w(__float_as_uint(R.c.w));
            // kItemPixels samples per item — x-neighbours of one box row — share the
            // item's record search, its six LDS reads and its decode (a third of a
            // sample's instructions and most of an iteration's dependent LDS round
            // trips); a box width that is no multiple wastes part of an item per row.
            const int bwn = (bw + kItemPixels - 1) / kItemPixels;
            const int dy = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)bwn));
            const int px0 = ((int)i - dy * bwn) * kItemPixels;
            const int x = (int)(xy & 0xFFFF) + px0, y = (int)(xy >> 16) + dy;
            unsigned long long *kp = &key[key_slot<TS>(x - X0, y - Y0)];
#pragma unroll
            for (int j = 0; j < kItemPixels; ++j) {
                unsigned long long k;
                if (fragment16(R, x + j, y, k) && (j == 0 || px0 + j < bw)) lds_key_min(kp + j, k);
            }
        } else {
            const uint32_t xy = packed_xy(q.box[r], X0, Y0);
            const int bw = box_w(packed_wh(q.box[r]));
            const TriXYZ t{q.x0[r], q.y0[r], q.z0[r], q.x1[r], q.y1[r], q.z1[r],
                           q.x2[r], q.y2[r], q.z2[r]};
            const uint32_t id = q.tri[r];
            // the item's samples share its record search, its twelve LDS reads, the nine
            // edge constants and the refined reciprocals (raster_math.h (2)): per sample
            // that was 150 vector instructions, a pair costs 175
            TriSetup st;
            {
                st.x0 = t.x0; st.y0 = t.y0; st.z0 = t.z0; st.x1 = t.x1; st.y1 = t.y1; st.z1 = t.z1;
                st.x2 = t.x2; st.y2 = t.y2; st.z2 = t.z2;
                st.l01 = t.x1 - t.x2; st.l02 = t.y1 - t.y2;
                st.l11 = t.x2 - t.x0; st.l12 = t.y2 - t.y0;
                st.l21 = t.x0 - t.x1; st.l22 = t.y0 - t.y1;
                st.l03 = q.pre.l03[r]; st.l13 = q.pre.l13[r]; st.l23 = q.pre.l23[r];
                st.r1 = q.pre.r1[r]; st.r2 = q.pre.r2[r]; st.r3 = q.pre.r3[r];
                st.fast = st.r1 != 0.0f;
                st.rej1 = st.rej2 = st.rej3 = 0.0f;
            }
            const int bwn = (bw + kItemPixels32 - 1) / kItemPixels32;
            // i / bwn for i < 1024, bwn <= 32: the approximate reciprocal is exact enough
            const int dy = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)bwn));
            const int px0 = ((int)i - dy * bwn) * kItemPixels32;
            const int x = (int)(xy & 0xFFFF) + px0, y = (int)(xy >> 16) + dy;
#pragma unroll
            for (int j = 0; j < kItemPixels32; ++j) {
                float n1, n2, n3;
                numerators(st, x + j, y, n1, n2, n3);
                unsigned long long k;
                if (fragment_from(st, id, n1, n2, n3, true, k) && (j == 0 || px0 + j < bw))
                    lds_key_min(&key[key_slot<TS>(x + j - X0, y - Y0)], k);
            }
        }
    }
}

// The same items — pairs of x-neighbours of the clipped boxes' rows — in RUNS: thread t takes
// items [t c, (t + 1) c) of the batch (c = ceil(total / 256)), finds the record of its first
// item by the search above ONCE and then walks: next pair of the row, next row, next record.
// Per item that is no search (6 dependent LDS reads and ~55 of ~230 vector instructions on
// batches of more than 64 records) and no division for the row; the record is re-read
// from LDS per item as before (registers: the 32-pixel kernel has none to spare), one round
// trip.  Lanes of a wavefront hold neighbouring records (consecutive LDS banks), every lane
// makes the same number of trips.
CR_DEV void sweep_runs32(const Tile<32> &c, const uint32_t *wo_, int total_)
{
    constexpr int TS = 32;
    CR_TILE_LOCALS(c);
    const int chunk = (total_ + kThreads - 1) / kThreads;
    int e = tid * chunk;
    int left = (e + chunk < total_ ? e + chunk : total_) - e;       // items of this thread's run
    if (left <= 0) return;
    uint32_t i;
    int r = find_record(q.pre.px_scan, wo_, e, i);
    int dy, px0;             // the item within its record: row of the box, first pixel of the pair
    {
        const int bwn = (box_w(packed_wh(q.box[r])) + kItemPixels32 - 1) / kItemPixels32;
        dy = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)bwn));
        px0 = ((int)i - dy * bwn) * kItemPixels32;
    }
    // ONE flat loop, its state stepped with selects: with `if (row done) { if (record done) ... }`
    // the compiler turned the walk into three nested loops (pairs of a row, rows of a record,
    // records) in which every lane waits for the wavefront's longest row and tallest box —
    // twice the time.  Records without work are stepped over (packed_skip); the first slot of
    // a wavefront's 64 may still be one: an idle trip.
    while (left > 0) {
        const uint32_t pb = q.box[r];
        const uint32_t xy = packed_xy(pb, X0, Y0), wh = packed_wh(pb);
        const int bw = box_w(wh), bh = box_h(wh);
        const TriXYZ t{q.x0[r], q.y0[r], q.z0[r], q.x1[r], q.y1[r], q.z1[r], q.x2[r], q.y2[r], q.z2[r]};
        const uint32_t id = q.tri[r];
        TriSetup st;
        {
            st.x0 = t.x0; st.y0 = t.y0; st.z0 = t.z0; st.x1 = t.x1; st.y1 = t.y1; st.z1 = t.z1;
            st.x2 = t.x2; st.y2 = t.y2; st.z2 = t.z2;
            st.l01 = t.x1 - t.x2; st.l02 = t.y1 - t.y2;
            st.l11 = t.x2 - t.x0; st.l12 = t.y2 - t.y0;
            st.l21 = t.x0 - t.x1; st.l22 = t.y0 - t.y1;
            st.l03 = q.pre.l03[r]; st.l13 = q.pre.l13[r]; st.l23 = q.pre.l23[r];
            st.r1 = q.pre.r1[r]; st.r2 = q.pre.r2[r]; st.r3 = q.pre.r3[r];
            st.fast = st.r1 != 0.0f;
            st.rej1 = st.rej2 = st.rej3 = 0.0f;
        }
        const int x = (int)(xy & 0xFFFF) + px0, y = (int)(xy >> 16) + dy;
#pragma unroll
        for (int j = 0; j < kItemPixels32; ++j) {
            float n1, n2, n3;
            numerators(st, x + j, y, n1, n2, n3);
            unsigned long long k;
            if (fragment_from(st, id, n1, n2, n3, true, k) && px0 + j < bw)
                lds_key_min(&key[key_slot<TS>(x + j - X0, y - Y0)], k);
        }
        left -= bw != 0 ? 1 : 0;
        px0 += kItemPixels32;
        const bool row_done = px0 >= bw;
        px0 = row_done ? 0 : px0;
        dy += row_done ? 1 : 0;
        const bool rec_done = dy >= bh;
        dy = rec_done ? 0 : dy;
        r += rec_done ? 1 + (int)packed_skip(pb) : 0;
    }
}

// 32-pixel tiles, the whole list ONE batch of large records: the pixels' owners take the tile (owner_tile).
template <bool CLEAR>
CR_DEV void owner_path32(const Tile<32> &c, int nrec)
{
    CR_TILE_LOCALS(c);
    // what depends on the triangle alone — the three denominators of mu.pyx:11-21 and
    // their refined reciprocals (raster_math.h (2)) — once per record, by the record's
    // thread, into the (unused) key plane: eight words per record
    // — and which of the four wavefronts' bands of eight rows the triangle can touch at all
    // (the exact rectangle test on box ∩ band, once per record instead of once per
    // record and wavefront), with the signs of the denominators and the window flag
    // in one word: a wavefront passes over a record that is not its business with
    // one LDS read
    float *pre = reinterpret_cast<float *>(key);
    if (tid < nrec) {
        // (the constants of a record with a box are in q.pre since the queue was written; one without a box
        // gets no band bit below and is never looked at)
        const uint32_t bwh = packed_wh(q.box[tid]), bxy = packed_xy(q.box[tid], X0, Y0);
        TriSetup st;
        st.x0 = q.x0[tid]; st.y0 = q.y0[tid]; st.z0 = q.z0[tid];
        st.x1 = q.x1[tid]; st.y1 = q.y1[tid]; st.z1 = q.z1[tid];
        st.x2 = q.x2[tid]; st.y2 = q.y2[tid]; st.z2 = q.z2[tid];
        st.l01 = st.x1 - st.x2; st.l02 = st.y1 - st.y2;
        st.l11 = st.x2 - st.x0; st.l12 = st.y2 - st.y0;
        st.l21 = st.x0 - st.x1; st.l22 = st.y0 - st.y1;
        st.l03 = q.pre.l03[tid]; st.l13 = q.pre.l13[tid]; st.l23 = q.pre.l23[tid];
        st.r1 = q.pre.r1[tid]; st.r2 = q.pre.r2[tid]; st.r3 = q.pre.r3[tid];
        st.fast = st.r1 != 0.0f;
        st.rej1 = rej_sign(st.l03); st.rej2 = rej_sign(st.l13); st.rej3 = rej_sign(st.l23);
        uint32_t flags = st.fast ? kOwnFast : 0u;
        flags |= (uint32_t)(st.rej1 > 0.0f ? 1 : st.rej1 < 0.0f ? 2 : 0) << 5;
        flags |= (uint32_t)(st.rej2 > 0.0f ? 1 : st.rej2 < 0.0f ? 2 : 0) << 7;
        flags |= (uint32_t)(st.rej3 > 0.0f ? 1 : st.rej3 < 0.0f ? 2 : 0) << 9;
        if (bwh != 0) {
            const int bx0 = (int)(bxy & 0xFFFF), by0 = (int)(bxy >> 16);
            const int bx1 = bx0 + box_w(bwh), by1 = by0 + box_h(bwh);
#pragma unroll
            for (int band = 0; band < 4; ++band) {
                const int ya = Y0 + 8 * band, yb = (ya + 8 < Y1) ? ya + 8 : Y1;
                if (by1 > ya && by0 < yb &&
                    !rect_surely_missed(st, bx0, bx1 - 1, by0 > ya ? by0 : ya, (by1 < yb ? by1 : yb) - 1))
                    flags |= 1u << band;
            }
        }
        float4 *o = reinterpret_cast<float4 *>(pre + 8 * tid);
        o[0] = make_float4(st.l03, st.l13, st.l23, __uint_as_float(flags));
        o[1] = make_float4(st.r1, st.r2, st.r3, 0.0f);
    }
    __syncthreads();
    if (L.addr32)
        owner_tile<CLEAR, uint32_t, WorkQueue>(q, pre, nrec, col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                               G.W, X0, Y0, X1, Y1);
    else
        owner_tile<CLEAR, size_t, WorkQueue>(q, pre, nrec, col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                             G.W, X0, Y0, X1, Y1);
    CR_STAMP(3);
}

// ---- the owners' kernel (kPathOwners): every batch of every tile goes to the pixel owners -----------------
// What the pixel owners need of a batch, and nothing else: nine coordinates, index, clipped box (11 KB) and
// the eight per-record words of owner_path32 (8 KB) — no key plane, no prefix sums, no second sweep in the
// kernel: 19.5 KB of LDS and (kernel_regs.py) 7 wavefronts per SIMD.  Exact on EVERY tile, whatever its
// records: a list of more than one batch (rare where this kernel is chosen) keeps the pixels' running
// minimum keys in registers across the batches and stores, after each batch, the pixels that batch won
// (a later batch's winner writes the pixel again: same thread, program order); the background last.
struct OwnerQueue {
    float x0[kThreads], y0[kThreads], z0[kThreads];
    float x1[kThreads], y1[kThreads], z1[kThreads];
    float x2[kThreads], y2[kThreads], z2[kThreads];
    uint32_t tri[kThreads];
    uint32_t box[kThreads];      // pack_box
};
// One batch of the owners' kernel into LDS: the thread's record (clipped box, coordinates, index) and its
// eight words — denominators, reciprocals, signs, window flag, and which of the four wavefronts' bands of
// eight rows the triangle can touch at all (owner_path32's, here straight from the registers the record
// was loaded into: no barrier between the queue and the words).
CR_DEV int owners_queue(const Tile<32> &c, uint32_t base, bool first)
{
    int tile_class = -1;
    [[maybe_unused]] constexpr int TS = 32;
    CR_TILE_LOCALS(c);
    OwnerQueue &oq = *reinterpret_cast<OwnerQueue *>(qraw);
    float *pre = reinterpret_cast<float *>(key);
    uint32_t id = 0, ebx = 0, eby = 0;
    TriXYZ t{};
    bool ok = base + tid < end;
    if (ok) ok = load_record(L, proj, G, base + tid, id, t, ebx, eby);
    uint32_t box_xy = 0, box_wh = 0;
    if (ok) {
        int xl = (int)(ebx & 0xFFFF), xr = (int)(ebx >> 16);
        int yt = (int)(eby & 0xFFFF), yb = (int)(eby >> 16);
        if (xl < X0) xl = X0;
        if (xr > X1) xr = X1;
        if (yt < Y0) yt = Y0;
        if (yb > Y1) yb = Y1;
        if (xl < xr && yt < yb) {
            box_xy = (uint32_t)xl | ((uint32_t)yt << 16);
            box_wh = (uint32_t)(xr - xl) | ((uint32_t)(yb - yt) << 16);
        }
    }
    if (first && __builtin_amdgcn_readfirstlane(wave) == 0)    // the tile's size class (TileLists::stats), reported at the tile's end
        tile_class = tile_class_of(wave_incl_sum((uint32_t)blocks_of(box_wh)), end - beg);
    if (!first) __syncthreads();    // the previous batch's readers are done with the queue
    oq.x0[tid] = t.x0; oq.y0[tid] = t.y0; oq.z0[tid] = t.z0;
    oq.x1[tid] = t.x1; oq.y1[tid] = t.y1; oq.z1[tid] = t.z1;
    oq.x2[tid] = t.x2; oq.y2[tid] = t.y2; oq.z2[tid] = t.z2;
    oq.tri[tid] = id;
    oq.box[tid] = pack_box(box_xy, box_wh, X0, Y0);
    uint32_t flags = 0;
    const TriSetup st = make_setup(t, true);
    if (box_wh != 0) {
        flags = st.fast ? kOwnFast : 0u;
        flags |= (uint32_t)(st.rej1 > 0.0f ? 1 : st.rej1 < 0.0f ? 2 : 0) << 5;
        flags |= (uint32_t)(st.rej2 > 0.0f ? 1 : st.rej2 < 0.0f ? 2 : 0) << 7;
        flags |= (uint32_t)(st.rej3 > 0.0f ? 1 : st.rej3 < 0.0f ? 2 : 0) << 9;
        const int bx0 = (int)(box_xy & 0xFFFF), by0 = (int)(box_xy >> 16);
        const int bx1 = bx0 + box_w(box_wh), by1 = by0 + box_h(box_wh);
#pragma unroll
        for (int band = 0; band < 4; ++band) {
            const int ya = Y0 + 8 * band, yb = (ya + 8 < Y1) ? ya + 8 : Y1;
            if (by1 > ya && by0 < yb &&
                !rect_surely_missed(st, bx0, bx1 - 1, by0 > ya ? by0 : ya, (by1 < yb ? by1 : yb) - 1))
                flags |= 1u << band;
        }
    }
    float4 *o = reinterpret_cast<float4 *>(pre + 8 * tid);
    o[0] = make_float4(st.l03, st.l13, st.l23, __uint_as_float(flags));
    o[1] = make_float4(st.fast ? st.r1 : 0.0f, st.r2, st.r3, 0.0f);
    __syncthreads();
    return tile_class;
}

// A list of more than one batch (rare where this kernel is chosen).  The one-batch path's register budget is
// spent on a thread's FOUR pixels; carried across batches as well, the same state spilled (and a kernel with
// scratch is slower for every workgroup, whether it takes the spilling path or not: the general kernel lost
// 12 % of the swap chain's frames to 48 bytes of it, profiles/r06/ab_compaction.txt).  So a long list is walked
// FOUR times, once per pixel of the thread, with that pixel's running minimum key in registers across the
// batches; each batch stores the pixel if it won it (a later batch's winner writes it again: same thread,
// program order), the background last.  Four times the record loads, on tiles that are rare here by choice.
template <bool CLEAR, typename I>
CR_DEV void owners_batches(const Tile<32> &c)
{
    [[maybe_unused]] constexpr int TS = 32;
    CR_TILE_LOCALS(c);
    OwnerQueue &oq = *reinterpret_cast<OwnerQueue *>(qraw);
    float *pre = reinterpret_cast<float *>(key);
    constexpr int XS = 8;
    const int Y = Y0 + (tid >> 3);
    const uint32_t my_band = 1u << (tid >> 6);
    int tile_class = -1;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const int x = X0 + (tid & 7) + XS * j;
        const bool mine_in = Y < Y1 && x < X1;
        const I pix = (I)((I)Y * (I)G.W + (I)x);
        unsigned long long best = make_key(zord(1e6f), KEY_LOW_PRIOR);
        if (!CLEAR && mine_in) best = make_key(zord_prior(*elem(zb, pix)), KEY_LOW_PRIOR);
        const float fx = (float)x, fy = (float)Y;
#pragma unroll 1
        for (uint32_t base = beg; base < end; base += kThreads) {
            const int cls = owners_queue(c, base, j == 0 && base == beg);
            if (j == 0 && base == beg) tile_class = cls;
            const int nrec = (int)((end - base) < (uint32_t)kThreads ? (end - base) : (uint32_t)kThreads);
            float w1 = 0.0f, w2 = 0.0f, w3 = 0.0f;
            int slot = -1;                  // the record of this batch that wins the pixel, if any
            for (int r = 0; r < nrec; ++r) {
                const float4 p0 = *reinterpret_cast<const float4 *>(pre + 8 * r);
                const uint32_t flags = __float_as_uint(p0.w);
                if (!(flags & my_band)) continue;           // (uniform over the wavefront)
                const uint32_t pb = oq.box[r];
                const uint32_t wh = packed_wh(pb), xy = packed_xy(pb, X0, Y0);
                const int bx0 = (int)(xy & 0xFFFF), by0 = (int)(xy >> 16);
                const int bx1 = bx0 + box_w(wh), by1 = by0 + box_h(wh);
                if (X0 + XS * j >= bx1 || X0 + XS * j + XS <= bx0) continue;     // the box misses block j (uniform)
                TriSetup st;
                {
                    const float4 p1 = *reinterpret_cast<const float4 *>(pre + 8 * r + 4);
                    st.x0 = oq.x0[r]; st.y0 = oq.y0[r]; st.z0 = oq.z0[r];
                    st.x1 = oq.x1[r]; st.y1 = oq.y1[r]; st.z1 = oq.z1[r];
                    st.x2 = oq.x2[r]; st.y2 = oq.y2[r]; st.z2 = oq.z2[r];
                    st.l01 = st.x1 - st.x2; st.l02 = st.y1 - st.y2;
                    st.l11 = st.x2 - st.x0; st.l12 = st.y2 - st.y0;
                    st.l21 = st.x0 - st.x1; st.l22 = st.y0 - st.y1;
                    st.l03 = p0.x; st.l13 = p0.y; st.l23 = p0.z; st.fast = (flags & kOwnFast) != 0;
                    st.r1 = p1.x; st.r2 = p1.y; st.r3 = p1.z;
                    auto sign_of = [](uint32_t two_bits) { return two_bits == 1u ? 1.0f : two_bits == 2u ? -1.0f : 0.0f; };
                    st.rej1 = sign_of((flags >> 5) & 3u); st.rej2 = sign_of((flags >> 7) & 3u); st.rej3 = sign_of((flags >> 9) & 3u);
                }
                // numerators() (mu.pyx:34 before the division), the operations in the reference's order
                const float n1 = st.l01 * (fy - st.y2) - st.l02 * (fx - st.x2);
                const float n2 = st.l11 * (fy - st.y0) - st.l12 * (fx - st.x0);
                const float n3 = st.l21 * (fy - st.y1) - st.l22 * (fx - st.x1);
                const bool live = Y >= by0 && Y < by1 && (unsigned)(x - bx0) < (unsigned)(bx1 - bx0) &&
                                  !surely_outside(st, n1, n2, n3);
                if (!wave_any(live)) continue;
                if (live) {
                    float b1, b2, b3;
                    quotients(st, n1, n2, n3, true, b1, b2, b3);
                    if (!(b1 < 0.0f || b2 < 0.0f || b3 < 0.0f)) {          // .pyx:215-216 (NaN passes)
                        const float z = interp(st.z0, st.z1, st.z2, b1, b2, b3);
                        if (z == z) {                                      // .pyx:220
                            const unsigned long long k = make_key(zord(z), 0xFFFFFFFEu - oq.tri[r]);
                            if (k < best) {
                                best = k;
                                w1 = b1; w2 = b2; w3 = b3;
                                slot = r;
                            }
                        }
                    }
                }
            }
            // ---- the pixel, if this batch won it: interpolate and store (.pyx:219, 226-242) ----------------
            if (mine_in && slot >= 0) {
                const uint32_t tid_w = oq.tri[slot];
                const uint32_t at = L.pos_of ? L.pos_of[tid_w] : tid_w;
                float cc[9], nn[9];
                load9(elem(col, (I)((I)at * 9)), cc);
                load9(elem(nrm, (I)((I)at * 9)), nn);
                const float zv = interp(oq.z0[slot], oq.z1[slot], oq.z2[slot], w1, w2, w3);
                float c0 = interp(cc[0], cc[3], cc[6], w1, w2, w3);
                float c1 = interp(cc[1], cc[4], cc[7], w1, w2, w3);
                float c2 = interp(cc[2], cc[5], cc[8], w1, w2, w3);
                const float n0 = interp(nn[0], nn[3], nn[6], w1, w2, w3);
                const float n1 = interp(nn[1], nn[4], nn[7], w1, w2, w3);
                const float n2 = interp(nn[2], nn[5], nn[8], w1, w2, w3);
                if (L.light.on) {
                    const float f = guro_factor(L.light, n0, n1, n2);
                    c0 *= f; c1 *= f; c2 *= f;
                }
                *elem(zb, pix) = zv;
                float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
                cp[0] = c0; cp[1] = c1; cp[2] = c2;
                np_[0] = n0; np_[1] = n1; np_[2] = n2;
                if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = (int32_t)tid_w;
            }
        }
        // ---- background: a pixel no batch won (fused clear) --------------------------------------------
        if (CLEAR && mine_in && (uint32_t)best == KEY_LOW_PRIOR) {
            *elem(zb, pix) = 1e6f;
            float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
            cp[0] = 0.0f; cp[1] = 0.0f; cp[2] = 0.0f;
            np_[0] = 0.0f; np_[1] = 0.0f; np_[2] = 0.0f;
            if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = -1;
        }
    }
    if (tid == 0) count_tile_class(L, tile_class);
    CR_STAMP(3);
}

// 64-pixel tiles, small records: each of the 16 lane groups takes one contiguous run of blocks,
// so a record is set up by (almost) one group only; tight loop, plain division.
CR_DEV void walk64_small(const Tile<64> &c, const uint32_t *wo, int total)
{
    constexpr int TS = 64;
    CR_TILE_LOCALS(c);
    const int l = tid & 15, lx = l & 3, ly = l >> 2;
    const int chunk = (total + 15) >> 4;
    int p = (tid >> 4) * chunk;
    const int pend = (p + chunk < total) ? (p + chunk) : total;
    if (p < pend) {
        uint32_t first;
        int r = find_record(q.big.blk_scan, wo, p, first);
        Work<TriXYZ> wk = load_work<TriXYZ>(q, r, X0, Y0);
        int b = (int)first;
        int by = (int)(((float)b + 0.5f) * (1.0f / (float)wk.nbx)), bx = b - by * wk.nbx;
        for (;;) {
            const int x = wk.bx0 + (bx << 2) + lx, y = wk.by0 + (by << 2) + ly;
            unsigned long long k;
            if (x < wk.bx1 && y < wk.by1 && fragment(wk.s, wk.id, x, y, k))
                lds_key_min(&key[key_slot<TS>(x - X0, y - Y0)], k);
            if (++p >= pend) break;
            if (++b < wk.nblk) {
                if (++bx == wk.nbx) { bx = 0; ++by; }
            } else {
                // p < pend guarantees a later record with blocks
                do { wk = load_work<TriXYZ>(q, ++r, X0, Y0); } while (wk.nblk == 0);
                b = bx = by = 0;
            }
        }
    }
}

// Large records (>= 16 blocks on average).  A triangle fills at most half of its
// pixel box, so first a coarse pass (one LANE per 4x4 block) discards blocks that
// lie entirely outside one edge; the survivors are then swept (one 16-lane GROUP
// per block): each wavefront takes a contiguous quarter of them and its four
// groups consecutive survivors, so the four blocks a wavefront works on at a
// time are neighbours.  In the sweep, lanes whose sign test is certainly negative
// are dead before any division (skipped wave-wide when nobody is live,
// raster_math.h (1)); the divisions that remain use the hoisted reciprocal (2).
template <int TS>
CR_DEV void sweep_blocks_culled(const Tile<TS> &c, uint32_t (&wo)[kThreads / 64 + 1], int total, uint32_t blk_excl)
{
    static_assert(TS <= 32, "a record has at most 64 blocks: one mask word");
    CR_TILE_LOCALS(c);
    const int l = tid & 15, lx = l & 3, ly = l >> 2;
    [[maybe_unused]] const bool allow_rej = !(dbg & 32), allow_fast = !(dbg & 64);
    q.big.mask[tid] = 0;
    if constexpr (TS == 32) q.big.blk_scan[tid] = blk_excl;
    __syncthreads();
    // ---- coarse pass.  Exact: each numerator is monotone in X and in Y (every
    // rounding step is), so its extreme over the block sits on a corner; a block
    // goes only if all four corners are "surely outside" the SAME edge, which is
    // then true of every pixel in it (raster_math.h (1)).
    coarse_cull(q, wo, total, tid, X0, Y0, (dbg & 4096) != 0);
    __syncthreads();
    // survivors per record -> the same two-level prefix as the block counts
    const uint32_t mine = (uint32_t)__popcll(q.big.mask[tid]);
    const uint32_t inc = wave_incl_sum(mine);
    q.big.blk_scan[tid] = inc - mine;
    if (lane == 63) q.wave_blocks[wave] = inc;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) wo[w + 1] = wo[w] + q.wave_blocks[w];
    const int work = (int)wo[kThreads / 64];       // surviving blocks
    // the survivors are almost all partly or fully covered, so wave-wide
    // rejection would rarely fire: plain contiguous runs, one per lane group
    const int chunk = (work + 15) >> 4;
    int p = (tid >> 4) * chunk;
    const int pend = (p + chunk < work) ? (p + chunk) : work;
    if (p < pend) {
        uint32_t first;
        int r = find_record(q.big.blk_scan, wo, p, first);
        Work<TriSetup> wk = load_work<TriSetup>(q, r, X0, Y0);
        float inv_nbx = 1.0f / (float)wk.nbx;
        // the record's survivor mask with everything before the current block cleared
        unsigned long long m = q.big.mask[r];
        for (uint32_t i = 0; i < first; ++i) m &= m - 1;
        for (;;) {
            const int b = __ffsll((long long)m) - 1;
            const int by = (int)(((float)b + 0.5f) * inv_nbx), bx = b - by * wk.nbx;
            const int x = wk.bx0 + (bx << 2) + lx, y = wk.by0 + (by << 2) + ly;
            float n1, n2, n3;
            numerators(wk.s, x, y, n1, n2, n3);
            unsigned long long k;
            if (x < wk.bx1 && y < wk.by1 && fragment_from(wk.s, wk.id, n1, n2, n3, allow_fast, k))
                lds_key_min(&key[key_slot<TS>(x - X0, y - Y0)], k);
            if (++p >= pend) break;   // (p < pend guarantees another survivor)
            m &= m - 1;
            if (m == 0) {
                do { m = q.big.mask[++r]; } while (m == 0);
                wk = load_work<TriSetup>(q, r, X0, Y0);
                inv_nbx = 1.0f / (float)wk.nbx;
            }
        }
    }
}

// 64-pixel tiles (up to 256 blocks per record), large records: no cull, dense walk
CR_DEV void walk64_dense(const Tile<64> &c, const uint32_t *wo, int total)
{
    constexpr int TS = 64;
    CR_TILE_LOCALS(c);
    const int l = tid & 15, lx = l & 3, ly = l >> 2;
    const bool allow_rej = !(dbg & 32), allow_fast = !(dbg & 64);
    const int wchunk = (total + kThreads / 64 - 1) / (kThreads / 64);
    int p = wave * wchunk + ((tid >> 4) & 3);
    const int pend = ((wave + 1) * wchunk < total) ? ((wave + 1) * wchunk) : total;
    if (p < pend) {
        uint32_t first;
        int r = find_record(q.big.blk_scan, wo, p, first);
        Work<TriSetup> wk = load_work<TriSetup>(q, r, X0, Y0);
        int b = (int)first;
        float inv_nbx = 1.0f / (float)wk.nbx;
        for (;;) {
            const int by = (int)(((float)b + 0.5f) * inv_nbx), bx = b - by * wk.nbx;
            const int x = wk.bx0 + (bx << 2) + lx, y = wk.by0 + (by << 2) + ly;
            float n1, n2, n3;
            numerators(wk.s, x, y, n1, n2, n3);
            const bool live = x < wk.bx1 && y < wk.by1 &&
                              !(allow_rej && surely_outside(wk.s, n1, n2, n3));
            if (wave_any(live)) {   // wavefront-uniform
                unsigned long long k;
                if (live && fragment_from(wk.s, wk.id, n1, n2, n3, allow_fast, k))
                    lds_key_min(&key[key_slot<TS>(x - X0, y - Y0)], k);
            }
            p += 4;
            if (p >= pend) break;
            b += 4;
            if (b >= wk.nblk) {
                do {
                    b -= wk.nblk;
                    wk = load_work<TriSetup>(q, ++r, X0, Y0);
                } while (b >= wk.nblk);
                inv_nbx = 1.0f / (float)wk.nbx;
            }
        }
    }
}

template <int TS, bool CLEAR, size_t QBYTES = raster_queue_bytes<TS>()>
CR_DEV void resolve_tile(const Tile<TS> &c, bool slotted)
{
    constexpr int kBatch = TS == 16 ? kBatch16 : kThreads;
    CR_TILE_LOCALS(c);
    // resolve: every pixel of the rectangle is written at most once (exactly once if CLEAR).
    // A part's pixels are taken by the first wavefronts in rows of its own width.
    const int npx = quad >= 0 ? rw * (TS / 2) : TS * TS;
    // Tile-coherent triangle order: a winner is known by the caller's index, its record sits at
    // pos_of[index].  That look-up is a 4-byte needle out of a table of T words, one 128-byte L2
    // request per distinct winner of a wavefront's pixels — on the 10 M-triangle frame 676 MB of the
    // 3.06 GB the launch fetched (profiles/r04/fetch_breakdown_synth10m.txt).  Every winner is a record of THIS tile's list,
    // whose entries (positions, a contiguous run) and their original indices (orig_of, near-contiguous)
    // the sweep has just read: they go into a hash table in the batch queue's LDS, free now, and the
    // pixels look their winners up there.  Lists too long for the table keep the global look-up.
    constexpr uint32_t kHashSlots = (QBYTES / sizeof(uint2)) >= 2048 ? 2048u : 1024u;
    static_assert(kHashSlots * sizeof(uint2) <= QBYTES, "the table takes the batch queue's place");
    uint2 *hash_tab = reinterpret_cast<uint2 *>(qraw);
    const bool hashed = L.pos_of && L.pairs && end - beg <= kHashSlots / 2 && !(dbg & (1 << 21));
    auto hash_of = [](uint32_t orig) { return (orig * 2654435761u) >> (kHashSlots == 2048u ? 21 : 22); };
    if (hashed) {
        for (uint32_t i = (uint32_t)tid; i < kHashSlots; i += kThreads) hash_tab[i] = make_uint2(0u, 0u);
        __syncthreads();
        for (uint32_t idx = beg + (uint32_t)tid; idx < end; idx += kThreads) {
            const uint2 e = reinterpret_cast<const uint2 *>(L.entries)[idx];
            const uint32_t at = e.x;
            if (at >= L.T) continue;
            const uint32_t tag = e.y + 1u;                               // (0 = empty slot)
            for (uint32_t h = hash_of(tag - 1u);; h = (h + 1u) & (kHashSlots - 1u)) {
                const uint32_t was = atomicCAS(&hash_tab[h].x, 0u, tag);
                if (was == 0u || was == tag) { hash_tab[h].y = at; break; }
            }
        }
        __syncthreads();
    }
    auto position_of = [&](uint32_t id) -> uint32_t {
        if (!L.pos_of) return id;
        if (hashed) {
            uint32_t h = hash_of(id);
            for (uint32_t probes = 0; probes < kHashSlots; ++probes, h = (h + 1u) & (kHashSlots - 1u)) {
                const uint2 e = hash_tab[h];
                if (e.x == id + 1u) return e.y;
                if (e.x == 0u) break;
            }
        }
        return L.pos_of[id];
    };
    auto resolve = [&](auto index_tag) {
    using I = decltype(index_tag);
    for (int p0 = tid; p0 < npx; p0 += kThreads) {
        const int dy = rw == TS ? p0 / TS : p0 / (TS / 2), dx = p0 - dy * rw;
        const int x = X0 + dx, y = Y0 + dy;
        if (x >= X1 || y >= Y1) continue;
        const I pix = (I)((I)y * (I)G.W + (I)x);
        const uint32_t low = (uint32_t)key[key_slot<TS>(dx, dy)];
        if (low == KEY_LOW_PRIOR) {
            if (CLEAR) {
                *elem(zb, pix) = 1e6f;
                float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
                cp[0] = 0.0f; cp[1] = 0.0f; cp[2] = 0.0f;
                np_[0] = 0.0f; np_[1] = 0.0f; np_[2] = 0.0f;
                if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = -1;
            }
            continue;
        }
        uint32_t id = 0xFFFFFFFEu - low;
        if (slotted) {
            id = 0xFFFFu - (low >> 16);
            if (((low >> 8) & 0xFFu) == (((end - beg - 1) / kBatch) & 0xFFu) && !(dbg & 2)) {
                // the winner's record is still in LDS (it came with the last batch)
                shade16_store(load_rec16(&recs[low & 0xFFu]), col, nrm, L.pos_of ? L.pos_of[id] : id, x, y, pix,
                              zb, cb, nb, L.light);
                if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = (int32_t)id;
                continue;
            }
        }
        if (dbg & 2) {   // ablation: no shading (development build)
            *elem(zb, pix) = (float)id;
            float *cp = elem(cb, (I)(pix * 3)), *np_ = elem(nb, (I)(pix * 3));
            cp[0] = 1.0f; cp[1] = 1.0f; cp[2] = 1.0f;
            np_[0] = 1.0f; np_[1] = 1.0f; np_[2] = 1.0f;
            continue;
        }
        shade_and_store(proj, col, nrm, position_of(id), x, y, pix, zb, cb, nb, L.light, (dbg >> 28) & 3);
        if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = (int32_t)id;
    }
    };
    if (L.addr32) resolve(uint32_t{}); else resolve(size_t{});
}

// Workgroup `b` of a raster launch (the kernels below hand in their LDS: k_frame runs binning
// wavefronts of another frame in the same launch): picks its tile, queues the tile's list batch by
// batch and hands each batch to the sweep that suits it, then resolves the pixels.
template <int TS, bool CLEAR, int PATH = kPathGeneral>
CR_DEV void raster_body(const float *__restrict__ proj, const float *__restrict__ col,
                        const float *__restrict__ nrm, const TileLists &L,
                        float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
                        int32_t *__restrict__ win, const Geom &G, int dbg_arg, int b,
                        unsigned long long *key, unsigned char *qraw)
{
    Tile<TS> c{proj, col, nrm, L, zb, cb, nb, win, G, key, qraw, 0, 0, 0, 0, 0, TS, -1, 0u, 0u};
#ifdef CRENDER_DEV_KNOBS
    c.dbg = dbg_arg;
#else
    (void)dbg_arg;
#endif
#ifdef CRENDER_STAMPS
    // frames of a swap chain stamp into a region of their slot (bits 24..26 of dbg_arg), 8192 workgroups each
    c.stamp_base = ((size_t)((dbg_arg >> 24) & 7) * 8192 + blockIdx.x) * 16;
#endif
    CR_TILE_DBG(c);
    CR_TILE_STAMPS(c);
    [[maybe_unused]] WorkQueue &q = *reinterpret_cast<WorkQueue *>(qraw);                 // (TS != 16 only)
    [[maybe_unused]] Rec16 *recs = reinterpret_cast<Rec16 *>(qraw);                       // (TS == 16 only)
    [[maybe_unused]] uint32_t *scan16 = reinterpret_cast<uint32_t *>(qraw + sizeof(Rec16) * kBatch16);
    [[maybe_unused]] uint32_t *wave16 = scan16 + kThreads;
    constexpr int kBatch = TS == 16 ? kBatch16 : kThreads;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- which tile, and which part of it (grid = [order builder, if ordered][3 * hmax helpers][ntiles
    // main workgroups, one tile each])
    int tile, quad;
    bool helper;
    if (!pick_tile<TS, CLEAR>(c, b, b, tile, quad, helper)) return;
    const int ty = G.ntx_magic ? (int)__umulhi((uint32_t)tile, G.ntx_magic) : tile / G.ntx;
    const int tx = tile - ty * G.ntx;
    int X0 = tx * TS, Y0 = G.y0 + ty * TS;
    int X1 = (X0 + TS < G.W) ? (X0 + TS) : G.W;
    int Y1 = (Y0 + TS < G.y1) ? (Y0 + TS) : G.y1;

    CR_STAMP(0);
#ifdef CRENDER_STAMPS
    if (g_stamps && threadIdx.x == 0) {
        g_stamps[stamp_base + 7] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
        g_stamps[stamp_base + 8] = (unsigned long long)tile;
        g_stamps[stamp_base + 10] = __builtin_amdgcn_s_memtime();
    }
#endif
    // the tile's triangle list: a run of the scanned index array, or (direct bins, offs == null)
    // a fixed-capacity slab of entries whose fill count k_setup_wave left in count[tile]
    uint32_t beg, end;
    if (L.offs) {
        beg = L.offs[tile];
        end = L.offs[tile + 1];
        if (end > L.capacity) end = L.capacity;
        if (beg > end) beg = end;
    } else {
        const uint32_t n = L.count[tile];
        beg = (uint32_t)tile * L.capacity;
        end = beg + (n < L.capacity ? n : L.capacity);
    }
    if (!helper) {
        if (L.heavy_flag && L.heavy_flag[tile]) quad = 0;
        // the other parity's counter of this tile: zero for the next frame
        if (tid == 0) L.count_next[tile] = 0;
    }
    // 32-pixel tiles of a small frame rendered alone (the launches with helper workgroups), fused clear: the key
    // plane's start value is a constant — written while the list length is still on its way, one barrier off
    // a workgroup's chain (lone k_frame<32,true> 15.9 -> 15.6 us).  Should the tile turn out to be the pixel
    // owners', they take the plane for their per-record words behind the queue's barrier.  Not on the large
    // frames: there most tiles are empty or the owners' and never need a key plane (bunny 4096^2 +3 % with it).
    // (Composite frames start from the depth buffer: see the batch loop.)
    const bool keys_early = PATH != kPathOwners && TS == 32 && CLEAR && L.nhelp > 0;
    if (keys_early) init_keys<TS, CLEAR>(c);
    int rw = TS;                 // width of this workgroup's rectangle in the key plane's terms
    if (quad >= 0) {
        constexpr int HS = TS / 2;
        if (end - beg >= L.quad_at) {           // four quadrants
            X0 += (quad & 1) * HS; Y0 += (quad >> 1) * HS;
            if (X1 > X0 + HS) X1 = X0 + HS;
            rw = HS;
        } else {                                // two halves; parts 2 and 3 have nothing to do
            if (quad >= 2) {
                if (tid == 0) L.heavy_slots[b] = 0;
                return;
            }
            Y0 += quad * HS;
        }
        if (Y1 > Y0 + HS) Y1 = Y0 + HS;
        if (X1 < X0) X1 = X0;
        if (Y1 < Y0) Y1 = Y0;
    }
    if (dbg & 1) end = beg;   // ablation: no coverage work (development build)

    const bool work = beg != end && X0 < X1 && Y0 < Y1;     // (uniform over the workgroup)
    if (!work) {
        // nothing to rasterize here: the rectangle keeps its content, or (fused clear) becomes
        // background — no key plane, no barriers
        if (CLEAR && X0 < X1 && Y0 < Y1) clear_rect<TS>(zb, cb, nb, win, G.W, X0, Y0, X1, Y1, L.vec_clear && quad < 0, tid);
        CR_STAMP(3);
    } else if constexpr (PATH == kPathOwners) {
        c.X0 = X0; c.Y0 = Y0; c.X1 = X1; c.Y1 = Y1; c.rw = rw; c.quad = quad; c.beg = beg; c.end = end;
        CR_STAMP(1);
#ifdef CRENDER_STAMPS
        if (g_stamps && tid == 0) {
            g_stamps[stamp_base + 4] = end - beg;
            g_stamps[stamp_base + 9] = (unsigned long long)(quad + 1);
        }
#endif
        if (end - beg <= (uint32_t)kThreads) {
            const int tile_class = owners_queue(c, beg, true);
            CR_STAMP(6);
            CR_STAMP(5);
            CR_STAMP(2);     // (diagnostic builds: "swept" = the queue is built; the record loop and the resolve are one function)
            const OwnerQueue &oq = *reinterpret_cast<const OwnerQueue *>(qraw);
            const float *pre = reinterpret_cast<const float *>(key);
            if (L.addr32)
                owner_tile<CLEAR, uint32_t, OwnerQueue>(oq, pre, (int)(end - beg), col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                                        G.W, X0, Y0, X1, Y1);
            else
                owner_tile<CLEAR, size_t, OwnerQueue>(oq, pre, (int)(end - beg), col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                                      G.W, X0, Y0, X1, Y1);
            if (tid == 0) count_tile_class(L, tile_class);
            CR_STAMP(3);
        } else {
            owners_batches<CLEAR, size_t>(c);
        }
    } else {
    // 16-pixel tiles with direct bins (at most 65536 triangles): a depth key's low word carries
    // the triangle index in its high half as usual and, in its low half, where the record sits
    // in LDS — batch and slot — so that the resolve takes the winner's edge constants from there
    const bool slotted = TS == 16 && !L.offs && !L.pairs;
    // first batch of the tile's list straight into registers
    uint32_t cur_id = 0, cur_bx = 0, cur_by = 0;
    TriXYZ cur_t{};
    bool cur_ok = tid < kBatch && beg + tid < end;
    if (cur_ok) cur_ok = load_record(L, proj, G, beg + tid, cur_id, cur_t, cur_bx, cur_by);

    c.X0 = X0; c.Y0 = Y0; c.X1 = X1; c.Y1 = Y1; c.rw = rw; c.quad = quad; c.beg = beg; c.end = end;
    // (32-pixel tiles: the keys once it is known that the tile is not the pixel owners', see below)
    if constexpr (TS != 32) init_keys<TS, CLEAR>(c);
    CR_STAMP(1);
#ifdef CRENDER_STAMPS
    if (g_stamps && tid == 0) {
        g_stamps[stamp_base + 4] = end - beg;
        g_stamps[stamp_base + 9] = (unsigned long long)(quad + 1);
    }
#endif

    [[maybe_unused]] int tile_class = -1;        // (wavefront 0 learns it with the first batch, reports it last)
    for (uint32_t base = beg; base < end; base += kBatch) {
        // ---- queue this batch: one record per thread, slot = thread index --------------
        uint32_t box_xy = 0, box_wh = 0;
        if (cur_ok) {
            int xl = (int)(cur_bx & 0xFFFF), xr = (int)(cur_bx >> 16);
            int yt = (int)(cur_by & 0xFFFF), yb = (int)(cur_by >> 16);
            if (xl < X0) xl = X0;
            if (xr > X1) xr = X1;
            if (yt < Y0) yt = Y0;
            if (yb > Y1) yb = Y1;
            if (xl < xr && yt < yb) {
                box_xy = (uint32_t)xl | ((uint32_t)yt << 16);
                box_wh = (uint32_t)(xr - xl) | ((uint32_t)(yb - yt) << 16);
                // A large triangle's pixel box covers about twice its area: a good part of the
                // entries of a frame of large triangles (bunny 4096^2: 8 per tile) name tiles the
                // triangle never touches.  Exact test on (box ∩ tile); not worth its ~80
                // instructions for a small box.
                if (TS >= 32 && (xr - xl) * (yb - yt) >= 256 &&
                    rect_surely_missed(make_setup(cur_t, false), xl, xr - 1, yt, yb - 1))
                    box_wh = 0;
            }
        }
        const uint32_t key_low = slotted ? ((0xFFFFu - (cur_id & 0xFFFFu)) << 16) | ((((base - beg) / kBatch) & 0xFFu) << 8) | (uint32_t)tid
                                         : 0xFFFFFFFEu - cur_id;
#ifdef CRENDER_STAMPS
        if (base == beg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CR_STAMP(5); }
#endif
        if constexpr (TS == 16) {
            // a short last batch goes pixel-parallel
            const uint32_t left = end - base;
            const uint32_t pix_max = (dbg >> 16) & 0xFF ? (uint32_t)((dbg >> 16) & 0xFF) - 1u : kPixelPathRecords;
            if (left <= pix_max) {
                pixel_path16(c, left, base == beg, cur_t, key_low, box_xy, box_wh);
                cur_ok = false;
                continue;   // (this was the list's last batch)
            }
        }
        // 16-pixel tiles: the work is flattened per PIXEL of the clipped boxes, not per block
        // (see the sweep below); elsewhere per 16-pixel block
        constexpr bool per_pixel = TS == 16;   // always the per-pixel sweep
        constexpr bool either = TS == 32;      // counted both ways, the batch picks its sweep
        // wave-inclusive scan of the work counts
        // (per-pixel work is counted in ITEMS of kItemPixels / kItemPixels32 x-neighbours of a box row)
        const uint32_t my_px = (uint32_t)(((box_w(box_wh) + kItemPixels32 - 1) / kItemPixels32) * box_h(box_wh));
        // (16-pixel tiles: an item is a PAIR of x-neighbours of a box row, see the sweep)
        const uint32_t my_blocks = per_pixel ? (uint32_t)(((box_w(box_wh) + kItemPixels - 1) / kItemPixels) * box_h(box_wh))
                                             : (uint32_t)blocks_of(box_wh);
        const uint32_t incl = wave_incl_sum(my_blocks);
        if constexpr (TS == 32) {
            // the tile's size class for the next frames' choice of kernel: the first wavefront's records
            if (base == beg && __builtin_amdgcn_readfirstlane(wave) == 0) tile_class = tile_class_of(incl, end - beg);   // (scalar)
        }
        uint32_t incl_px = my_px;
        if constexpr (either) incl_px = wave_incl_sum(my_px);
        // previous batch's sweeps must be over before the queue is overwritten (the first batch has
        // none before it: the barrier behind the queue orders the key initialisation too)
        if (base != beg) __syncthreads();
        if constexpr (TS == 16) {
            if (tid < kBatch) put_rec16(&recs[tid], cur_t, key_low, box_xy, box_wh);
            scan16[tid] = incl - my_blocks;
            if (lane == 63) wave16[wave] = incl;
        } else {
            q.x0[tid] = cur_t.x0; q.y0[tid] = cur_t.y0; q.z0[tid] = cur_t.z0;
            q.x1[tid] = cur_t.x1; q.y1[tid] = cur_t.y1; q.z1[tid] = cur_t.z1;
            q.x2[tid] = cur_t.x2; q.y2[tid] = cur_t.y2; q.z2[tid] = cur_t.z2;
            q.tri[tid] = cur_id;
            {
                const unsigned long long live = __builtin_amdgcn_ballot_w64(box_wh != 0);
                const unsigned long long after = lane == 63 ? 0ull : live >> (lane + 1);
                const uint32_t skip = after ? (uint32_t)__builtin_ctzll(after) : (uint32_t)(63 - lane);
                q.box[tid] = pack_box(box_xy, box_wh, X0, Y0) | (skip << 26);
            }
            if constexpr (!either) q.big.blk_scan[tid] = incl - my_blocks;     // (32-pixel tiles: once the sweep is chosen)
            if (lane == 63) q.wave_blocks[wave] = incl;
            if constexpr (either) {
                if (lane == 63) q.wave_px[wave] = incl_px;
                // every record's thread works out, ONCE, what an item of its record would otherwise work out
                // again (9 items of two pixels per record on the 10 M small triangles: 40 of an item's 175
                // vector instructions) — here, with the queue, not behind a barrier of its own once the batch
                // has picked its sweep: the pixel owners take their constants from here too, and a batch
                // that goes by culled blocks overwrites them (q.big shares the memory)
                if (box_wh != 0) {
                    const TriSetup mine = make_setup(cur_t, true);
                    q.pre.l03[tid] = mine.l03; q.pre.l13[tid] = mine.l13; q.pre.l23[tid] = mine.l23;
                    q.pre.r1[tid] = mine.fast ? mine.r1 : 0.0f; q.pre.r2[tid] = mine.r2; q.pre.r3[tid] = mine.r3;
                }
                q.pre.px_scan[tid] = incl_px - my_px;
            }
        }
        __syncthreads();  // queue complete
#ifdef CRENDER_STAMPS
        if (base == beg) CR_STAMP(6);
#endif

        // ---- sweep: the batch's work items, flattened and split evenly -------------------------
        {
            uint32_t wo[kThreads / 64 + 1];
            wo[0] = 0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) wo[w + 1] = wo[w] + (TS == 16 ? wave16[w] : q.wave_blocks[w]);
            const int total = (int)wo[kThreads / 64];
            const int nrec = (int)((end - base) < (uint32_t)kBatch ? (end - base) : (uint32_t)kBatch);
            const uint32_t blk_excl = incl - my_blocks;
            bool small_by_pixel = false;    // 32-pixel tiles: small records go per pixel too
            if constexpr (either) small_by_pixel = total < 16 * nrec && !(dbg & 8192);
#ifdef CRENDER_STAMPS
            if (base == beg) CR_STAMP(13);
#endif
            // next batch: issue its loads now, they complete under the sweeps
            const uint32_t nxt = base + kBatch + tid;
            cur_ok = tid < kBatch && nxt < end;
            if (cur_ok) cur_ok = load_record(L, proj, G, nxt, cur_id, cur_t, cur_bx, cur_by);
            if constexpr (TS == 32) {
                // the whole list is this one batch of large records: the pixels' owners take it from here
                if (base == beg && end - beg <= (uint32_t)kBatch && !small_by_pixel && !(dbg & 32768)) {
                    // (a heavy tile's hand-off words go back to zero HERE: this path returns, and every wavefront
                    // read them before the queue's barrier.  Without it a frame of large triangles on a split
                    // 32-pixel plan left its flags up, and the next frame cleared only the upper half of the
                    // tiles it no longer covered: test_dispatch_order_hint_never_changes_pixels[True-32].
                    // A `break` to the common exit instead costs 76 spilled registers.)
#if !defined(CRENDER_FAULT) || CRENDER_FAULT != 1      // (-DCRENDER_FAULT=1: round 4's defect back in, for the state check's own test)
                    if (quad >= 0 && tid == 0) {
                        if (!helper) L.heavy_flag[tile] = 0;
                        else L.heavy_slots[b] = 0;
                    }
#endif
                    owner_path32<CLEAR>(c, nrec);
                    if (tid == 0) count_tile_class(L, tile_class);
                    return;
                }
                if (base == beg && !keys_early) {
                    init_keys<TS, CLEAR>(c);
                    __syncthreads();
                }
#ifdef CRENDER_STAMPS
                if (base == beg) CR_STAMP(14);         // key plane initialised (composite frames)
#endif
            }
            if constexpr (per_pixel) {
                sweep_items<TS>(c, scan16, wo, total, nrec);
            } else if (small_by_pixel) {
                if constexpr (TS == 32) {
                    uint32_t wop[kThreads / 64 + 1];
                    wop[0] = 0;
#pragma unroll
                    for (int w = 0; w < kThreads / 64; ++w) wop[w + 1] = wop[w] + q.wave_px[w];
                    // (item by item — thread t takes items t, t + 256, ... with a search per item — only as a
                    // development knob: the run-wise walk is faster on every workload once the key plane is
                    // swizzled, T-Rex 1024^2 pipelined +12 %, 10 M small triangles' raster launch -4 %)
                    const int items = (int)wop[kThreads / 64];
#ifdef CRENDER_STAMPS
                    if (g_stamps && base == beg && tid == 0) g_stamps[stamp_base + 12] = (unsigned long long)items;
#endif
#ifdef CRENDER_RUNS_MAX_AVG
                    if ((dbg & (1 << 30)) || items > CRENDER_RUNS_MAX_AVG * nrec) sweep_items<TS>(c, q.pre.px_scan, wop, items, nrec); else
#else
                    if (dbg & (1 << 30)) sweep_items<TS>(c, q.pre.px_scan, wop, items, nrec); else
#endif
                    sweep_runs32(c, wop, items);
#ifdef CRENDER_STAMPS
                    if (base == beg) CR_STAMP(15);     // thread 0's run of the first batch walked
#endif
                }
            } else if constexpr (TS == 64) {
                if ((total < 16 * nrec && !(dbg & 8192)) || (dbg & 128)) walk64_small(c, wo, total);
                else walk64_dense(c, wo, total);
            } else {
#ifdef CRENDER_STAMPS
                if (g_stamps && base == beg && tid == 0) g_stamps[stamp_base + 12] = (1ull << 32) | (unsigned long long)total;
#endif
                sweep_blocks_culled<TS>(c, wo, total, blk_excl);
            }
        }
    }
    __syncthreads();

    CR_STAMP(2);
    resolve_tile<TS, CLEAR>(c, slotted);
    if constexpr (TS == 32) if (tid == 0) count_tile_class(L, tile_class);
    CR_STAMP(3);
#ifdef CRENDER_STAMPS
    if (g_stamps && threadIdx.x == 0) g_stamps[stamp_base + 11] = __builtin_amdgcn_s_memtime();
#endif
    }   // work
    // hand-off words of a heavy tile go back to zero once every wavefront has read them
    if (quad >= 0) {
        __syncthreads();
        if (tid == 0) {
            if (!helper) L.heavy_flag[tile] = 0;
            else L.heavy_slots[b] = 0;
        }
    }
}

// LDS and wavefronts per SIMD of a kernel by tile size and path (see kPath*)
template <int TS, int PATH>
constexpr size_t path_queue_bytes()
{
    return PATH == kPathOwners ? sizeof(OwnerQueue) : raster_queue_bytes<TS>();
}
template <int TS, int PATH>
constexpr int path_waves()
{
    return TS == 16 ? kWavesPerSimd16 : TS != 32 ? 1 : PATH == kPathOwners ? kWavesPerSimdOwners : kWavesPerSimd32;
}
template <int TS, bool CLEAR, int PATH = kPathGeneral>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(path_waves<TS, PATH>())))
void k_raster(const float *__restrict__ proj, const float *__restrict__ col,
              const float *__restrict__ nrm, TileLists L,
              float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
              int32_t *__restrict__ win, Geom G, int dbg_arg)
{
    // (the pixel owners read the plane as float4: their eight words per record — the whole of it in kPathOwners)
    __shared__ __attribute__((aligned(16))) unsigned long long key[TS * TS];
    __shared__ __attribute__((aligned(16))) unsigned char qraw[path_queue_bytes<TS, PATH>()];
    raster_body<TS, CLEAR, PATH>(proj, col, nrm, L, zb, cb, nb, win, G, dbg_arg, (int)blockIdx.x, key, qraw);
}

// One launch per frame for a stream of frames (crender_pipeline_*, direct bins): the raster pass of
// frame i and, in its first `nsetup` workgroups, the binning pass of the NEXT frame on the same
// stream — k_setup_wave's wavefronts, one per workgroup (threads 64..255 leave at once), working
// into another plan.  Nothing in the launch depends on anything else in it.  The binning pass is a
// latency chain of 216 wavefronts (T-Rex) that a launch of its own stretches to 8.6 us; here it
// costs neither a launch nor the GPU's time between two launches of a stream.
struct RasterArgs {
    const float *proj, *col, *nrm;
    TileLists L;
    float *zb, *cb, *nb;
    int32_t *win;
    Geom G;
    int dbg;
};
static_assert(kSetupWaveLds <= raster_queue_bytes<16>() && kSetupWaveLds <= raster_queue_bytes<32>() &&
              kSetupWaveLds <= sizeof(OwnerQueue),
              "a binning wavefront works in the raster workgroup's batch queue");
static_assert(sizeof(OwnerQueue) >= 128 + kOrderMaxTiles, "build_order keeps a byte per tile in the queue");
struct FrameArgs {
    RasterArgs R;
    SetupArgs S;
    int nsetup;
};
template <int TS, bool CLEAR, int PATH = kPathGeneral>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(path_waves<TS, PATH>())))
void k_frame(FrameArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned long long key[TS * TS];   // (the pixel owners read it as float4)
    __shared__ __attribute__((aligned(16))) unsigned char qraw[path_queue_bytes<TS, PATH>()];
    if ((int)blockIdx.x < A.nsetup) {
        if (threadIdx.x < kWave)
            setup_wave_body<TS, true>(A.S.tri_in, A.S.nrm, A.S.proj_out, A.S.count, A.S.bins, A.S.dcap, A.S.hdr,
                                      A.S.hv, A.S.T, A.S.P, A.S.G, (int64_t)blockIdx.x, qraw);
        return;
    }
    raster_body<TS, CLEAR, PATH>(A.R.proj, A.R.col, A.R.nrm, A.R.L, A.R.zb, A.R.cb, A.R.nb, A.R.win, A.R.G, A.R.dbg,
                                 (int)blockIdx.x - A.nsetup, key, qraw);
}

// ---- second implementation: global 64-bit atomics ---------------------------------
__global__ __launch_bounds__(kThreads) void k_keys_init(unsigned long long *__restrict__ keys,
                                                        const float *__restrict__ zb,
                                                        size_t first_pix, size_t npix, int clear)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    const unsigned long long key_clear = make_key(zord(1e6f), KEY_LOW_PRIOR);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npix; i += stride)
        keys[first_pix + i] = clear ? key_clear : make_key(zord_prior(zb[first_pix + i]), KEY_LOW_PRIOR);
}

// one wavefront per triangle, 8x8 pixel steps over the pixel box
__global__ __launch_bounds__(kThreads) void k_cover_atomic(const float *__restrict__ proj,
                                                           const float *__restrict__ nrm,
                                                           unsigned long long *__restrict__ keys,
                                                           int64_t T, int W, int H, int y0, int y1)
{
    const int l = threadIdx.x & 63, lx = l & 7, ly = l >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
    for (int64_t t = wave; t < T; t += nwaves) {
        const float *nn = nrm + t * 9;
        if (backface(nn[2], nn[5], nn[8])) continue;
        const TriXYZ tr = load_tri(proj + t * 9);
        int xl, xr, yt, yb;
        pixel_box(tr.x0, tr.y0, tr.x1, tr.y1, tr.x2, tr.y2, W, H, xl, xr, yt, yb);
        if (xl - xr == 0 || yt - yb == 0) continue;  // .pyx:209
        if (yt < y0) yt = y0;
        if (yb > y1) yb = y1;
        for (int yy = yt + ly; yy < yb; yy += 8)
            for (int xx = xl + lx; xx < xr; xx += 8) {
                unsigned long long k;
                if (fragment(tr, (uint32_t)t, xx, yy, k)) {
                    unsigned long long *slot = &keys[(size_t)yy * W + xx];
                    if (k < *slot) atomicMin(slot, k);
                }
            }
    }
}

__global__ __launch_bounds__(kThreads) void k_resolve_global(const float *__restrict__ proj,
                                                             const float *__restrict__ col,
                                                             const float *__restrict__ nrm,
                                                             const unsigned long long *__restrict__ keys,
                                                             float *__restrict__ zb, float *__restrict__ cb,
                                                             float *__restrict__ nb, int32_t *__restrict__ win,
                                                             int W, size_t first_pix, size_t npix, int clear)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npix; i += stride) {
        const size_t pix = first_pix + i;
        const uint32_t low = (uint32_t)keys[pix];
        if (low == KEY_LOW_PRIOR) {
            if (clear) {
                zb[pix] = 1e6f;
                cb[pix * 3] = 0.0f; cb[pix * 3 + 1] = 0.0f; cb[pix * 3 + 2] = 0.0f;
                nb[pix * 3] = 0.0f; nb[pix * 3 + 1] = 0.0f; nb[pix * 3 + 2] = 0.0f;
                if (win) win[pix] = -1;
            }
            continue;
        }
        const uint32_t id = 0xFFFFFFFEu - low;
        shade_and_store(proj, col, nrm, id, (int)(pix % W), (int)(pix / W), pix, zb, cb, nb);
        if (win) win[pix] = (int32_t)id;
    }
}

// ---- self-check hook: shortcut (2) of raster_math.h against the plain division ----------
__global__ __launch_bounds__(kThreads) void k_divcheck(const float *__restrict__ num,
                                                       const float *__restrict__ den,
                                                       float *__restrict__ out_tail,
                                                       float *__restrict__ out_div, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float a = num[i], d = den[i];
        const bool win = in_div_window(a) && in_div_window(d);
        out_tail[i] = win ? div_tail(a, d, refined_rcp(d)) : a / d;
        out_div[i] = a / d;
    }
}

}  // namespace

namespace crender_detail {

// The size classes the plan's tiles reported last (TileLists::stats) -> the kernel its next frames get.
// Reads the plan's own pinned records, newest first: the record of launch t carries the sums of launch
// t - 1 (0, 0 for a plan's first launch and for other tile sizes: no opinion).  Three quarters of the covered
// tiles holding large records make a frame the pixel owners'; anything else keeps the general kernel.
// Until a record with an opinion has landed the triangle count decides: at most four triangles per tile are
// large ones if they cover anything (bunny 4096^2: 1.9, T-Rex 8192^2: 0.2; T-Rex 1024^2 on 32-pixel tiles: 13).
// True: a record newer than the last one taken was there.
bool raster_path_hint(crender_plan *plan)
{
    if (plan->L.ts != 32) return false;
    for (uint64_t back = 0; back + 1 < (uint64_t)kUsageRing && back < plan->ticket; ++back) {
        const uint64_t t = plan->ticket - back;
        if (t <= plan->hint_ticket) break;
        const volatile uint32_t *rec = plan->usage + kUsageWords * (int)(t % kUsageRing);
        const uint32_t seq = (uint32_t)t ^ plan->usage_salt;
        if (__atomic_load_n(const_cast<const uint32_t *>(rec), __ATOMIC_ACQUIRE) != seq || rec[7] != seq) continue;
        const uint32_t nl = rec[4], ns = rec[5];
        if (rec[0] != seq || rec[7] != seq) continue;       // (rewritten under the read)
        plan->hint_ticket = t;
        if (nl + ns == 0) return false;
        plan->auto_path = nl >= 3u * ns ? kPathOwners : kPathGeneral;
        plan->auto_known = true;
        return true;
    }
    if (!plan->auto_known && plan->last_T >= 0)
        plan->auto_path = plan->last_T <= 4 * (int64_t)plan->L.g.ntiles ? kPathOwners : kPathGeneral;
    return false;
}

}  // namespace crender_detail

namespace {

template <int TS>
int run_raster_pass(crender_plan *plan, const float *proj, const float *d_col, const float *d_nrm,
                    float *d_z, float *d_color, float *d_normal, int32_t *d_winner, unsigned flags,
                    hipStream_t s, const SetupArgs *with_setup = nullptr)
{
    const Layout &L = plan->L;
    const Geom G = L.g;
#ifdef CRENDER_STAMPS
    const int dbg = dev_knobs() | (plan->stamp_slot << 24);
#else
    const int dbg = dev_knobs();
#endif
    const bool direct = plan->last_frame_direct;          // direct bins of 48-byte entries (small scenes)
    const bool pairbins = plan->last_frame_pairbins;      // fixed-capacity slabs of (position, index) pairs (k_bin_wave)
    const int par = plan->parity;
    TileLists tl;
    tl.offs = direct || pairbins ? nullptr : plan->offs();
    tl.count = plan->count(par);
    tl.count_next = plan->count(par ^ 1);
    tl.entries = pairbins ? reinterpret_cast<const uint32_t *>(plan->pairbins()) : plan->entries();
    tl.pairs = !direct && plan->last_frame_pairs;
    tl.bins = plan->direct();
    tl.capacity = direct ? (uint32_t)L.direct_cap : pairbins ? (uint32_t)L.pair_cap : (uint32_t)L.capacity;
    tl.T = (uint32_t)plan->last_T;
    tl.orig_of = plan->orig_of;
    tl.pos_of = plan->pos_of;
    const bool split = direct && plan->frame_hmax > 0 && !(dbg & 2048);    // (lone frames: the plan's thresholds; frames in flight: long lists only)
    tl.heavy_flag = split ? plan->hflag() : nullptr;
    tl.heavy_slots = split ? plan->hslots() : nullptr;
    tl.heavy_ctr_next = plan->hdr() + 2 + (par ^ 1);
    tl.heavy_ctr = plan->hdr() + 2 + par;
    tl.nhelp = split ? 3 * plan->frame_hmax : 0;
    tl.quad_at = plan->frame_quad_at;
    // ordered launches: read the order the previous launch left, leave one for the next
    const bool ordered = direct && L.ordered && plan->frame_lone && !(dbg & 1024);
    const int hp = plan->hint_par;
    tl.order = ordered ? plan->order(hp) : nullptr;
    tl.hint = plan->hint(hp);
    tl.order_next = ordered ? plan->order(hp ^ 1) : nullptr;
    tl.hint_next = plan->hint(hp ^ 1);
    tl.grouped_next = ordered ? plan->grouped(hp ^ 1) : nullptr;
    tl.hint_bad = plan->hdr() + 5 + par;
    tl.hint_bad_next = plan->hdr() + 5 + (par ^ 1);
    tl.addr32 = (uint64_t)G.H * (uint64_t)G.W * 12ull < (1ull << 32) &&
                (uint64_t)(plan->last_T > 0 ? plan->last_T : 1) * 36ull < (1ull << 32);
    if (ordered) plan->hint_par = hp ^ 1;
    plan->last_ordered = ordered;
    const uintptr_t any = (uintptr_t)d_z | (uintptr_t)d_color | (uintptr_t)d_normal | (uintptr_t)d_winner;
    tl.vec_clear = (any & 15u) == 0 && (G.W & 3) == 0;
    tl.light = Light{plan->light[0], plan->light[1], plan->light[2], (flags & CRENDER_FUSED_GURO) ? 1 : 0};
    // Which kernel (32-pixel plans; kPath*): the caller's choice (crender_plan_set_raster_path, or the
    // process-wide default), else what the size classes of the plan's last reported frame suggest — a landed
    // usage record of one of the two launches before this one carries them.  Every choice renders every
    // tile exactly; a wrong one costs time.
    int path = kPathGeneral;
    if constexpr (TS == 32) {
        raster_path_hint(plan);
        path = plan->forced_path >= 0 ? plan->forced_path : default_raster_path() >= 0 ? default_raster_path() : plan->auto_path;
    }
    plan->last_path = path;
    // this launch's usage record (crender_plan_poll_bin_usage)
    plan->ticket++;
    const int uslot = (int)(plan->ticket % kUsageRing);
    plan->usage_mode[uslot] = direct ? 1 : pairbins ? 2 : 0;
    tl.hdr = plan->hdr();
    tl.usage = plan->usage_dev + kUsageWords * uslot;
    tl.usage_seq = (uint32_t)plan->ticket ^ plan->usage_salt;
    // (not on the split launches of small frames rendered alone: a device-scope atomic takes 1.5-2 us to come
    // back, and the one of the tile that ends such a launch ends it that much later — lone T-Rex 1024^2 16.8
    // against 15.2 us, profiles/r06/ab_stats_atomics.txt; on a frame of 130 us and more it is noise)
    tl.stats = TS == 32 && !(split && plan->frame_lone) ? plan->stats((int)(plan->ticket & 1u)) : nullptr;
    tl.stats_prev = TS == 32 ? plan->stats((int)((plan->ticket & 1u) ^ 1u)) : nullptr;
    tl.path = (uint32_t)path;
    const unsigned grid = (unsigned)(G.ntiles + tl.nhelp + (ordered ? 1 : 0));
    plan->awaiting[par ^ 1] = false;     // zeroed by this launch
    plan->unrastered[par] = false;
    const bool clear = (flags & CRENDER_FUSED_CLEAR) != 0;
#define CR_LAUNCH_FRAME(P)                                                                                              \
    do {                                                                                                                \
        if (clear) hipLaunchKernelGGL((k_frame<TS, true, P>), dim3(grid + (unsigned)nsetup), dim3(kThreads), 0, s, fa); \
        else hipLaunchKernelGGL((k_frame<TS, false, P>), dim3(grid + (unsigned)nsetup), dim3(kThreads), 0, s, fa);      \
    } while (0)
#define CR_LAUNCH_RASTER(P)                                                                                             \
    do {                                                                                                                \
        if (clear) hipLaunchKernelGGL((k_raster<TS, true, P>), dim3(grid), dim3(kThreads), 0, s, proj, d_col, d_nrm,    \
                                      tl, d_z, d_color, d_normal, d_winner, G, dbg);                                    \
        else hipLaunchKernelGGL((k_raster<TS, false, P>), dim3(grid), dim3(kThreads), 0, s, proj, d_col, d_nrm,         \
                                tl, d_z, d_color, d_normal, d_winner, G, dbg);                                          \
    } while (0)
    if constexpr (TS <= 32) {
        if (with_setup) {
            // this frame's raster pass and another plan's binning pass in one launch (k_frame)
            const int nsetup = (int)((with_setup->T + kWave - 1) / kWave);
            const RasterArgs ra{proj, d_col, d_nrm, tl, d_z, d_color, d_normal, d_winner, G, dbg};
            const FrameArgs fa{ra, *with_setup, nsetup};
            if constexpr (TS == 32) {
                if (path == kPathOwners) CR_LAUNCH_FRAME(kPathOwners);
                else CR_LAUNCH_FRAME(kPathGeneral);
            } else {
                CR_LAUNCH_FRAME(kPathGeneral);
            }
            CR_LAUNCH_CHECK("k_frame");
            return CRENDER_OK;
        }
    }
    if constexpr (TS == 32) {
        if (path == kPathOwners) CR_LAUNCH_RASTER(kPathOwners);
        else CR_LAUNCH_RASTER(kPathGeneral);
    } else {
        CR_LAUNCH_RASTER(kPathGeneral);
    }
#undef CR_LAUNCH_FRAME
#undef CR_LAUNCH_RASTER
    CR_LAUNCH_CHECK("k_raster");
    return CRENDER_OK;
}

}  // namespace

namespace crender_detail {

int raster_pass(crender_plan *plan, const float *proj, const float *d_col, const float *d_nrm, int64_t T,
                float *d_z, float *d_color, float *d_normal, int32_t *d_winner, unsigned flags,
                void *stream, const SetupArgs *with_setup)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    if (T != plan->last_T) return fail(CRENDER_EINVAL, "T differs from the prepared frame's");
    if ((flags & CRENDER_FUSED_GURO) && !(flags & CRENDER_FUSED_CLEAR))
        return fail(CRENDER_EINVAL, "CRENDER_FUSED_GURO needs CRENDER_FUSED_CLEAR (the reference shades the whole "
                                    "buffer after every render: only a frame that starts from cleared buffers "
                                    "can shade its own pixels instead)");
    if (!d_z || !d_color || !d_normal) return fail(CRENDER_EINVAL, "null framebuffer pointer");
    if (T > 0 && (!proj || !d_col || !d_nrm)) return fail(CRENDER_EINVAL, "null triangle array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return CR_BY_TILE(run_raster_pass<16>(plan, proj, d_col, d_nrm, d_z, d_color, d_normal, d_winner, flags, s, with_setup),
                      run_raster_pass<32>(plan, proj, d_col, d_nrm, d_z, d_color, d_normal, d_winner, flags, s, with_setup),
                      run_raster_pass<64>(plan, proj, d_col, d_nrm, d_z, d_color, d_normal, d_winner, flags, s));
}

}  // namespace crender_detail

extern "C" {

int crender_selfcheck_division(const float *d_num, const float *d_den, float *d_out_tail,
                               float *d_out_div, int64_t n, void *stream)
{
    if (!d_num || !d_den || !d_out_tail || !d_out_div || n < 0)
        return fail(CRENDER_EINVAL, "crender_selfcheck_division: bad argument");
    if (n == 0) return CRENDER_OK;
    hipLaunchKernelGGL(k_divcheck, dim3(grid_for((size_t)n, 8192)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_num, d_den, d_out_tail, d_out_div, (size_t)n);
    CR_LAUNCH_CHECK("k_divcheck");
    return CRENDER_OK;
}

size_t crender_atomic_scratch_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    return sizeof(unsigned long long) * (size_t)H * (size_t)W;
}

int crender_raster_atomic(const float *d_tri_proj, const float *d_col, const float *d_nrm, int64_t T,
                          float *d_z, float *d_color, float *d_normal, int32_t *d_winner, int H, int W,
                          int y0, int y1, unsigned flags, void *d_keys, void *stream)
{
    if (!d_z || !d_color || !d_normal || !d_keys || H <= 0 || W <= 0 || y0 < 0 || y1 > H || y0 >= y1 ||
        T < 0 || T > 0xFFFFFFF0ll)
        return fail(CRENDER_EINVAL, "crender_raster_atomic: bad argument");
    if (T > 0 && (!d_tri_proj || !d_col || !d_nrm))
        return fail(CRENDER_EINVAL, "crender_raster_atomic: null triangle array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t first = (size_t)y0 * W, npix = (size_t)(y1 - y0) * W;
    const int clear = (flags & CRENDER_FUSED_CLEAR) ? 1 : 0;
    unsigned long long *keys = static_cast<unsigned long long *>(d_keys);
    hipLaunchKernelGGL(k_keys_init, dim3(grid_for(npix, 4096)), dim3(kThreads), 0, s, keys, d_z, first,
                       npix, clear);
    CR_LAUNCH_CHECK("k_keys_init");
    if (T > 0) {
        hipLaunchKernelGGL(k_cover_atomic, dim3(grid_for((size_t)T * 64, 8192)), dim3(kThreads), 0, s,
                           d_tri_proj, d_nrm, keys, T, W, H, y0, y1);
        CR_LAUNCH_CHECK("k_cover_atomic");
    }
    hipLaunchKernelGGL(k_resolve_global, dim3(grid_for(npix, 8192)), dim3(kThreads), 0, s, d_tri_proj,
                       d_col, d_nrm, keys, d_z, d_color, d_normal, d_winner, W, first, npix, clear);
    CR_LAUNCH_CHECK("k_resolve_global");
    return CRENDER_OK;
}

#ifdef CRENDER_STAMPS
CRENDER_API int crender_debug_set_stamps(void *d_buf);
// diagnostic build: point the kernels at a stamp buffer (ntiles * 8 u64), or detach with null
int crender_debug_set_stamps(void *d_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(d_buf);
    CR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p));
    return CRENDER_OK;
}
#endif

}  // extern "C"
