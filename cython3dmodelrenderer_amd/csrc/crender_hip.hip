// crender_hip.hip — gfx950 (MI355X / CDNA4) kernels and the C ABI of include/crender_hip.h.
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
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/crender_hip.h"
#include "raster_math.h"

using namespace crender;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *what)
{
    g_last_error = what;
    return code;
}

int fail_hip(hipError_t e, const char *where)
{
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return CRENDER_EHIP;
}

#define CR_HIP(expr)                                             \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return fail_hip(_e, #expr);        \
    } while (0)

#define CR_LAUNCH_CHECK(name)                                    \
    do {                                                         \
        hipError_t _e = hipGetLastError();                       \
        if (_e != hipSuccess) return fail_hip(_e, name);         \
    } while (0)

constexpr int kThreads = 256;           // 4 wavefronts per workgroup
// Waves per SIMD asked of k_raster (an upper bound on its VGPRs): the 16-pixel kernel needs 72
// registers as it is (7 waves); the 32-pixel kernel is held to 80 (6 waves, 6 workgroups per
// CU instead of 5: bunny 4096^2 +5 %, T-Rex 8192^2 raster 0.381 -> 0.341 ms; the fused-clear
// instantiation fits without spilling, the compositing one spills 4 registers).  The 64-pixel
// kernel is limited by its 48 KB of LDS, not by registers.  (r01 A/B, same box.)
constexpr int kWavesPerSimd16 = 7, kWavesPerSimd32 = 6;   // (32-pixel tiles: 28.7 KB of LDS = 5 workgroups per CU)
constexpr int kItemPixels = 2;      // samples per work item of the per-pixel sweep of 16-pixel tiles
constexpr int kItemPixels32 = 2;    // the same for the small-record batches of 32-pixel tiles
constexpr uint32_t kPixelPathRecords = 8;   // k_raster<16>: batches this short go pixel-parallel
constexpr uint32_t kNoTiles = 0xFFFFFFFFu;

// Strip geometry shared by the binning and raster kernels.
struct Geom {
    int W, H;      // full frame
    int y0, y1;    // strip rows
    int ntx, nty;  // tiles across / down the strip
    int ntiles;
    int tile_stride;  // odd-ish multiplier coprime to ntiles: scatters the dispatch order
    // k_raster decodes its tile index once per workgroup; a runtime integer division costs ~30
    // scalar instructions behind a v_rcp, three of them were 0.3 us at the head of every tile
    // (and most of the kernel's SALU instructions).  Division by multiplication instead:
    uint32_t ntx_magic;   // floor(2^32 / ntx) + 1: n / ntx == umulhi(n, magic) for n * ntx < 2^32; 0 = divide
    double inv_ntiles;    // 1 / ntiles, for the 48-bit product of the scatter map
};

ProjConst make_proj(const float *P16, int w, int h)
{
    ProjConst P;
    std::memcpy(P.p, P16, sizeof P.p);
    P.xs = (float)((double)w / 2.0);   // .pyx:109
    P.ys = (float)((double)h / 2.0);
    return P;
}

// ---- coalesced staging of [n][9] float chunks through LDS -------------------------
// Triangle records are 36 B, so per-thread vector loads would be misaligned; a block
// copies its contiguous chunk with unit-stride loads and each thread then reads its own
// record at a 9-dword stride (odd => conflict-free across the 32 banks).
template <int NT = kThreads>
CR_DEV void stage_in(const float *__restrict__ g, float *__restrict__ s, int nfloats)
{
    const bool aligned = (((uintptr_t)g) & 15u) == 0;
    if (aligned) {
        const int n4 = nfloats >> 2;
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        float4 *s4 = reinterpret_cast<float4 *>(s);
        for (int i = threadIdx.x; i < n4; i += NT) s4[i] = g4[i];
        for (int i = (n4 << 2) + threadIdx.x; i < nfloats; i += NT) s[i] = g[i];
    } else {
        for (int i = threadIdx.x; i < nfloats; i += NT) s[i] = g[i];
    }
}

template <int NT = kThreads>
CR_DEV void stage_out(float *__restrict__ g, const float *__restrict__ s, int nfloats)
{
    const bool aligned = (((uintptr_t)g) & 15u) == 0;
    if (aligned) {
        const int n4 = nfloats >> 2;
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        for (int i = threadIdx.x; i < n4; i += NT) g4[i] = s4[i];
        for (int i = (n4 << 2) + threadIdx.x; i < nfloats; i += NT) g[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < nfloats; i += NT) g[i] = s[i];
    }
}

// ---- K1 standalone: project_on_screen_multithread, .pyx:106-130 -------------------
__global__ __launch_bounds__(kThreads) void k_project(const float *__restrict__ in,
                                                      float *__restrict__ out, int64_t T,
                                                      ProjConst P)
{
    __shared__ __attribute__((aligned(16))) float s[kThreads * 9];
    for (int64_t b0 = (int64_t)blockIdx.x * kThreads; b0 < T; b0 += (int64_t)gridDim.x * kThreads) {
        const int n = (int)((T - b0) < kThreads ? (T - b0) : kThreads);
        stage_in(in + b0 * 9, s, n * 9);
        __syncthreads();
        if ((int)threadIdx.x < n) {
            float *v = s + threadIdx.x * 9;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float r[3] = {v[3 * c], v[3 * c + 1], v[3 * c + 2]};
                project_vertex(P, r);
                v[3 * c] = r[0];
                v[3 * c + 1] = r[1];
                v[3 * c + 2] = r[2];
            }
        }
        __syncthreads();
        stage_out(out + b0 * 9, s, n * 9);
        __syncthreads();
    }
}

// ---- clear: __cinit__ buffer state, .pyx:65-67 ------------------------------------
__global__ __launch_bounds__(kThreads) void k_clear(float *__restrict__ zb, float *__restrict__ cb,
                                                    float *__restrict__ nb, int32_t *__restrict__ win,
                                                    size_t first_pix, size_t npix)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npix; i += stride) {
        zb[first_pix + i] = 1e6f;
        if (win) win[first_pix + i] = -1;
    }
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npix * 3; i += stride) {
        cb[first_pix * 3 + i] = 0.0f;
        nb[first_pix * 3 + i] = 0.0f;
    }
}

// Minimum / maximum over the 64 lanes of a wavefront, left in every lane: DPP row shifts and row
// broadcasts carry the running value to lane 63, one readlane hands it out (a butterfly of
// __shfl_xor is six ds_swizzle / ds_bpermute round trips per value, in the middle of the binning
// wavefronts' latency chain).
template <bool MAX>
CR_DEV int wave_reduce(int v)
{
    const int id = MAX ? (int)0x80000000 : 0x7FFFFFFF;
    auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = op(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
CR_DEV void wave_box(int &X0, int &X1, int &Y0, int &Y1)
{
    X0 = wave_reduce<false>(X0); X1 = wave_reduce<true>(X1);
    Y0 = wave_reduce<false>(Y0); Y1 = wave_reduce<true>(Y1);
}

// ---- binning ----------------------------------------------------------------------
// Tile range of a triangle packed as tx0 | tx1 << 16 (x) and ty0 | ty1 << 16 (y),
// inclusive; kNoTiles in .x marks a culled / empty triangle.  `bx`, `by` receive the pixel box
// (xl | xr << 16, yt | yb << 16; rows clipped to the strip).
template <int TS>
CR_DEV uint2 tile_range(const TriXYZ &t, const Geom &G, uint32_t &bx, uint32_t &by)
{
    int xl, xr, yt, yb;
    pixel_box(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2, G.W, G.H, xl, xr, yt, yb);
    // .pyx:209 skips an empty box; rows outside the strip never produce samples.
    if (yt < G.y0) yt = G.y0;
    if (yb > G.y1) yb = G.y1;
    bx = (uint32_t)xl | ((uint32_t)xr << 16);
    by = (uint32_t)yt | ((uint32_t)yb << 16);
    if (xl >= xr || yt >= yb) return make_uint2(kNoTiles, 0);
    const uint32_t tx0 = xl / TS, tx1 = (xr - 1) / TS;
    const uint32_t ty0 = (yt - G.y0) / TS, ty1 = (yb - 1 - G.y0) / TS;
    return make_uint2(tx0 | (tx1 << 16), ty0 | (ty1 << 16));
}
template <int TS>
CR_DEV uint2 tile_range(const TriXYZ &t, const Geom &G)
{
    uint32_t bx, by;
    return tile_range<TS>(t, G, bx, by);
}

// Visit every tile of each lane's tile range (r.x == kNoTiles: none).  Narrow ranges are
// walked by their own lane; a range wider than kWideTiles is walked by the whole wavefront,
// 64 tiles at a time, so one screen-filling triangle does not serialise a wavefront behind a
// single lane.  Must be called by all 64 lanes.  f(tx, ty, lane that owns the range).
constexpr int kWideTiles = 16;
template <typename F>
CR_DEV void for_each_tile_xy(uint2 r, F f)
{
    const int lane = threadIdx.x & 63;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (r.x != kNoTiles) {
        tx0 = r.x & 0xFFFF; tx1 = r.x >> 16; ty0 = r.y & 0xFFFF; ty1 = r.y >> 16;
    }
    const int mine = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    const bool wide = mine > kWideTiles;
    if (!wide)
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) f(tx, ty, lane);
    unsigned long long m = __ballot(wide);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t rx = __shfl(r.x, src, 64), ry = __shfl(r.y, src, 64);
        const int sx0 = rx & 0xFFFF, sx1 = rx >> 16, sy0 = ry & 0xFFFF, sy1 = ry >> 16;
        const int w = sx1 - sx0 + 1, n = w * (sy1 - sy0 + 1);
        for (int i = lane; i < n; i += 64) {
            const int dy = i / w;
            f(sx0 + (i - dy * w), sy0 + dy, src);
        }
    }
}
// the same with a tile index and the owner's payload (a shuffle from a lane that has left a
// divergent loop would read nothing, so the payload is fetched while every lane is active)
template <typename F>
CR_DEV void for_each_tile(uint2 r, uint32_t payload, int ntx, F f)
{
    const int lane = threadIdx.x & 63;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (r.x != kNoTiles) {
        tx0 = r.x & 0xFFFF; tx1 = r.x >> 16; ty0 = r.y & 0xFFFF; ty1 = r.y >> 16;
    }
    const int mine = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    const bool wide = mine > kWideTiles;
    if (!wide)
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) f(ty * ntx + tx, payload);
    unsigned long long m = __ballot(wide);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t rx = __shfl(r.x, src, 64), ry = __shfl(r.y, src, 64);
        const uint32_t pay = __shfl(payload, src, 64);
        const int sx0 = rx & 0xFFFF, sx1 = rx >> 16, sy0 = ry & 0xFFFF, sy1 = ry >> 16;
        const int w = sx1 - sx0 + 1, n = w * (sy1 - sy0 + 1);
        for (int i = lane; i < n; i += 64) {
            const int dy = i / w;
            f((sy0 + dy) * ntx + sx0 + (i - dy * w), pay);
        }
    }
}

// Binning mode of k_setup (scan path: scenes of more than 65536 triangles, or direct bins
// switched off): count list lengths in an LDS histogram or with global atomics and store each
// triangle's tile range for k_fill.
enum { kBinCountLds = 0, kBinCountGlobal = 1 };
constexpr int kDirectMaxTilesPerTriangle = 4096;  // beyond this the scan path is used instead

#ifdef CRENDER_STAMPS
// Diagnostic build only: phase timestamps per workgroup of the setup kernels, 8 per workgroup
// (crender_debug_set_setup_stamps); same clock as k_raster's stamps.
__device__ unsigned long long *g_setup_stamps = nullptr;
#define CR_SETUP_STAMP(slot)                                                                 \
    do {                                                                                     \
        if (g_setup_stamps && threadIdx.x == 0)                                              \
            g_setup_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64();                \
    } while (0)
#else
#define CR_SETUP_STAMP(slot) do { } while (0)
#endif

// dynamic LDS: [hist: ntiles u32 if kBinCountLds][verts: 256*9 f32][normals: 256*9 f32]
template <int TS, bool PROJECT, int BIN>
__global__ __launch_bounds__(kThreads) void k_setup(const float *__restrict__ tri_in,
                                                    const float *__restrict__ nrm,
                                                    float *__restrict__ proj_out,
                                                    uint2 *__restrict__ trange,
                                                    uint32_t *__restrict__ count, int64_t T,
                                                    int64_t chunk, ProjConst P, Geom G)
{
    constexpr bool LDS_HIST = BIN == kBinCountLds;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // two 16-bit counters per word (a block's chunk is < 65536 triangles)
    const int hist_words = LDS_HIST ? ((((G.ntiles + 1) >> 1) + 3) & ~3) : 0;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw);
    float *sv = reinterpret_cast<float *>(smem_raw) + hist_words;
    float *sn = sv + kThreads * 9;

    if (LDS_HIST) {
        for (int i = threadIdx.x; i < hist_words; i += kThreads) hist[i] = 0;
    }
    const int64_t c0 = (int64_t)blockIdx.x * chunk;
    const int64_t c1 = (c0 + chunk < T) ? (c0 + chunk) : T;
    __syncthreads();
    for (int64_t b0 = c0; b0 < c1; b0 += kThreads) {
        const int n = (int)((c1 - b0) < kThreads ? (c1 - b0) : kThreads);
        stage_in(tri_in + b0 * 9, sv, n * 9);
        stage_in(nrm + b0 * 9, sn, n * 9);
        __syncthreads();
        uint2 r_keep = make_uint2(kNoTiles, 0);
        if ((int)threadIdx.x < n) {
            float *v = sv + threadIdx.x * 9;
            const float *nn = sn + threadIdx.x * 9;
            float a[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = v[i];
            if (PROJECT) {
#pragma unroll
                for (int c = 0; c < 3; ++c) project_vertex(P, a + 3 * c);
#pragma unroll
                for (int i = 0; i < 9; ++i) v[i] = a[i];
            }
            uint2 r = make_uint2(kNoTiles, 0);
            if (!backface(nn[2], nn[5], nn[8])) {
                const TriXYZ t{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]};
                r = tile_range<TS>(t, G);
            }
            trange[b0 + threadIdx.x] = r;
            r_keep = r;
        }
        // list lengths: LDS histogram or global counters
        for_each_tile(r_keep, 0u, G.ntx, [&](int tile, uint32_t) {
            if (LDS_HIST) atomicAdd(&hist[tile >> 1], (tile & 1) ? 0x10000u : 1u);
            else atomicAdd(&count[tile], 1u);
        });
        __syncthreads();
        if (PROJECT) stage_out(proj_out + b0 * 9, sv, n * 9);
        __syncthreads();
    }
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < G.ntiles; i += kThreads) {
            const uint32_t c = (hist[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
            if (c) atomicAdd(&count[i], c);
        }
    }
}

// ---- direct bins: one wavefront per 64 triangles ---------------------------------------------
// Scenes of up to 65536 triangles skip the count / scan / fill passes: every tile owns a
// fixed-capacity slab of 48-byte ENTRIES and the setup kernel appends to it directly, so the
// frame is two launches.  An entry carries everything the raster kernel needs to sweep the
// triangle — the nine projected coordinates, the triangle index and its pixel box — so that a
// tile's workgroup gets its records with ONE dependent load after the list length instead of two
// (index, then a gather of the vertices: 0.9 us of every covered tile's 6 us, in-kernel stamps).
struct __attribute__((aligned(16))) BinEntry {
    float v[9];      // x0 y0 z0 x1 y1 z1 x2 y2 z2 (projected)
    uint32_t id;     // triangle index
    uint32_t bx, by; // pixel box: xl | xr << 16, yt | yb << 16 (.pyx:132-175, rows clipped to the strip)
};
static_assert(sizeof(BinEntry) == 48, "three 16-byte pieces");
constexpr int kEntryPieces = sizeof(BinEntry) / 16;

// ---- heavy tiles (direct bins, 16-pixel tiles) ---------------------------------------------
// A tile whose list reaches kHeavyAt entries is rasterized by two workgroups (upper and lower
// half), from kQuadAt entries on by four (one per 8x8 quadrant), instead of one: a workgroup's
// time grows with the trips its sweep takes and on T-Rex 1024^2 the 105 tiles with >= 32 records
// set the end of the raster launch, 4 us after the median tile (in-kernel timeline, profiles/r02).  The append that crosses kHeavyAt registers
// the tile: it draws an index from the frame's counter (hdr[2 + parity]) and, if one of the
// launch's `hmax` helper triples is still free, raises the tile's flag and writes tile + 1 into
// the triple's three slot words.  k_raster's helper workgroups take parts 1..3 (part 1 alone when
// the list stays below kQuadAt); the tile's own workgroup takes part 0 when the flag is up.  Flag and slots are reset by their
// readers, the counter of the NEXT frame by k_raster.
constexpr uint32_t kHeavyAt = 32;     // lists from here on are split in two halves (16 x 8 pixels),
constexpr uint32_t kQuadAt = 64;      // from here on in four quadrants (8 x 8)
// 32-pixel tiles of a small frame rendered ALONE (the swap chain's plans at depth 1): every covered
// tile goes to four workgroups, one per 16 x 16 quadrant — T-Rex 1024^2 has 312 covered 32-pixel
// tiles, 99 of them with 32 records or more: split by list length the launch keeps two workgroups
// per CU busy and takes 22 us; the 16-pixel plan's launch takes 17.
constexpr uint32_t heavy_at(int ts) { return ts == 32 ? 1u : kHeavyAt; }
constexpr uint32_t quad_at(int ts) { return ts == 32 ? 1u : kQuadAt; }
struct HeavyReg {
    uint32_t *ctr = nullptr;     // this frame's counter; null = no splitting
    uint32_t *flag = nullptr;    // [ntiles]
    uint32_t *slots = nullptr;   // [3 * hmax]
    uint32_t hmax = 0;
    uint32_t heavy_at = kHeavyAt;   // the append that makes a list this long registers the tile
    // dispatch-order hint (build_order): tiles the raster launch will clear in groups without
    // looking at their lists.  The first entry that lands in such a tile declares the hint stale.
    const unsigned char *grouped = nullptr;   // [ntiles], null = the launch is not ordered
    uint32_t *hint_bad = nullptr;
};

CR_DEV void first_entry_of(const HeavyReg &hv, uint32_t tile)
{
    if (hv.grouped && hv.grouped[tile]) *hv.hint_bad = 1u;
}

CR_DEV void register_heavy(const HeavyReg &hv, uint32_t tile)
{
    const uint32_t idx = atomicAdd(hv.ctr, 1u);
    if (idx < hv.hmax) {
        hv.flag[tile] = 1u;
        hv.slots[3 * idx] = tile + 1; hv.slots[3 * idx + 1] = tile + 1; hv.slots[3 * idx + 2] = tile + 1;
    }
}

// The 64 entries of the wavefront as staged in LDS (NP 16-byte pieces each): any lane can write
// any owner's entry.
template <int NP>
CR_DEV void put_entry(float4 *__restrict__ dst, const float4 *img, int owner)
{
    const float4 *src = img + owner * NP;
#pragma unroll
    for (int k = 0; k < NP; ++k) dst[k] = src[k];
}

// Append with one returning global atomic per (triangle, tile) pair — the fallback for a
// wavefront whose triangles span more tiles than its LDS histogram holds (large triangles).
template <int NP>
CR_DEV void bin_direct_append(uint2 r_keep, const float4 *img, int ntx,
                              uint32_t *__restrict__ count, float4 *__restrict__ bins,
                              uint32_t dcap, uint32_t *__restrict__ hdr, const HeavyReg hv)
{
    const int lane = threadIdx.x & 63;
    char *const bin_bytes = reinterpret_cast<char *>(bins);      // (slab <= kDirectBinBytes: 32-bit offsets)
    auto entry_at = [&](uint32_t tile, uint32_t slot) {
        return reinterpret_cast<float4 *>(bin_bytes + (uint32_t)((tile * dcap + slot) * (uint32_t)sizeof(BinEntry)));
    };
    // A lane's returning atomics are independent of each other: kPassC of them are issued, then their
    // entries stored (a loop of "atomic, wait, store" would pay one memory round trip per tile).
    {
        int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
        if (r_keep.x != kNoTiles) {
            tx0 = r_keep.x & 0xFFFF; tx1 = r_keep.x >> 16; ty0 = r_keep.y & 0xFFFF; ty1 = r_keep.y >> 16;
        }
        const int wd = tx1 - tx0 + 1, cnt = wd * (ty1 - ty0 + 1);
        if (cnt <= kWideTiles) {
            constexpr int kPassC = 4;
            int cx = 0, rowbase = ty0 * ntx + tx0;    // tile k of the range, stepped
#pragma unroll 1
            for (int k0 = 0; k0 < cnt; k0 += kPassC) {
                uint32_t slot[kPassC], tile[kPassC];
                bool on[kPassC];
#pragma unroll
                for (int k = 0; k < kPassC; ++k) {
                    tile[k] = (uint32_t)(rowbase + cx);
                    on[k] = k0 + k < cnt;
                    slot[k] = on[k] ? atomicAdd(&count[tile[k]], 1u) : 0u;
                    if (++cx == wd) { cx = 0; rowbase += ntx; }
                }
#pragma unroll
                for (int k = 0; k < kPassC; ++k) {
                    if (on[k]) {
                        if (slot[k] < dcap) put_entry<NP>(entry_at(tile[k], slot[k]), img, lane);
                        else atomicMax(&hdr[1], slot[k] + 1);
                        if (hv.ctr && slot[k] == hv.heavy_at - 1) register_heavy(hv, tile[k]);
                        if (slot[k] == 0) first_entry_of(hv, tile[k]);
                    }
                }
            }
            r_keep.x = kNoTiles;   // done; only wide ranges are left for the cooperative walk
        }
    }
    // Wide ranges: all of the wavefront's wide ranges are flattened into one run of
    // (triangle, tile) pairs and walked 64 x 8 at a time, the round's atomics all in
    // flight before its first store.  (One range after another, each paying its own
    // memory round trip, a wavefront holding 20 large triangles took 40 us: that was the
    // whole binning pass of bunny 4096^2.)
    if (__ballot(r_keep.x != kNoTiles) == 0) return;
    {
        int wd = 0, cnt = 0, sx0 = 0, sy0 = 0;
        if (r_keep.x != kNoTiles) {
            sx0 = r_keep.x & 0xFFFF; sy0 = r_keep.y & 0xFFFF;
            wd = (int)(r_keep.x >> 16) - sx0 + 1;
            cnt = wd * ((int)(r_keep.y >> 16) - sy0 + 1);      // <= kDirectMaxTilesPerTriangle
        }
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int total = __shfl(incl, 63, 64);
        constexpr int kRound = 8;
        for (int base = 0; base < total; base += 64 * kRound) {    // uniform: every lane takes
            const int j0 = base + lane;                             // every trip (shuffles inside)
            uint32_t slot[kRound], tw[kRound];      // tw = tile | owner lane << 20 (direct bins: < 2^16 tiles)
#pragma unroll
            for (int u = 0; u < kRound; ++u) {
                const int j = j0 + 64 * u;
                int own = 0;     // first lane whose inclusive count exceeds j
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1)
                    if (__shfl(incl, own + step - 1, 64) <= j) own += step;
                own &= 63;                                          // (j >= total: unused)
                const int ow = __shfl(wd, own, 64), ocnt = __shfl(cnt, own, 64);
                const int ox0 = __shfl(sx0, own, 64), oy0 = __shfl(sy0, own, 64);
                const int i = j - (__shfl(incl, own, 64) - ocnt);   // tile number within the range
                const int dy = (int)(((float)i + 0.5f) * (1.0f / (float)(ow > 0 ? ow : 1)));  // exact: i < 2^22
                const uint32_t tile = (uint32_t)((oy0 + dy) * ntx + ox0 + (i - dy * ow));
                const bool want = j < total;
                tw[u] = tile | ((uint32_t)own << 20) | (want ? 0x80000000u : 0u);       // (bit 31: entry wanted)
                slot[u] = want ? atomicAdd(&count[tile], 1u) : 0u;
            }
#pragma unroll
            for (int u = 0; u < kRound; ++u) {
                if (tw[u] >> 31) {
                    const uint32_t tile = tw[u] & 0xFFFFFu;
                    if (slot[u] < dcap) put_entry<NP>(entry_at(tile, slot[u]), img, (int)((tw[u] >> 20) & 63u));
                    else atomicMax(&hdr[1], slot[u] + 1);
                    if (hv.ctr && slot[u] == hv.heavy_at - 1) register_heavy(hv, tile);
                    if (slot[u] == 0) first_entry_of(hv, tile);
                }
            }
        }
    }
}

// The 256-thread k_setup above walks a chain of barriers with ceil(T / 256) workgroups: 54 for
// T-Rex, a fifth of the chip's CUs, 11 us per launch of which 2 us were the staging of its inputs
// alone (in-kernel stamps, profiles/r02).  Direct bins need no block-level cooperation, so here a
// workgroup IS one wavefront (its barrier is free): 64 triangles staged through LDS with
// unit-stride float4 loads, projected, culled and boxed; ceil(T / 64) workgroups spread over the
// CUs.  The appends are aggregated per wavefront — consecutive triangles of a mesh land in the
// same few tiles, and one returning global atomic per (triangle, tile) pair serialises on the
// counters of the busy tiles (250 entries on one counter: 2.8 us at the 11 ns one address takes):
//   A  count the wavefront's entries per tile in an LDS histogram over its tile bounding box,
//   B  one returning global atomic per touched tile reserves a run of that tile's slab,
//   C  the entries take consecutive slots of the run (LDS cursors) and are written out.
// A wavefront whose box exceeds the histogram (large triangles) appends pair by pair.
constexpr int kWave = 64;
constexpr int kWaveHistTiles = 512;       // 8 rounds of 64 lanes in pass B
// (the body is a function of its own — one wavefront, lanes = threads 0..63 of the workgroup, LDS
// handed in — so that k_frame can run it beside a raster launch's workgroups)
struct SetupArgs {
    const float *tri_in, *nrm;
    float *proj_out;
    uint32_t *count;
    float4 *bins;
    uint32_t dcap;
    uint32_t *hdr;
    HeavyReg hv;
    int64_t T;
    ProjConst P;
    Geom G;
};
constexpr size_t kSetupWaveLds = sizeof(float4) * kWave * kEntryPieces + sizeof(uint32_t) * kWaveHistTiles;
// The binning wavefront's lanes talk through LDS among themselves only: its "barrier" is the LDS
// queue's own order (a wavefront's LDS operations complete in issue order) made explicit to the
// compiler and to the wait counters — no s_barrier, so that the same code can run as one wavefront
// of a wider workgroup (k_frame) whose other wavefronts have left.
CR_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int TS, bool PROJECT>
CR_DEV void setup_wave_body(const float *__restrict__ tri_in, const float *__restrict__ nrm,
                            float *__restrict__ proj_out, uint32_t *__restrict__ count,
                            float4 *__restrict__ bins, uint32_t dcap, uint32_t *__restrict__ hdr,
                            const HeavyReg hv, int64_t T, const ProjConst &P, const Geom &G,
                            int64_t group, unsigned char *lds)
{
    constexpr int NP = kEntryPieces;                         // 16-byte pieces per entry
    float4 *img = reinterpret_cast<float4 *>(lds);                                   // the wavefront's entries
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + sizeof(float4) * kWave * NP);
    const int lane = threadIdx.x;
    const int64_t b0 = group * kWave;
    const int n = (int)((T - b0) < kWave ? (T - b0) : kWave);
    CR_SETUP_STAMP(0);
    // the histogram is zeroed whole while the inputs are on their way (it used to be zeroed, as far
    // as needed, between the bounding box and pass A: one more wait in the chain)
#pragma unroll
    for (int i = 0; i < kWaveHistTiles / kWave; ++i) hist[i * kWave + lane] = 0;
    // (each lane loads and stores its own 36-byte record — every byte of every line is some lane's —
    // instead of going through an LDS staging buffer: one LDS round trip and a wait less in a kernel
    // that is a chain of waits)
    // only the normals' z components are needed (.pyx:202): three strided loads per lane
    float nz0 = 0.0f, nz1 = 0.0f, nz2 = 0.0f;
    float a[9] = {};
    if (lane < n) {
        load9(tri_in + (b0 + lane) * 9, a);
        const float *nn = nrm + (b0 + lane) * 9;
        nz0 = nn[2]; nz1 = nn[5]; nz2 = nn[8];
    }
    CR_SETUP_STAMP(1);      // inputs requested
    uint2 r = make_uint2(kNoTiles, 0);
    if (lane < n) {
        if (PROJECT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) project_vertex(P, a + 3 * c);
            float *o = proj_out + (b0 + lane) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) o[i] = a[i];
        }
        uint32_t bx = 0, by = 0;
        const TriXYZ t{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]};
        if (!backface(nz0, nz1, nz2)) r = tile_range<TS>(t, G, bx, by);
        if (r.x != kNoTiles) {
            const int ntl = (int)((r.x >> 16) - (r.x & 0xFFFF) + 1) * (int)((r.y >> 16) - (r.y & 0xFFFF) + 1);
            if (ntl > kDirectMaxTilesPerTriangle) {
                atomicMax(&hdr[1], 0xFFFFFFFFu);  // sticky: this scene needs the scan path
                r.x = kNoTiles;
            }
        }

        // this triangle's entry, as every tile of its range will get it
        float4 *e = img + lane * NP;
        e[0] = make_float4(a[0], a[1], a[2], a[3]);
        e[1] = make_float4(a[4], a[5], a[6], a[7]);
        e[2] = make_float4(a[8], __uint_as_float((uint32_t)(b0 + lane)), __uint_as_float(bx), __uint_as_float(by));
    }
    // tile bounding box of the wavefront's ranges
    int X0 = 0x7FFFFFFF, X1 = -1, Y0 = 0x7FFFFFFF, Y1 = -1;
    if (r.x != kNoTiles) {
        X0 = r.x & 0xFFFF; X1 = r.x >> 16; Y0 = r.y & 0xFFFF; Y1 = r.y >> 16;
    }
    wave_box(X0, X1, Y0, Y1);
    wave_lds_sync();        // projected vertices and entries visible to every lane
    CR_SETUP_STAMP(2);      // projected, ranges known
    if (X1 < 0) return;     // nothing to bin (uniform)
    const int bw = X1 - X0 + 1, area = bw * (Y1 - Y0 + 1);
    if (area > kWaveHistTiles) {
        bin_direct_append<NP>(r, img, G.ntx, count, bins, dcap, hdr, hv);
        return;
    }
    for_each_tile_xy(r, [&](int tx, int ty, int) { atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u); });
    wave_lds_sync();
    CR_SETUP_STAMP(3);      // pass A done
    {
        // kWaveHistTiles / kWave rounds of 64 tiles at most, kPassB of them in flight together (all
        // eight at once held 24 registers across the atomics' round trip: with the raster body's
        // budget of 72 that put k_frame's binning wavefronts 40 registers into scratch; most
        // wavefronts have one round, few more than two)
        constexpr int kPassB = 2;
        const float rbw = 1.0f / (float)bw;
        const int nr = (area + kWave - 1) / kWave;      // rounds that have tiles at all (uniform; mostly 1)
#pragma unroll 1
        for (int k0 = 0; k0 < nr; k0 += kPassB) {
            uint32_t c[kPassB], t[kPassB], base[kPassB];
#pragma unroll
            for (int k = 0; k < kPassB; ++k) {
                const int i = (k0 + k) * kWave + lane;
                c[k] = i < area ? hist[i] : 0u;
                const int dy = (int)(((float)i + 0.5f) * rbw);          // exact: i < 2^22
                t[k] = (uint32_t)((Y0 + dy) * G.ntx + X0 + (i - dy * bw));
            }
#pragma unroll
            for (int k = 0; k < kPassB; ++k)
                base[k] = c[k] ? atomicAdd(&count[t[k]], c[k]) : 0u;     // issued together
#pragma unroll
            for (int k = 0; k < kPassB; ++k) {
                if (c[k]) {
                    hist[(k0 + k) * kWave + lane] = base[k];
                    if (base[k] + c[k] > dcap) atomicMax(&hdr[1], base[k] + c[k]);
                    if (hv.ctr && base[k] < hv.heavy_at && base[k] + c[k] >= hv.heavy_at) register_heavy(hv, t[k]);
                    if (base[k] == 0) first_entry_of(hv, t[k]);
                }
            }
        }
    }
    wave_lds_sync();
    CR_SETUP_STAMP(4);      // pass B done (global atomics returned)
    // a lane's own (narrow) range, kPassC tiles at a time: their LDS cursors first, then the entries.
    // (All sixteen at once, with 64-bit addresses, were the register peak of the whole body: 86
    // VGPRs beside k_raster's 68.)  The slab is at most kDirectBinBytes long: 32-bit byte offsets.
    {
        int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
        if (r.x != kNoTiles) {
            tx0 = r.x & 0xFFFF; tx1 = r.x >> 16; ty0 = r.y & 0xFFFF; ty1 = r.y >> 16;
        }
        const int wd = tx1 - tx0 + 1, cnt = wd * (ty1 - ty0 + 1);
        if (cnt <= kWideTiles) {
            constexpr int kPassC = 4;
            int cx = 0, hrow = (ty0 - Y0) * bw + (tx0 - X0), trow = ty0 * G.ntx + tx0;
            const float4 *mine = img + lane * NP;
            char *const bin_bytes = reinterpret_cast<char *>(bins);
#pragma unroll 1
            for (int k0 = 0; k0 < cnt; k0 += kPassC) {
                uint32_t slot[kPassC], tile[kPassC];
#pragma unroll
                for (int k = 0; k < kPassC; ++k) {
                    tile[k] = (uint32_t)(trow + cx);
                    slot[k] = k0 + k < cnt ? atomicAdd(&hist[hrow + cx], 1u) : 0xFFFFFFFFu;
                    if (++cx == wd) { cx = 0; hrow += bw; trow += G.ntx; }
                }
                float4 e[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) e[q] = mine[q];
#pragma unroll
                for (int k = 0; k < kPassC; ++k) {
                    if (slot[k] < dcap) {
                        float4 *dst = reinterpret_cast<float4 *>(bin_bytes + (uint32_t)((tile[k] * dcap + slot[k]) * (uint32_t)sizeof(BinEntry)));
#pragma unroll
                        for (int q = 0; q < NP; ++q) dst[q] = e[q];
                    }
                }
            }
            r.x = kNoTiles;   // done; only wide ranges are left for the cooperative walk
        }
    }
    for_each_tile_xy(r, [&](int tx, int ty, int owner) {
        const uint32_t slot = atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u);
        if (slot < dcap) put_entry<NP>(bins + ((size_t)(ty * G.ntx + tx) * dcap + slot) * NP, img, owner);
    });
    CR_SETUP_STAMP(5);      // entries issued
}

template <int TS, bool PROJECT>
__global__ __launch_bounds__(kWave) void k_setup_wave(const float *__restrict__ tri_in,
                                                      const float *__restrict__ nrm,
                                                      float *__restrict__ proj_out,
                                                      uint32_t *__restrict__ count,
                                                      float4 *__restrict__ bins, uint32_t dcap,
                                                      uint32_t *__restrict__ hdr, HeavyReg hv, int64_t T,
                                                      ProjConst P, Geom G)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kSetupWaveLds];
    setup_wave_body<TS, PROJECT>(tri_in, nrm, proj_out, count, bins, dcap, hdr, hv, T, P, G,
                                 (int64_t)blockIdx.x, lds);
}

// ---- scan path, one wavefront per 64 triangles ------------------------------------------------
// The count and fill passes of the scan path in the shape of k_setup_wave: no block-wide histogram
// (k_setup's 32 KB + 18 KB of LDS hold a CU to three workgroups whose loads, arithmetic and stores
// take turns: 3.2 TB/s on 10 M triangles), ceil(T / 64) independent wavefronts instead, a few KB of
// LDS each, so that a CU always has loads of some of them in flight.  List lengths are aggregated
// per wavefront over its tile bounding box — neighbouring triangles of a mesh, or of a model kept
// in tile-coherent order, share their tiles — and cost one global atomic per touched tile; a
// wavefront whose box exceeds the histogram counts pair by pair.
template <int TS, bool PROJECT>
__global__ __launch_bounds__(kWave) void k_count_wave(const float *__restrict__ tri_in,
                                                      const float *__restrict__ nrm,
                                                      float *__restrict__ proj_out,
                                                      uint2 *__restrict__ trange,
                                                      uint32_t *__restrict__ count, int64_t T,
                                                      ProjConst P, Geom G)
{
    __shared__ __attribute__((aligned(16))) float sv[kWave * 9];
    __shared__ uint32_t hist[kWaveHistTiles];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    const int n = (int)((T - b0) < kWave ? (T - b0) : kWave);
    stage_in<kWave>(tri_in + b0 * 9, sv, n * 9);
#pragma unroll
    for (int i = 0; i < kWaveHistTiles / kWave; ++i) hist[i * kWave + lane] = 0;   // (while the inputs are on their way)
    float nz0 = 0.0f, nz1 = 0.0f, nz2 = 0.0f;      // .pyx:202 looks at the normals' z only
    if (lane < n) {
        const float *nn = nrm + (b0 + lane) * 9;
        nz0 = nn[2]; nz1 = nn[5]; nz2 = nn[8];
    }
    __syncthreads();
    uint2 r = make_uint2(kNoTiles, 0);
    if (lane < n) {
        float *v = sv + lane * 9;
        float a[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] = v[i];
        if (PROJECT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) project_vertex(P, a + 3 * c);
#pragma unroll
            for (int i = 0; i < 9; ++i) v[i] = a[i];
        }
        const TriXYZ t{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]};
        if (!backface(nz0, nz1, nz2)) r = tile_range<TS>(t, G);
        trange[b0 + lane] = r;
    }
    int X0 = 0x7FFFFFFF, X1 = -1, Y0 = 0x7FFFFFFF, Y1 = -1;
    if (r.x != kNoTiles) {
        X0 = r.x & 0xFFFF; X1 = r.x >> 16; Y0 = r.y & 0xFFFF; Y1 = r.y >> 16;
    }
    wave_box(X0, X1, Y0, Y1);
    __syncthreads();
    if (PROJECT) stage_out<kWave>(proj_out + b0 * 9, sv, n * 9);
    if (X1 < 0) return;     // nothing to count (uniform)
    const int bw = X1 - X0 + 1, area = bw * (Y1 - Y0 + 1);
    if (area > kWaveHistTiles) {
        for_each_tile(r, 0u, G.ntx, [&](int tile, uint32_t) { atomicAdd(&count[tile], 1u); });
        return;
    }
    for_each_tile_xy(r, [&](int tx, int ty, int) { atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u); });
    __syncthreads();
    const float rbw = 1.0f / (float)bw;
    for (int i = lane; i < area; i += kWave) {
        const uint32_t c = hist[i];
        const int dy = (int)(((float)i + 0.5f) * rbw);              // exact: i < 2^22
        if (c) atomicAdd(&count[(Y0 + dy) * G.ntx + X0 + (i - dy * bw)], c);
    }
}

// The fill pass in the same shape: (A) the wavefront's entries per tile in LDS, (B) one returning
// atomic per touched tile on the list's cursor reserves a run, (C) LDS cursors hand out its slots.
// trange is read once (k_fill's block histograms need two sweeps).  The pass is a latency chain —
// ranges, then offsets + returning atomics, then the entries: three memory round trips per wavefront,
// 8 192 wavefronts resident — so a wavefront takes kFillPer x 64 triangles through each round trip
// together (10 M triangles: 97 -> 78 us with two, 77 with four; profiles/r03/ab_fill_groups.txt).
constexpr int kFillPer = 2;
__global__ __launch_bounds__(kWave) void k_fill_wave(const uint2 *__restrict__ trange,
                                                     const uint32_t *__restrict__ offs,
                                                     uint32_t *__restrict__ cursor,
                                                     uint32_t *__restrict__ entries,
                                                     uint32_t capacity, int64_t T, Geom G)
{
    __shared__ uint32_t hist[kWaveHistTiles];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * (kWave * kFillPer);
    uint2 r[kFillPer];
#pragma unroll
    for (int p = 0; p < kFillPer; ++p)
        r[p] = (b0 + p * kWave + lane < T) ? trange[b0 + p * kWave + lane] : make_uint2(kNoTiles, 0);
#pragma unroll
    for (int i = 0; i < kWaveHistTiles / kWave; ++i) hist[i * kWave + lane] = 0;   // (while the ranges are on their way)
    int X0 = 0x7FFFFFFF, X1 = -1, Y0 = 0x7FFFFFFF, Y1 = -1;
#pragma unroll
    for (int p = 0; p < kFillPer; ++p) {
        if (r[p].x != kNoTiles) {
            const int x0 = r[p].x & 0xFFFF, x1 = r[p].x >> 16, y0 = r[p].y & 0xFFFF, y1 = r[p].y >> 16;
            X0 = x0 < X0 ? x0 : X0; X1 = x1 > X1 ? x1 : X1; Y0 = y0 < Y0 ? y0 : Y0; Y1 = y1 > Y1 ? y1 : Y1;
        }
    }
    wave_box(X0, X1, Y0, Y1);
    if (X1 < 0) return;
    const int bw = X1 - X0 + 1, area = bw * (Y1 - Y0 + 1);
    if (area > kWaveHistTiles) {
#pragma unroll
        for (int p = 0; p < kFillPer; ++p)
            for_each_tile(r[p], (uint32_t)(b0 + p * kWave + lane), G.ntx, [&](int tile, uint32_t id) {
                const uint32_t pos = offs[tile] + atomicAdd(&cursor[tile], 1u);
                if (pos < capacity) entries[pos] = id;
            });
        return;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kFillPer; ++p)
        for_each_tile_xy(r[p], [&](int tx, int ty, int) { atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u); });
    __syncthreads();
    {
        constexpr int kRounds = kWaveHistTiles / kWave;
        const float rbw = 1.0f / (float)bw;
        uint32_t c[kRounds], t[kRounds], base[kRounds];
        const int nr = (area + kWave - 1) / kWave;      // rounds that have tiles at all (uniform; mostly 1)
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            c[k] = 0u; t[k] = 0u;
            if (k < nr) {
                const int i = k * kWave + lane;
                c[k] = i < area ? hist[i] : 0u;
                const int dy = (int)(((float)i + 0.5f) * rbw);          // exact: i < 2^22
                t[k] = (uint32_t)((Y0 + dy) * G.ntx + X0 + (i - dy * bw));
            }
        }
#pragma unroll
        for (int k = 0; k < kRounds; ++k)                            // all in flight together
            base[k] = c[k] ? offs[t[k]] + atomicAdd(&cursor[t[k]], c[k]) : 0u;
#pragma unroll
        for (int k = 0; k < kRounds; ++k)
            if (c[k]) hist[k * kWave + lane] = base[k];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kFillPer; ++p)
        for_each_tile_xy(r[p], [&](int tx, int ty, int owner) {
            const uint32_t pos = atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u);
            if (pos < capacity) entries[pos] = (uint32_t)(b0 + p * kWave + owner);
        });
}

// Exclusive scan of count[0..ntiles) into offs[0..ntiles]; count is zeroed (k_fill uses
// it as the per-tile cursor); hdr[0] / hdr[4] = low / high word of the number of list entries
// this frame needs.  The running sum is kept in 64 bits and the offsets saturate at 2^32 - 1, so
// a frame that needs more entries than 32 bits can index reports an unsatisfiable figure instead
// of a wrapped one (k_raster clamps every range to the capacity: such tiles come out empty).
// One 1024-thread workgroup walks the array in coalesced slabs of 4096 counters (4
// consecutive ones per thread), the next slab's loads in flight while the current one is
// scanned with wavefront shuffles + one LDS exchange of the 16 wavefront totals.
__global__ __launch_bounds__(1024) void k_scan(uint32_t *__restrict__ count,
                                               uint32_t *__restrict__ offs,
                                               uint32_t *__restrict__ hdr, int ntiles)
{
    __shared__ unsigned long long wave_total[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto load4 = [&](int base, uint32_t c[4]) {
        const int i = base + tid * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = (i + k < ntiles) ? count[i + k] : 0u;
    };
    auto sat = [](unsigned long long v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; };
    uint32_t cur[4], nxt[4];
    load4(0, cur);
    unsigned long long carry = 0;
    int buf = 0;
    for (int base = 0; base < ntiles; base += 4096, buf ^= 1) {
        if (base + 4096 < ntiles) load4(base + 4096, nxt);
        const unsigned long long s = (unsigned long long)cur[0] + cur[1] + cur[2] + cur[3];
        unsigned long long incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_total[buf][wave] = incl;
        __syncthreads();   // (the other buffer is free: its readers passed the previous barrier)
        unsigned long long before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned long long t = wave_total[buf][w];
            if (w < wave) before += t;
            total += t;
        }
        unsigned long long run = carry + before + incl - s;
        const int i = base + tid * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < ntiles) {
                offs[i + k] = sat(run);
                count[i + k] = 0;
            }
            run += cur[k];
        }
        carry += total;
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
    if (tid == 0) {
        offs[ntiles] = sat(carry);
        hdr[0] = (uint32_t)carry;
        hdr[4] = (uint32_t)(carry >> 32);
    }
}

// dynamic LDS: [cur: ntiles u32] when LDS_HIST
template <bool LDS_HIST>
__global__ __launch_bounds__(kThreads) void k_fill(const uint2 *__restrict__ trange,
                                                   const uint32_t *__restrict__ offs,
                                                   uint32_t *__restrict__ cursor,
                                                   uint32_t *__restrict__ entries,
                                                   uint32_t capacity, int64_t T, int64_t chunk,
                                                   Geom G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *cur = reinterpret_cast<uint32_t *>(smem_raw);
    const int64_t c0 = (int64_t)blockIdx.x * chunk;
    const int64_t c1 = (c0 + chunk < T) ? (c0 + chunk) : T;
    const uint2 none = make_uint2(kNoTiles, 0);
    constexpr int U = 4;   // tile ranges in flight per thread: the loop is latency-bound
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < G.ntiles; i += kThreads) cur[i] = 0;
        __syncthreads();
        // sweep 1: how many entries this block adds to each tile list
        for (int64_t b0 = c0; b0 < c1; b0 += (int64_t)U * kThreads) {
            uint2 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = b0 + (int64_t)u * kThreads + threadIdx.x;
                r[u] = t < c1 ? trange[t] : none;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                for_each_tile(r[u], 0u, G.ntx, [&](int tile, uint32_t) { atomicAdd(&cur[tile], 1u); });
        }
        __syncthreads();
        // reserve a contiguous run in every touched list
        for (int i = threadIdx.x; i < G.ntiles; i += kThreads) {
            const uint32_t c = cur[i];
            if (c) cur[i] = offs[i] + atomicAdd(&cursor[i], c);
        }
        __syncthreads();
    }
    for (int64_t b0 = c0; b0 < c1; b0 += (int64_t)U * kThreads) {
        uint2 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = b0 + (int64_t)u * kThreads + threadIdx.x;
            r[u] = t < c1 ? trange[t] : none;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = b0 + (int64_t)u * kThreads + threadIdx.x;
            for_each_tile(r[u], (uint32_t)t, G.ntx, [&](int tile, uint32_t id) {
                uint32_t pos;
                if (LDS_HIST) pos = atomicAdd(&cur[tile], 1u);
                else pos = offs[tile] + atomicAdd(&cursor[tile], 1u);
                if (pos < capacity) entries[pos] = id;
            });
        }
    }
}

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

// Lower an LDS depth key (order-independent: the final key is the minimum over all fragments).
CR_DEV void lds_key_min(unsigned long long *slot, unsigned long long k)
{
    // No pre-read of the key: a non-returning ds_min_u64 does not stall the wavefront, whereas
    // "load, compare, then maybe atomic" puts two dependent LDS round trips on every trip's
    // critical path (T-Rex 1024^2 raster 24.2 -> 23.6 us).
    __hip_atomic_fetch_min(slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Inclusive sum over the 64 lanes of a wavefront with DPP row shifts and row broadcasts (the
// shape LLVM's atomic optimizer emits on gfx9): six vector adds, no LDS.  A scan built from
// __shfl_up is six ds_bpermute round trips in a dependent chain, and the batch's scan sits on
// every covered tile's critical path.
CR_DEV uint32_t wave_incl_sum(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return (uint32_t)x;
}

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
CR_DEV uint32_t packed_wh(uint32_t b) { return ((b >> 12) & 0x7Fu) | ((b >> 19) << 16); }
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
    const uint32_t *entries;    // scan path: triangle indices
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
    int nhelp;                  // 3 * hmax helper workgroups
    // dispatch order (see build_order): null when the launch is not ordered
    int addr32;                  // framebuffer and attribute byte offsets fit 32 bits (see elem())
    const uint32_t *order, *hint, *hint_bad;
    uint32_t *order_next, *hint_next, *hint_bad_next;
    unsigned char *grouped_next;
    int vec_clear;              // planes 16-byte aligned and W % 4 == 0
    Light light;                // CRENDER_FUSED_GURO: illumination applied as pixels are stored
};

// One record of a tile's list: projected vertices, triangle index, pixel box.  false = a stale
// index (beyond the frame's triangle count): no work.
CR_DEV bool load_record(const TileLists &L, const float *__restrict__ proj, const Geom &G, uint32_t idx,
                        uint32_t &id, TriXYZ &t, uint32_t &ebx, uint32_t &eby)
{
    if (L.offs) {
        id = L.entries[idx];
        if (id >= L.T) return false;
        t = load_tri(proj + (size_t)id * 9);
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
    return id < L.T;
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
constexpr int kOrderMaxTiles = 8192;   // the builder keeps one byte per tile in the 13 KB of the batch queue
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
template <bool CLEAR, typename I>
CR_DEV void owner_tile(const WorkQueue &q, const float *pre, int nrec, const float *__restrict__ col, const float *__restrict__ nrm,
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
            if (__any(live)) {                                   // wavefront-uniform
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

// the batch: records array-of-structures on 16-pixel tiles (Rec16), else the WorkQueue
template <int TS>
constexpr size_t raster_queue_bytes()
{
    return TS == 16 ? sizeof(Rec16) * kBatch16 + sizeof(uint32_t) * (kThreads + 8) : sizeof(WorkQueue);
}

// Workgroup `b` of a raster launch (the kernels below hand in their LDS: k_frame runs binning
// wavefronts of another frame in the same launch).
template <int TS, bool CLEAR>
CR_DEV void raster_body(const float *__restrict__ proj, const float *__restrict__ col,
                        const float *__restrict__ nrm, const TileLists &L,
                        float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
                        int32_t *__restrict__ win, const Geom &G, int dbg_arg, int b,
                        unsigned long long *key, unsigned char *qraw)
{
#ifdef CRENDER_STAMPS
    // frames of a swap chain stamp into a region of their slot (bits 24..26 of dbg_arg), 8192 workgroups each
    const size_t stamp_base = ((size_t)((dbg_arg >> 24) & 7) * 8192 + blockIdx.x) * 16;
#endif
    WorkQueue &q = *reinterpret_cast<WorkQueue *>(qraw);                 // (TS != 16 only)
    Rec16 *recs = reinterpret_cast<Rec16 *>(qraw);                       // (TS == 16 only)
    uint32_t *scan16 = reinterpret_cast<uint32_t *>(qraw + sizeof(Rec16) * kBatch16);
    uint32_t *wave16 = scan16 + kThreads;
    constexpr int kBatch = TS == 16 ? kBatch16 : kThreads;
#ifdef CRENDER_DEV_KNOBS
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- which tile, and which part of it --------------------------------------------------
    // grid = [order builder, if ordered][3 * hmax helpers][ntiles main workgroups, one tile each]
    if (L.order_next) {
        if (b == 0) {
            build_order(L.count, G.ntx, G.nty, L.order_next, L.grouped_next, L.hint_next, reinterpret_cast<uint32_t *>(qraw), group_tiles(TS));
            return;
        }
        b -= 1;
    }
    const bool helper = b < L.nhelp;
    int quad = -1;               // -1 = the whole tile, 0..3 = one part of a heavy tile (half or quadrant)
    int tile;
    if (helper) {
        // part 1..3 of the heavy tile registered in this workgroup's slot, if any
        const uint32_t v = L.heavy_slots[b];
        if (v == 0) return;                        // (same word for every thread: uniform)
        tile = (int)v - 1;
        quad = 1 + b % 3;
    } else {
        const int m = b - L.nhelp;
        if (m == 0 && tid == 0) {
            if (L.heavy_ctr_next) *L.heavy_ctr_next = 0;
            *L.hint_bad_next = 0;
        }
        if (L.order && L.hint[0] && !*L.hint_bad) {
            const int ns = (int)L.hint[1], ng = (int)L.hint[2];
            if (m < ns) {
                tile = (int)L.order[m];
            } else {
                // the order's last section: up to kGroup empty tiles per workgroup, cleared with two
                // float4 stores per thread and tile (no list to look at: the binning pass vouches
                // for their emptiness, see first_entry_of)
                if (m >= ns + ng) return;
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
                return;
            }
        } else {
            tile = (dbg & 8) ? xcd_band_tile(m, G.ntiles) : m;
            // Large grids: scatter the dispatch order (block b -> tile b * stride mod ntiles) so
            // that a band of covered tiles is spread over the whole launch instead of arriving
            // together (T-Rex 8192^2: 0.446 -> 0.402 ms).  Small grids are faster in raster order
            // (T-Rex 1024^2: 24.7 vs 29.1 us), so the scatter starts at 32768 tiles.
            if ((G.ntiles >= 32768) != ((dbg & 256) != 0)) {
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
            if (!(dbg & 512)) {
                const int ty = G.ntx_magic ? (int)__umulhi((uint32_t)tile, G.ntx_magic) : tile / G.ntx;
                const int t = tile - ty * G.ntx + 9 * ty;
                const int tx = G.ntx_magic ? t - (int)__umulhi((uint32_t)t, G.ntx_magic) * G.ntx : t % G.ntx;
                tile = ty * G.ntx + tx;
            }
        }
    }
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
    int rw = TS;                 // width of this workgroup's rectangle in the key plane's terms
    if (quad >= 0) {
        constexpr int HS = TS / 2;
        if (end - beg >= quad_at(TS)) {         // four quadrants
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
    } else {
    // 16-pixel tiles with direct bins (at most 65536 triangles): a depth key's low word carries
    // the triangle index in its high half as usual and, in its low half, where the record sits
    // in LDS — batch and slot — so that the resolve takes the winner's edge constants from there
    const bool slotted = TS == 16 && !L.offs;
    // first batch of the tile's list straight into registers
    uint32_t cur_id = 0, cur_bx = 0, cur_by = 0;
    TriXYZ cur_t{};
    bool cur_ok = tid < kBatch && beg + tid < end;
    if (cur_ok) cur_ok = load_record(L, proj, G, beg + tid, cur_id, cur_t, cur_bx, cur_by);
    if (cur_ok && L.orig_of) cur_id = L.orig_of[cur_id];      // from here on: the caller's index

    // depth keys of the tile: the prior buffer value (or the cleared value) per pixel
    // (32-pixel tiles: once it is known that the tile is not the pixel owners', see below)
    auto init_keys = [&]() {
        const unsigned long long key_clear = make_key(zord(1e6f), KEY_LOW_PRIOR);
        for (int p = tid; p < TS * TS; p += kThreads) {
            unsigned long long k = key_clear;
            if (!CLEAR) {
                const int x = X0 + (p % TS), y = Y0 + (p / TS);
                if (x < X1 && y < Y1) k = make_key(zord_prior(zb[(size_t)y * G.W + x]), KEY_LOW_PRIOR);
            }
            key[p] = k;
        }
    };
    if constexpr (TS != 32) init_keys();
    CR_STAMP(1);
#ifdef CRENDER_STAMPS
    if (g_stamps && tid == 0) {
        g_stamps[stamp_base + 4] = end - beg;
        g_stamps[stamp_base + 9] = (unsigned long long)(quad + 1);
    }
#endif

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
            // Short batch on a 16-pixel tile: one PIXEL per thread, every thread walks the
            // records (LDS broadcast reads), the running minimum stays in a register — no
            // block scan, no record search, no LDS atomics.  A wavefront (4 rows of the tile)
            // skips a record whose box misses its rows.
            const uint32_t left = end - base;
            const uint32_t pix_max = (dbg >> 16) & 0xFF ? (uint32_t)((dbg >> 16) & 0xFF) - 1u : kPixelPathRecords;
            if (left <= pix_max) {
                if (base != beg) __syncthreads();
                if (tid < (int)left) put_rec16(&recs[tid], cur_t, key_low, box_xy, box_wh);
                __syncthreads();
#ifdef CRENDER_STAMPS
                if (base == beg) CR_STAMP(6);
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
                    if (!__any(in)) continue;
                    const Rec16Regs R = load_rec16(&recs[r]);
                    unsigned long long k;
                    if (in && fragment16(R, px, py, k) && k < best) best = k;
                }
                key[tid] = best;
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
            q.box[tid] = pack_box(box_xy, box_wh, X0, Y0);
            if constexpr (!either) q.big.blk_scan[tid] = incl - my_blocks;     // (32-pixel tiles: once the sweep is chosen)
            if (lane == 63) q.wave_blocks[wave] = incl;
            if constexpr (either) {
                if (lane == 63) q.wave_px[wave] = incl_px;
            }
        }
        __syncthreads();  // queue complete
#ifdef CRENDER_STAMPS
        if (base == beg) CR_STAMP(6);
#endif

        // ---- sweep: the batch's work items, flattened and split evenly -------------------------
        {
            const int l = tid & 15, lx = l & 3, ly = l >> 2;
            uint32_t wo[kThreads / 64 + 1];
            wo[0] = 0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) wo[w + 1] = wo[w] + (TS == 16 ? wave16[w] : q.wave_blocks[w]);
            const int total = (int)wo[kThreads / 64];
            const int nrec = (int)((end - base) < (uint32_t)kBatch ? (end - base) : (uint32_t)kBatch);
            const uint32_t blk_excl = incl - my_blocks;
            bool small_by_pixel = false;    // 32-pixel tiles: small records go per pixel too
            if constexpr (either) small_by_pixel = total < 16 * nrec && !(dbg & 8192);
            if constexpr (either) {
                if (small_by_pixel) {
                    // every record's thread works out, ONCE, what an item of its record would otherwise
                    // work out again (9 items of two pixels per record on the 10 M small triangles:
                    // 40 of an item's 175 vector instructions)
                    const TriSetup mine = make_setup(cur_t, true);
                    q.pre.l03[tid] = mine.l03; q.pre.l13[tid] = mine.l13; q.pre.l23[tid] = mine.l23;
                    q.pre.r1[tid] = mine.fast ? mine.r1 : 0.0f; q.pre.r2[tid] = mine.r2; q.pre.r3[tid] = mine.r3;
                    q.pre.px_scan[tid] = incl_px - my_px;
                    __syncthreads();
                }
            }
            // next batch: issue its loads now, they complete under the sweeps
            const uint32_t nxt = base + kBatch + tid;
            cur_ok = tid < kBatch && nxt < end;
            if (cur_ok) cur_ok = load_record(L, proj, G, nxt, cur_id, cur_t, cur_bx, cur_by);
            if (cur_ok && L.orig_of) cur_id = L.orig_of[cur_id];
            // Per-pixel sweep: every pixel of every clipped box is one work item; thread t takes
            // items t, t + 256, ...  All lanes work on a sample that lies in its box (a 4x4 block
            // of a small box is mostly empty: 71 % of T-Rex 1024^2's block lanes were inside their
            // box, 40-50 % on its busiest tiles, 35 % for the 10 M small triangles), and there is
            // no per-group record walk.  The item's record comes from a two-level search of the
            // prefix sums (most batches fit the first wavefront's 64 slots: then no wavefront
            // selection and only log2 of the record count steps).
            auto sweep_pixels = [&](const uint32_t *scan, const uint32_t *wo_, int total_) {
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
                        unsigned long long *kp = &key[(y - Y0) * TS + (x - X0)];
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
                        unsigned long long *kp = &key[(y - Y0) * TS + (x - X0)];
#pragma unroll
                        for (int j = 0; j < kItemPixels32; ++j) {
                            float n1, n2, n3;
                            numerators(st, x + j, y, n1, n2, n3);
                            unsigned long long k;
                            if (fragment_from(st, id, n1, n2, n3, true, k) && (j == 0 || px0 + j < bw)) lds_key_min(kp + j, k);
                        }
                    }
                }
            };
            if constexpr (TS == 32) {
                // the whole list is this one batch of large records: the pixels' owners take it from here
                if (base == beg && end - beg <= (uint32_t)kBatch && !small_by_pixel && !(dbg & 32768)) {
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
                        const TriSetup st = make_setup(TriXYZ{q.x0[tid], q.y0[tid], q.z0[tid], q.x1[tid], q.y1[tid], q.z1[tid],
                                                               q.x2[tid], q.y2[tid], q.z2[tid]}, true);
                        const uint32_t bwh = packed_wh(q.box[tid]), bxy = packed_xy(q.box[tid], X0, Y0);
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
                        owner_tile<CLEAR, uint32_t>(q, pre, nrec, col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                                    G.W, X0, Y0, X1, Y1);
                    else
                        owner_tile<CLEAR, size_t>(q, pre, nrec, col, nrm, L.pos_of, L.light, zb, cb, nb, win,
                                                  G.W, X0, Y0, X1, Y1);
                    CR_STAMP(3);
                    return;
                }
                if (base == beg) {
                    init_keys();
                    __syncthreads();
                }
            }
            if constexpr (per_pixel) {
                sweep_pixels(scan16, wo, total);
            } else if (small_by_pixel) {
                uint32_t wop[kThreads / 64 + 1];
                wop[0] = 0;
#pragma unroll
                for (int w = 0; w < kThreads / 64; ++w) wop[w + 1] = wop[w] + q.wave_px[w];
                sweep_pixels(q.pre.px_scan, wop, (int)wop[kThreads / 64]);
            } else if (TS == 64 && ((total < 16 * nrec && !(dbg & 8192)) || (dbg & 128))) {
                // Small records: each of the 16 lane groups takes one contiguous run of blocks,
                // so a record is set up by (almost) one group only; tight loop, plain division.
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
                            lds_key_min(&key[(y - Y0) * TS + (x - X0)], k);
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
            } else {
                // Large records (>= 16 blocks on average).  A triangle fills at most half of its
                // pixel box, so first a coarse pass (one LANE per 4x4 block) discards blocks that
                // lie entirely outside one edge; the survivors are then swept (one 16-lane GROUP
                // per block): each wavefront takes a contiguous quarter of them and its four
                // groups consecutive survivors, so the four blocks a wavefront works on at a
                // time are neighbours.  In the sweep, lanes whose sign test is certainly negative
                // are dead before any division (skipped wave-wide when nobody is live,
                // raster_math.h (1)); the divisions that remain use the hoisted reciprocal (2).
                const bool allow_rej = !(dbg & 32), allow_fast = !(dbg & 64);
                if constexpr (TS <= 32) {          // a record has at most 64 blocks: one mask word
                    q.big.mask[tid] = 0;
                    if constexpr (either) q.big.blk_scan[tid] = blk_excl;
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
                                lds_key_min(&key[(y - Y0) * TS + (x - X0)], k);
                            if (++p >= pend) break;   // (p < pend guarantees another survivor)
                            m &= m - 1;
                            if (m == 0) {
                                do { m = q.big.mask[++r]; } while (m == 0);
                                wk = load_work<TriSetup>(q, r, X0, Y0);
                                inv_nbx = 1.0f / (float)wk.nbx;
                            }
                        }
                    }
                } else {
                    // 64-pixel tiles (up to 256 blocks per record): no cull, dense walk
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
                            if (__any(live)) {   // wavefront-uniform
                                unsigned long long k;
                                if (live && fragment_from(wk.s, wk.id, n1, n2, n3, allow_fast, k))
                                    lds_key_min(&key[(y - Y0) * TS + (x - X0)], k);
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
            }
        }
    }
    __syncthreads();

    CR_STAMP(2);
    // resolve: every pixel of the rectangle is written at most once (exactly once if CLEAR).
    // A part's pixels are taken by the first wavefronts in rows of its own width.
    const int npx = quad >= 0 ? rw * (TS / 2) : TS * TS;
    auto resolve = [&](auto index_tag) {
    using I = decltype(index_tag);
    for (int p0 = tid; p0 < npx; p0 += kThreads) {
        const int dy = rw == TS ? p0 / TS : p0 / (TS / 2), dx = p0 - dy * rw;
        const int x = X0 + dx, y = Y0 + dy;
        if (x >= X1 || y >= Y1) continue;
        const int p = dy * TS + dx;
        const I pix = (I)((I)y * (I)G.W + (I)x);
        const uint32_t low = (uint32_t)key[p];
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
        shade_and_store(proj, col, nrm, L.pos_of ? L.pos_of[id] : id, x, y, pix, zb, cb, nb, L.light);
        if (win) *reinterpret_cast<int32_t *>(elem(reinterpret_cast<float *>(win), pix)) = (int32_t)id;
    }
    };
    if (L.addr32) resolve(uint32_t{}); else resolve(size_t{});
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

template <int TS, bool CLEAR>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(TS == 16 ? kWavesPerSimd16 : TS == 32 ? kWavesPerSimd32 : 1)))
void k_raster(const float *__restrict__ proj, const float *__restrict__ col,
              const float *__restrict__ nrm, TileLists L,
              float *__restrict__ zb, float *__restrict__ cb, float *__restrict__ nb,
              int32_t *__restrict__ win, Geom G, int dbg_arg)
{
    __shared__ __attribute__((aligned(16))) unsigned long long key[TS * TS];   // (the pixel owners read it as float4)
    __shared__ __attribute__((aligned(16))) unsigned char qraw[raster_queue_bytes<TS>()];
    raster_body<TS, CLEAR>(proj, col, nrm, L, zb, cb, nb, win, G, dbg_arg, (int)blockIdx.x, key, qraw);
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
static_assert(kSetupWaveLds <= raster_queue_bytes<16>() && kSetupWaveLds <= raster_queue_bytes<32>(),
              "a binning wavefront works in the raster workgroup's batch queue");
struct FrameArgs {
    RasterArgs R;
    SetupArgs S;
    int nsetup;
};
template <int TS, bool CLEAR>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(TS == 16 ? kWavesPerSimd16 : TS == 32 ? kWavesPerSimd32 : 1)))
void k_frame(FrameArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned long long key[TS * TS];   // (the pixel owners read it as float4)
    __shared__ __attribute__((aligned(16))) unsigned char qraw[raster_queue_bytes<TS>()];
    if ((int)blockIdx.x < A.nsetup) {
        if (threadIdx.x < kWave)
            setup_wave_body<TS, true>(A.S.tri_in, A.S.nrm, A.S.proj_out, A.S.count, A.S.bins, A.S.dcap, A.S.hdr,
                                      A.S.hv, A.S.T, A.S.P, A.S.G, (int64_t)blockIdx.x, qraw);
        return;
    }
    raster_body<TS, CLEAR>(A.R.proj, A.R.col, A.R.nrm, A.R.L, A.R.zb, A.R.cb, A.R.nb, A.R.win, A.R.G, A.R.dbg,
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

// Sort key of a triangle for the tile-coherent order: Morton code of the 32-pixel tile its
// projected centroid falls in (an ordering heuristic only: nothing exact depends on it).
__global__ __launch_bounds__(kThreads) void k_tile_order_keys(const float *__restrict__ tri, int64_t T,
                                                              ProjConst P, int W, int H,
                                                              uint32_t *__restrict__ keys)
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
        uint32_t a = (uint32_t)x >> 5, b = (uint32_t)y >> 5, m = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) m |= ((a >> k) & 1u) << (2 * k) | ((b >> k) & 1u) << (2 * k + 1);
        keys[t] = m;
    }
}

// ---- host side --------------------------------------------------------------------
constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

int pick_tile(int H, int W, int tile)
{
    if (tile == 16 || tile == 32 || tile == 64) return tile;
    // measured on MI355X (profiles/r01): small frames are latency-bound per tile and want many
    // small tiles; large frames want 32-pixel tiles (full 128-B rows of z, 7 workgroups per CU)
    return ((int64_t)H * W <= (int64_t)1024 * 1024) ? 16 : 32;
}

struct Layout {
    int ts;
    Geom g;
    int64_t max_T;
    int64_t capacity;
    int64_t direct_cap;   // entries per tile of the direct bins, 0 = scene too large for them
    int hmax;             // helper triples of a raster launch (heavy tiles split in four), 0 = none
    bool ordered;         // raster launches leave a dispatch order for the next one (build_order)
    size_t count_stride;  // u32 words between the two parities of the per-tile counters
    size_t off_hdr, off_count, off_hflag, off_hslots, off_hint, off_order, off_grouped, off_offs,
           off_trange, off_proj, off_entries, off_direct, total;
};
constexpr int kMaxHeavyHelped32 = 512;
constexpr int kMaxHeavyHelped = 128;   // +384 workgroups per raster launch (9 % at 1024 x 1024)

// Direct bins are for small scenes (the README benchmark): one launch fewer than the
// count / scan / fill path matters when a frame takes tens of microseconds.
constexpr int64_t kDirectMaxTriangles = 1 << 16;
constexpr int kDirectMaxTiles = 1 << 16;    // beyond: count / scan / fill
constexpr int64_t kDirectBinBytes = 512ll << 20;   // per-tile capacity = this budget / 48 B / tiles, <= 1024

bool make_layout(int H, int W, int y0, int y1, int64_t max_T, int64_t cap, int tile, Layout &L)
{
    if (H <= 0 || W <= 0 || y0 < 0 || y1 > H || y0 >= y1 || max_T < 0) return false;
    if (W > 65535 || H > 65535) return false;             // pixel boxes are packed in 16 bits
    if (max_T > 0xFFFFFFF0ll) return false;               // triangle index lives in 32 key bits
    L.ts = pick_tile(H, W, tile);
    L.g.W = W; L.g.H = H; L.g.y0 = y0; L.g.y1 = y1;
    L.g.ntx = (W + L.ts - 1) / L.ts;
    L.g.nty = (y1 - y0 + L.ts - 1) / L.ts;
    L.g.ntiles = L.g.ntx * L.g.nty;
    {   // a stride near ntiles / golden ratio, made coprime to ntiles
        auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
        int k = (int)(L.g.ntiles * 0.6180339887) | 1;
        while (k > 1 && gcd(k, L.g.ntiles) != 1) k += 2;
        L.g.tile_stride = k < 1 ? 1 : k;
        // both dividends of k_raster's tile decode stay below ntiles + 9 * nty
        const unsigned long long lim = ((unsigned long long)L.g.ntiles + 9ull * (unsigned)L.g.nty + L.g.ntx) * (unsigned)L.g.ntx;
        L.g.ntx_magic = (lim < (1ull << 32) && L.g.ntx > 1) ? (uint32_t)((1ull << 32) / (unsigned)L.g.ntx) + 1u : 0u;
        L.g.inv_ntiles = 1.0 / (double)L.g.ntiles;
    }
    L.max_T = max_T;
    if (cap <= 0) cap = 4 * max_T + 4 * (int64_t)L.g.ntiles + 65536;
    if (cap > 0xFFFFFFF0ll) cap = 0xFFFFFFF0ll;
    L.capacity = cap;
    L.direct_cap = 0;
    if (max_T <= kDirectMaxTriangles && L.g.ntiles <= kDirectMaxTiles) {
        L.direct_cap = kDirectBinBytes / (int64_t)sizeof(BinEntry) / L.g.ntiles;
        if (L.direct_cap > 1024) L.direct_cap = 1024;
    }
    // heavy tiles are split with direct bins only (the small-frame regime, where a single tile's
    // latency sets the end of the launch): on 16-pixel tiles and — for frames rendered alone on the
    // swap chain's 32-pixel plans, whose 300 covered workgroups would leave most of 256 CUs idle — on
    // 32-pixel tiles of small frames, where most covered tiles are heavy
    L.hmax = 0;
    if (L.direct_cap >= 2 * (int64_t)kHeavyAt) {
        if (L.ts == 16) L.hmax = L.g.ntiles / 8 < kMaxHeavyHelped ? L.g.ntiles / 8 : kMaxHeavyHelped;
        else if (L.ts == 32 && L.g.ntiles <= 2048) L.hmax = L.g.ntiles / 2 < kMaxHeavyHelped32 ? L.g.ntiles / 2 : kMaxHeavyHelped32;
    }
    size_t o = 0;
    // [header | counters, parity 0 and 1 | heavy flags | heavy slots | order hints] are zeroed at creation
    L.off_hdr = o;     o = align_up(o + 64);
    L.count_stride = align_up(sizeof(uint32_t) * (size_t)(L.g.ntiles + 1)) / sizeof(uint32_t);
    L.off_count = o;   o = o + 2 * L.count_stride * sizeof(uint32_t);
    L.off_hflag = o;   o = align_up(o + sizeof(uint32_t) * (size_t)L.g.ntiles);
    L.off_hslots = o;  o = align_up(o + sizeof(uint32_t) * 3 * (size_t)(L.hmax > 0 ? L.hmax : 1));
    // small frames are dispatched in the order the previous frame suggests (build_order)
    L.ordered = (L.ts == 16 || (L.ts == 32 && L.g.ntiles <= 2048)) && L.direct_cap > 0 && L.g.ntiles <= kOrderMaxTiles;
    L.off_hint = o;    o = align_up(o + sizeof(uint32_t) * 8);                      // two headers of 4 words
    L.off_order = o;   o = align_up(o + sizeof(uint32_t) * 2 * (size_t)(L.ordered ? L.g.ntiles : 0));
    L.off_grouped = o; o = align_up(o + 2 * (size_t)(L.ordered ? L.g.ntiles : 0));
    L.off_offs = o;    o = align_up(o + sizeof(uint32_t) * (size_t)(L.g.ntiles + 1));
    L.off_trange = o;  o = align_up(o + sizeof(uint2) * (size_t)max_T);
    L.off_proj = o;    o = align_up(o + sizeof(float) * 9 * (size_t)max_T);
    L.off_entries = o; o = align_up(o + sizeof(uint32_t) * (size_t)cap);
    L.off_direct = o;  o = align_up(o + sizeof(BinEntry) * (size_t)L.g.ntiles * (size_t)L.direct_cap);
    L.total = o;
    return true;
}

}  // namespace

struct crender_plan {
    Layout L;
    unsigned char *ws;
    int stamp_slot = 0;        // (diagnostic build: which region of the stamp buffer its raster launches use)
    // optional per-frame HIP events (crender_plan_timing_begin): 3 per frame —
    // before the binning passes, before k_raster, after k_raster
    std::vector<hipEvent_t> events;
    int timed_frames = 0;
    bool direct_ok = true;        // cleared once a frame overflowed the direct bins
    bool last_frame_direct = false;
    int64_t last_T = -1;          // triangle count of the last bin pass (crender_draw must match)
    // The per-tile counters exist twice.  Frame f bins into parity f & 1 and its raster pass
    // zeroes the OTHER parity for frame f + 1, so no raster workgroup ever writes a counter that
    // another workgroup of the same launch reads (the four workgroups of a heavy tile all read
    // its count).  awaiting[p]: parity p was binned into and not zeroed since.
    unsigned frame_no = 0;
    int parity = 0;               // of the last bin pass
    bool awaiting[2] = {false, false};
    uint32_t *count(int par) const { return reinterpret_cast<uint32_t *>(ws + L.off_count) + (size_t)par * L.count_stride; }
    uint32_t *hflag() const { return reinterpret_cast<uint32_t *>(ws + L.off_hflag); }
    uint32_t *hslots() const { return reinterpret_cast<uint32_t *>(ws + L.off_hslots); }
    int hint_par = 0;             // order / hint buffer the next raster launch reads (it writes the other)
    float light[3] = {0.f, 0.f, 0.f};   // crender_plan_set_light (CRENDER_FUSED_GURO)
    const uint32_t *orig_of = nullptr, *pos_of = nullptr;   // crender_plan_set_triangle_order
    bool frame_lone = true;       // the last bin pass belonged to a frame rendered for latency (no
                                  // CRENDER_OVERLAPPED_FRAMES): ordered dispatch and split heavy tiles
    uint32_t *hint(int k) const { return reinterpret_cast<uint32_t *>(ws + L.off_hint) + 4 * k; }
    uint32_t *order(int k) const { return reinterpret_cast<uint32_t *>(ws + L.off_order) + (size_t)k * L.g.ntiles; }
    unsigned char *grouped(int k) const { return ws + L.off_grouped + (size_t)k * L.g.ntiles; }
    float4 *direct() const { return reinterpret_cast<float4 *>(ws + L.off_direct); }
    bool timing() const { return !events.empty() && (size_t)(timed_frames + 1) * 3 <= events.size(); }
    hipEvent_t ev(int k) const { return events[(size_t)timed_frames * 3 + k]; }
    uint32_t *hdr() const { return reinterpret_cast<uint32_t *>(ws + L.off_hdr); }
    uint32_t *offs() const { return reinterpret_cast<uint32_t *>(ws + L.off_offs); }
    uint2 *trange() const { return reinterpret_cast<uint2 *>(ws + L.off_trange); }
    float *proj() const { return reinterpret_cast<float *>(ws + L.off_proj); }
    uint32_t *entries() const { return reinterpret_cast<uint32_t *>(ws + L.off_entries); }
};

// Swap chain of `depth` (crender_pipeline_*): frame i runs entirely on the pipeline's stream
// i % depth with plan i % depth into the framebuffer set the caller passes for it; frames in
// flight target DIFFERENT framebuffer sets, so nothing orders them and they overlap freely on the GPU.  No HIP
// event sits between frames: on MI355X / ROCm 7.2 an event record + cross-stream wait opens a
// 7-12 us bubble (rocprofv3 timeline, profiles/r01), a third of a 1024^2 frame.
constexpr int kMaxPipelineDepth = 8;
struct crender_pipeline {
    int depth = 0;
    crender_plan *plan[kMaxPipelineDepth] = {};
    hipStream_t s[kMaxPipelineDepth] = {};
    hipEvent_t done[kMaxPipelineDepth] = {};
    hipEvent_t mark = nullptr;
    // look-ahead (crender_pipeline_set_lookahead): slot k alternates between plan[k] and ahead[k];
    // the launch that rasterizes one of them bins the slot's NEXT frame into the other (k_frame)
    crender_plan *ahead[kMaxPipelineDepth] = {};
    int sel[kMaxPipelineDepth] = {};           // 0: plan[k] holds / takes the current frame, 1: ahead[k]
    struct Primed {                            // what the slot's other plan has been binned for
        bool ok = false;
        const float *tri = nullptr, *nrm = nullptr;
        int64_t T = 0;
        float P[16] = {};
        unsigned flags = 0;
    } primed[kMaxPipelineDepth];
    uint64_t n = 0;           // frames submitted since the last join
    // optional HIP events around every frame's launches on its own stream
    // (crender_pipeline_timing_begin): 2 per frame
    std::vector<hipEvent_t> events;
    int timed_frames = 0;
    bool timing() const { return (size_t)(timed_frames + 1) * 2 <= events.size(); }
    const void *last_tri = nullptr, *last_nrm = nullptr;
    int64_t last_T = -1;
    hipStream_t last_caller = nullptr;
    bool synced = false;
    struct Bound {   // crender_pipeline_bind
        bool set = false, has_P = false;
        const float *tri = nullptr, *col = nullptr, *nrm = nullptr;
        int64_t T = 0;
        float P[16] = {};
        float *z = nullptr, *color = nullptr, *normal = nullptr;
        int32_t *winner = nullptr;
        unsigned flags = 0;
    } bound[kMaxPipelineDepth];
};

namespace {

int grid_for(size_t items, int cap)
{
    size_t b = (items + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = (size_t)cap;
    return (int)b;
}

// LDS histograms up to this many tiles: k_setup packs 16-bit counters (32 KiB + 18 KiB of
// staging), k_fill needs 32-bit cursors (64 KiB, the dynamic-LDS limit is raised for it).
constexpr int kMaxLdsHistTiles = 16384;
constexpr int64_t kWaveScanBelow = 1 << 18;   // (the filler keeps larger models tile-coherent)

// Frame = bin pass (K1 + binning into the plan) + raster pass (K2 from the plan's bins).
int dev_knobs()
{
#ifdef CRENDER_DEV_KNOBS
    static const int dbg = std::getenv("CRENDER_DEBUG") ? std::atoi(std::getenv("CRENDER_DEBUG")) : 0;
    return dbg;
#else
    return 0;
#endif
}

// `defer` (crender_pipeline's look-ahead): when the pass is the one-launch direct-bin kernel its
// arguments are handed back instead of launched, for k_frame to run it inside a raster launch.
template <int TS>
int run_bin_pass(crender_plan *plan, bool project, const float *d_tri, const float *d_nrm, int64_t T,
                 const ProjConst &P, unsigned flags, hipStream_t s, SetupArgs *defer = nullptr,
                 bool *deferred = nullptr)
{
    const Layout &L = plan->L;
    const Geom G = L.g;
    const int dbg = dev_knobs();
    const bool direct = L.direct_cap > 0 && plan->direct_ok && !(flags & CRENDER_NO_DIRECT_BINS) &&
                        !(dbg & 16);
    plan->last_frame_direct = direct;
    plan->last_T = T;
    const int par = (int)(plan->frame_no++ & 1u);
    plan->parity = par;
    plan->frame_lone = !(flags & CRENDER_OVERLAPPED_FRAMES) || (dbg & 16384);
    if (plan->awaiting[par]) {
        // this parity was binned into and no raster pass has run since (two crender_prepare calls
        // in a row): start over from the state crender_plan_create leaves
        CR_HIP(hipMemsetAsync(plan->ws + L.off_count, 0, L.off_order - L.off_count, s));
        CR_HIP(hipMemsetAsync(plan->hdr() + 2, 0, 5 * sizeof(uint32_t), s));   // heavy counters, hint_bad
        plan->awaiting[0] = plan->awaiting[1] = false;
    }
    plan->awaiting[par] = true;
    uint32_t *count = plan->count(par);

    // contiguous chunk of triangles per block, a multiple of the block size
    auto chunking = [T](int64_t max_blocks, int64_t &nblk, int64_t &chunk) {
        nblk = (T + kThreads - 1) / kThreads;
        if (nblk > max_blocks) nblk = max_blocks;
        chunk = (T + nblk - 1) / nblk;
        chunk = (chunk + kThreads - 1) / kThreads * kThreads;
        nblk = (T + chunk - 1) / chunk;
    };
    // block-private LDS histograms pay off when a block's chunk is dense in tiles; a small
    // scene on a large tile grid would only zero and flush mostly empty histograms
    const bool lds_hist = G.ntiles <= 4096 || (G.ntiles <= kMaxLdsHistTiles && T >= 16 * (int64_t)G.ntiles);
    // Scan path: one wavefront per 64 triangles (k_count_wave / k_fill_wave) where neighbouring
    // triangles can be expected to share tiles — a mesh, or a large model kept in tile-coherent
    // order; a large triangle soup in arbitrary order keeps the block histograms.
    const bool wave_scan = (plan->orig_of != nullptr || T < kWaveScanBelow) && !(dbg & 4);
    if (T > 0 && direct) {
        // direct bins: one wavefront per 64 triangles
        HeavyReg hv;
        if (L.hmax > 0 && plan->frame_lone && !(dbg & 2048)) {
            hv.ctr = plan->hdr() + 2 + par; hv.flag = plan->hflag(); hv.slots = plan->hslots();
            hv.hmax = (uint32_t)L.hmax;
            hv.heavy_at = heavy_at(TS);
        }
        if (L.ordered) {
            hv.grouped = plan->grouped(plan->hint_par);    // of the order this frame's raster pass reads
            hv.hint_bad = plan->hdr() + 5 + par;
        }
        const unsigned nwg = (unsigned)((T + kWave - 1) / kWave);
        if (defer && project && TS <= 32) {
            *defer = SetupArgs{d_tri, d_nrm, plan->proj(), count, plan->direct(), (uint32_t)L.direct_cap,
                               plan->hdr(), hv, T, P, G};
            *deferred = true;
            return CRENDER_OK;
        }
        if (project)
            hipLaunchKernelGGL((k_setup_wave<TS, true>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm,
                               plan->proj(), count, plan->direct(), (uint32_t)L.direct_cap, plan->hdr(),
                               hv, T, P, G);
        else
            hipLaunchKernelGGL((k_setup_wave<TS, false>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm,
                               plan->proj(), count, plan->direct(), (uint32_t)L.direct_cap, plan->hdr(),
                               hv, T, P, G);
        CR_LAUNCH_CHECK("k_setup_wave");
    } else if (T > 0 && wave_scan) {
        const unsigned nwg = (unsigned)((T + kWave - 1) / kWave);
        if (project)
            hipLaunchKernelGGL((k_count_wave<TS, true>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm,
                               plan->proj(), plan->trange(), count, T, P, G);
        else
            hipLaunchKernelGGL((k_count_wave<TS, false>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm,
                               plan->proj(), plan->trange(), count, T, P, G);
        CR_LAUNCH_CHECK("k_count_wave");
    } else if (T > 0) {
        int64_t nblk, chunk;
        chunking(2048, nblk, chunk);
        const size_t hist_bytes = lds_hist ? sizeof(uint32_t) * (size_t)((((G.ntiles + 1) >> 1) + 3) & ~3) : 0;
        while (lds_hist && chunk > 65280) {   // 16-bit block-local counters
            nblk *= 2;
            chunk = ((T + nblk - 1) / nblk + kThreads - 1) / kThreads * kThreads;
            nblk = (T + chunk - 1) / chunk;
        }
        const size_t setup_lds = hist_bytes + sizeof(float) * kThreads * 9 * 2;
#define CR_SETUP(PROJ, BIN)                                                                          \
    hipLaunchKernelGGL((k_setup<TS, PROJ, BIN>), dim3((unsigned)nblk), dim3(kThreads), setup_lds, s, \
                       d_tri, d_nrm, plan->proj(), plan->trange(), count, T, chunk, P, G)
        if (project) {
            if (lds_hist) CR_SETUP(true, kBinCountLds);
            else CR_SETUP(true, kBinCountGlobal);
        } else {
            if (lds_hist) CR_SETUP(false, kBinCountLds);
            else CR_SETUP(false, kBinCountGlobal);
        }
#undef CR_SETUP
        CR_LAUNCH_CHECK("k_setup");
    }
    if (!direct) {
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, count, plan->offs(), plan->hdr(),
                           G.ntiles);
        CR_LAUNCH_CHECK("k_scan");
        if (T > 0) {
            int64_t nblk, chunk;
            chunking(1024, nblk, chunk);
            const size_t lds = lds_hist ? sizeof(uint32_t) * (size_t)G.ntiles : 0;
            if (!wave_scan && lds_hist && lds > 48 * 1024) {
                static const hipError_t attr = hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&k_fill<true>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                if (attr != hipSuccess) return fail_hip(attr, "hipFuncSetAttribute(k_fill)");
            }
            if (wave_scan)
                hipLaunchKernelGGL(k_fill_wave, dim3((unsigned)((T + kWave * kFillPer - 1) / (kWave * kFillPer))), dim3(kWave), 0, s,
                                   plan->trange(), plan->offs(), count, plan->entries(),
                                   (uint32_t)L.capacity, T, G);
            else if (lds_hist)
                hipLaunchKernelGGL((k_fill<true>), dim3((unsigned)nblk), dim3(kThreads), lds, s,
                                   plan->trange(), plan->offs(), count, plan->entries(),
                                   (uint32_t)L.capacity, T, chunk, G);
            else
                hipLaunchKernelGGL((k_fill<false>), dim3((unsigned)nblk), dim3(kThreads), 0, s,
                                   plan->trange(), plan->offs(), count, plan->entries(),
                                   (uint32_t)L.capacity, T, chunk, G);
            CR_LAUNCH_CHECK("k_fill");
        }
    }
    return CRENDER_OK;
}

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
    const bool direct = plan->last_frame_direct;
    const int par = plan->parity;
    TileLists tl;
    tl.offs = direct ? nullptr : plan->offs();
    tl.count = plan->count(par);
    tl.count_next = plan->count(par ^ 1);
    tl.entries = plan->entries();
    tl.bins = plan->direct();
    tl.capacity = direct ? (uint32_t)L.direct_cap : (uint32_t)L.capacity;
    tl.T = (uint32_t)plan->last_T;
    tl.orig_of = plan->orig_of;
    tl.pos_of = plan->pos_of;
    const bool split = direct && L.hmax > 0 && plan->frame_lone && !(dbg & 2048);
    tl.heavy_flag = split ? plan->hflag() : nullptr;
    tl.heavy_slots = split ? plan->hslots() : nullptr;
    tl.heavy_ctr_next = plan->hdr() + 2 + (par ^ 1);
    tl.nhelp = split ? 3 * L.hmax : 0;
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
    const uintptr_t any = (uintptr_t)d_z | (uintptr_t)d_color | (uintptr_t)d_normal | (uintptr_t)d_winner;
    tl.vec_clear = (any & 15u) == 0 && (G.W & 3) == 0;
    tl.light = Light{plan->light[0], plan->light[1], plan->light[2], (flags & CRENDER_FUSED_GURO) ? 1 : 0};
    const unsigned grid = (unsigned)(G.ntiles + tl.nhelp + (ordered ? 1 : 0));
    if constexpr (TS <= 32) {
        if (with_setup) {
            // this frame's raster pass and another plan's binning pass in one launch (k_frame)
            const int nsetup = (int)((with_setup->T + kWave - 1) / kWave);
            const RasterArgs ra{proj, d_col, d_nrm, tl, d_z, d_color, d_normal, d_winner, G, dbg};
            const FrameArgs fa{ra, *with_setup, nsetup};
            if (flags & CRENDER_FUSED_CLEAR)
                hipLaunchKernelGGL((k_frame<TS, true>), dim3(grid + (unsigned)nsetup), dim3(kThreads), 0, s, fa);
            else
                hipLaunchKernelGGL((k_frame<TS, false>), dim3(grid + (unsigned)nsetup), dim3(kThreads), 0, s, fa);
            CR_LAUNCH_CHECK("k_frame");
            plan->awaiting[par ^ 1] = false;
            return CRENDER_OK;
        }
    }
    if (flags & CRENDER_FUSED_CLEAR)
        hipLaunchKernelGGL((k_raster<TS, true>), dim3(grid), dim3(kThreads), 0, s, proj, d_col, d_nrm, tl,
                           d_z, d_color, d_normal, d_winner, G, dbg);
    else
        hipLaunchKernelGGL((k_raster<TS, false>), dim3(grid), dim3(kThreads), 0, s, proj, d_col, d_nrm, tl,
                           d_z, d_color, d_normal, d_winner, G, dbg);
    CR_LAUNCH_CHECK("k_raster");
    plan->awaiting[par ^ 1] = false;     // zeroed by this launch
    return CRENDER_OK;
}

#define CR_BY_TILE(call16, call32, call64) \
    (plan->L.ts == 16 ? (call16) : plan->L.ts == 32 ? (call32) : (call64))

int bin_pass(crender_plan *plan, bool project, const float *d_tri, const float *d_nrm, int64_t T,
             const float *P16, unsigned flags, void *stream, SetupArgs *defer = nullptr,
             bool *deferred = nullptr)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    if (T < 0 || T > plan->L.max_T) return fail(CRENDER_EINVAL, "T exceeds the plan's max_T");
    if (T > 0 && (!d_tri || !d_nrm)) return fail(CRENDER_EINVAL, "null triangle array");
    if (project && !P16) return fail(CRENDER_EINVAL, "null projection matrix");
    ProjConst P;
    std::memset(&P, 0, sizeof P);
    if (project) P = make_proj(P16, plan->L.g.W, plan->L.g.H);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return CR_BY_TILE(run_bin_pass<16>(plan, project, d_tri, d_nrm, T, P, flags, s, defer, deferred),
                      run_bin_pass<32>(plan, project, d_tri, d_nrm, T, P, flags, s, defer, deferred),
                      run_bin_pass<64>(plan, project, d_tri, d_nrm, T, P, flags, s, defer, deferred));
}

int raster_pass(crender_plan *plan, const float *proj, const float *d_col, const float *d_nrm, int64_t T,
                float *d_z, float *d_color, float *d_normal, int32_t *d_winner, unsigned flags,
                void *stream, const SetupArgs *with_setup = nullptr)
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

int tile_frame(crender_plan *plan, bool project, const float *d_tri, const float *d_col,
               const float *d_nrm, int64_t T, const float *P16, float *d_z, float *d_color,
               float *d_normal, int32_t *d_winner, unsigned flags, void *stream)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    if (!d_z || !d_color || !d_normal) return fail(CRENDER_EINVAL, "null framebuffer pointer");
    if (T > 0 && !d_col) return fail(CRENDER_EINVAL, "null triangle array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool timing = plan->timing();
    if (timing) CR_HIP(hipEventRecord(plan->ev(0), s));
    int rc = bin_pass(plan, project, d_tri, d_nrm, T, P16, flags, stream);
    if (rc != CRENDER_OK) return rc;
    if (timing) CR_HIP(hipEventRecord(plan->ev(1), s));
    rc = raster_pass(plan, project ? plan->proj() : d_tri, d_col, d_nrm, T, d_z, d_color, d_normal,
                     d_winner, flags, stream);
    if (rc != CRENDER_OK) return rc;
    if (timing) {
        CR_HIP(hipEventRecord(plan->ev(2), s));
        plan->timed_frames++;
    }
    return CRENDER_OK;
}

}  // namespace

// =========================== C ABI ================================================
extern "C" {
#ifdef CRENDER_STAMPS
CRENDER_API int crender_debug_set_stamps(void *d_buf);
CRENDER_API int crender_debug_set_setup_stamps(void *d_buf);
#endif

int crender_abi_version(void) { return CRENDER_ABI_VERSION; }

const char *crender_last_error(void) { return g_last_error.c_str(); }

int crender_projection_matrix(double fov_deg, double z_near, double z_far, int h, int w, float *P16)
{
    if (!P16 || h <= 0 || w <= 0) return fail(CRENDER_EINVAL, "crender_projection_matrix: bad argument");
    // .pyx:54-59: fov -> C float; f evaluated in double from that float, stored as float
    const float fovf = (float)fov_deg;
    const float f = (float)(1.0 / std::tan((double)fovf / 2 / 180 * M_PI));
    const float zn = (float)z_near, zf = (float)z_far;
    const float a = (float)((double)h / (double)w);
    // .pyx:83-90: q and f/a in float arithmetic; -z_near*q = exact double product of two
    // floats rounded once to float32
    const float q = zf / (zf - zn);
    std::memset(P16, 0, 16 * sizeof(float));
    P16[0] = f / a;
    P16[5] = f;
    P16[10] = q;
    P16[11] = 1.0f;
    P16[14] = (float)((double)(-zn) * (double)q);
    return CRENDER_OK;
}

int crender_project(const float *d_tri_in, float *d_tri_out, int64_t T, const float *P16, int w,
                    int h, void *stream)
{
    if (T < 0 || !P16 || w <= 0 || h <= 0) return fail(CRENDER_EINVAL, "crender_project: bad argument");
    if (T == 0) return CRENDER_OK;
    if (!d_tri_in || !d_tri_out) return fail(CRENDER_EINVAL, "crender_project: null array");
    const ProjConst P = make_proj(P16, w, h);
    hipLaunchKernelGGL(k_project, dim3(grid_for((size_t)T, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_tri_in, d_tri_out, T, P);
    CR_LAUNCH_CHECK("k_project");
    return CRENDER_OK;
}

int crender_clear(float *d_z, float *d_color, float *d_normal, int32_t *d_winner, int H, int W, int y0,
                  int y1, void *stream)
{
    if (!d_z || !d_color || !d_normal || H <= 0 || W <= 0 || y0 < 0 || y1 > H || y0 >= y1)
        return fail(CRENDER_EINVAL, "crender_clear: bad argument");
    const size_t first = (size_t)y0 * W, npix = (size_t)(y1 - y0) * W;
    hipLaunchKernelGGL(k_clear, dim3(grid_for(npix * 3, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_z, d_color, d_normal, d_winner, first, npix);
    CR_LAUNCH_CHECK("k_clear");
    return CRENDER_OK;
}

size_t crender_plan_workspace_bytes(int H, int W, int y0, int y1, int64_t max_T, int64_t bin_capacity,
                                    int tile)
{
    Layout L;
    if (!make_layout(H, W, y0, y1, max_T, bin_capacity, tile, L)) return 0;
    return L.total;
}

int crender_plan_create(crender_plan **out, int H, int W, int y0, int y1, int64_t max_T,
                        int64_t bin_capacity, int tile, void *d_workspace, size_t workspace_bytes,
                        void *stream)
{
    if (!out) return fail(CRENDER_EINVAL, "crender_plan_create: null out");
    *out = nullptr;
    Layout L;
    if (!make_layout(H, W, y0, y1, max_T, bin_capacity, tile, L))
        return fail(CRENDER_EINVAL, "crender_plan_create: bad geometry (need 0 <= y0 < y1 <= H, "
                                    "H, W <= 65535, max_T >= 0)");
    if (!d_workspace) return fail(CRENDER_EINVAL, "crender_plan_create: null workspace");
    if (((uintptr_t)d_workspace & (kAlign - 1)) != 0)
        return fail(CRENDER_EINVAL, "crender_plan_create: workspace must be 256-byte aligned");
    if (workspace_bytes < L.total) return fail(CRENDER_ENOMEM, "crender_plan_create: workspace too small");
    crender_plan *p = new (std::nothrow) crender_plan();
    if (!p) return fail(CRENDER_ENOMEM, "crender_plan_create: host allocation failed");
    p->L = L;
    p->ws = static_cast<unsigned char *>(d_workspace);
    // header + per-tile counters start at zero; the frame kernels keep them zero
    hipError_t e = hipMemsetAsync(p->ws, 0, L.off_offs, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        delete p;
        return fail_hip(e, "hipMemsetAsync(workspace)");
    }
    *out = p;
    return CRENDER_OK;
}

void crender_plan_destroy(crender_plan *plan)
{
    if (!plan) return;
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    delete plan;
}

int crender_plan_timing_begin(crender_plan *plan, int max_frames)
{
    if (!plan || max_frames < 0) return fail(CRENDER_EINVAL, "crender_plan_timing_begin: bad argument");
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    plan->events.clear();
    plan->timed_frames = 0;
    plan->events.reserve((size_t)max_frames * 3);
    for (int i = 0; i < max_frames * 3; ++i) {
        hipEvent_t e;
        CR_HIP(hipEventCreate(&e));
        plan->events.push_back(e);
    }
    return CRENDER_OK;
}

int crender_plan_timing_end(crender_plan *plan, void *stream, int *frames, double *bin_ms_avg,
                            double *raster_ms_avg)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    CR_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    double bin = 0.0, ras = 0.0;
    const int n = plan->timed_frames;
    for (int f = 0; f < n; ++f) {
        float a = 0.f, b = 0.f;
        CR_HIP(hipEventElapsedTime(&a, plan->events[(size_t)f * 3], plan->events[(size_t)f * 3 + 1]));
        CR_HIP(hipEventElapsedTime(&b, plan->events[(size_t)f * 3 + 1], plan->events[(size_t)f * 3 + 2]));
        bin += a;
        ras += b;
    }
    if (frames) *frames = n;
    if (bin_ms_avg) *bin_ms_avg = n ? bin / n : 0.0;
    if (raster_ms_avg) *raster_ms_avg = n ? ras / n : 0.0;
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    plan->events.clear();
    plan->timed_frames = 0;
    return CRENDER_OK;
}

int crender_plan_last_bin_usage(crender_plan *plan, void *stream, int64_t *needed, int64_t *capacity)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    uint32_t h[5] = {0, 0, 0, 0, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    CR_HIP(hipMemcpyAsync(h, plan->hdr(), sizeof h, hipMemcpyDeviceToHost, s));
    CR_HIP(hipStreamSynchronize(s));
    if (plan->last_frame_direct) {
        // direct bins: per-tile figures.  h[1] is sticky: the longest list that did not fit
        // (0xFFFFFFFF = a triangle spans too many tiles).  On overflow this plan switches to
        // the count / scan / fill path for good; the caller renders the frame again.
        const int64_t cap = plan->L.direct_cap;
        if (h[1] > (uint32_t)cap) plan->direct_ok = false;  // h[1] stays set: the answer is repeatable
        if (needed) *needed = h[1] > (uint32_t)cap ? (int64_t)h[1] : 0;
        if (capacity) *capacity = cap;
        return CRENDER_OK;
    }
    if (needed) *needed = (int64_t)(((unsigned long long)h[4] << 32) | h[0]);
    if (capacity) *capacity = plan->L.capacity;
    return CRENDER_OK;
}

int crender_plan_set_triangle_order(crender_plan *plan, const uint32_t *d_orig_of, const uint32_t *d_pos_of)
{
    if (!plan || ((d_orig_of == nullptr) != (d_pos_of == nullptr)))
        return fail(CRENDER_EINVAL, "crender_plan_set_triangle_order: both arrays or neither");
    plan->orig_of = d_orig_of;
    plan->pos_of = d_pos_of;
    return CRENDER_OK;
}

int crender_plan_set_light(crender_plan *plan, const float *light3)
{
    if (!plan || !light3) return fail(CRENDER_EINVAL, "crender_plan_set_light: bad argument");
    plan->light[0] = light3[0]; plan->light[1] = light3[1]; plan->light[2] = light3[2];
    return CRENDER_OK;
}

int crender_plan_last_frame_direct(crender_plan *plan)
{
    return plan && plan->last_frame_direct ? 1 : 0;
}

int crender_raster(crender_plan *plan, const float *d_tri_proj, const float *d_col, const float *d_nrm,
                   int64_t T, float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                   unsigned flags, void *stream)
{
    return tile_frame(plan, false, d_tri_proj, d_col, d_nrm, T, nullptr, d_z, d_color, d_normal,
                      d_winner, flags, stream);
}

int crender_prepare(crender_plan *plan, const float *d_tri, const float *d_nrm, int64_t T,
                    const float *P16, unsigned flags, void *stream)
{
    return bin_pass(plan, P16 != nullptr, d_tri, d_nrm, T, P16, flags, stream);
}

int crender_draw(crender_plan *plan, const float *d_tri_proj, const float *d_col, const float *d_nrm,
                 int64_t T, float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                 unsigned flags, void *stream)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    return raster_pass(plan, d_tri_proj ? d_tri_proj : plan->proj(), d_col, d_nrm, T, d_z, d_color,
                       d_normal, d_winner, flags, stream);
}

int crender_render_model(crender_plan *plan, const float *d_tri, const float *d_col, const float *d_nrm,
                         int64_t T, const float *P16, float *d_z, float *d_color, float *d_normal,
                         int32_t *d_winner, unsigned flags, void *stream)
{
    return tile_frame(plan, true, d_tri, d_col, d_nrm, T, P16, d_z, d_color, d_normal, d_winner, flags,
                      stream);
}

#ifdef CRENDER_STAMPS
int crender_debug_set_setup_stamps(void *d_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(d_buf);
    CR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_setup_stamps), &p, sizeof p));
    return CRENDER_OK;
}
// diagnostic build: point the kernels at a stamp buffer (ntiles * 8 u64), or detach with null
int crender_debug_set_stamps(void *d_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(d_buf);
    CR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p));
    return CRENDER_OK;
}
#endif

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

static int crender_render_model_on(crender_plan *plan, const float *d_tri, const float *d_col,
                                   const float *d_nrm, int64_t T, const float *P16, float *d_z,
                                   float *d_color, float *d_normal, int32_t *d_winner, unsigned flags,
                                   hipStream_t s)
{
    return tile_frame(plan, P16 != nullptr, d_tri, d_col, d_nrm, T, P16, d_z, d_color, d_normal,
                      d_winner, flags, s);
}

int crender_pipeline_create(crender_pipeline **out, crender_plan *const *plans, int depth)
{
    if (!out || !plans || depth < 1 || depth > kMaxPipelineDepth)
        return fail(CRENDER_EINVAL, "crender_pipeline_create: need 1..8 plans");
    for (int i = 0; i < depth; ++i)
        for (int j = 0; j <= i; ++j)
            if (!plans[i] || (j < i && plans[i] == plans[j]))
                return fail(CRENDER_EINVAL, "crender_pipeline_create: plans must be distinct");
    *out = nullptr;
    crender_pipeline *p = new (std::nothrow) crender_pipeline();
    if (!p) return fail(CRENDER_ENOMEM, "crender_pipeline_create: host allocation failed");
    p->depth = depth;
    hipError_t e = hipSuccess;
    for (int k = 0; k < depth && e == hipSuccess; ++k) {
        p->plan[k] = plans[k];
        plans[k]->stamp_slot = k;
        e = hipStreamCreateWithFlags(&p->s[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->mark, hipEventDisableTiming);
    if (e != hipSuccess) {
        crender_pipeline_destroy(p);
        return fail_hip(e, "crender_pipeline_create");
    }
    *out = p;
    return CRENDER_OK;
}

void crender_pipeline_destroy(crender_pipeline *p)
{
    if (!p) return;
    for (int k = 0; k < p->depth; ++k) {
        if (p->s[k]) (void)hipStreamSynchronize(p->s[k]);
        if (p->done[k]) (void)hipEventDestroy(p->done[k]);
        if (p->s[k]) (void)hipStreamDestroy(p->s[k]);
    }
    if (p->mark) (void)hipEventDestroy(p->mark);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    delete p;
}

int crender_pipeline_frame(crender_pipeline *p, const float *d_tri, const float *d_col,
                           const float *d_nrm, int64_t T, const float *P16, float *d_z, float *d_color,
                           float *d_normal, int32_t *d_winner, unsigned flags, void *stream)
{
    if (!p) return fail(CRENDER_EINVAL, "null pipeline");
    hipStream_t caller = static_cast<hipStream_t>(stream);
    if (!p->synced || d_tri != p->last_tri || d_nrm != p->last_nrm || T != p->last_T ||
        caller != p->last_caller) {
        // new inputs, or the first frame after a join: whatever produced the inputs, and whatever
        // touched the framebuffers last, was enqueued on the caller's stream
        // (an event record + cross-stream wait opens a 7-12 us bubble in each queue: skipped when the
        // caller's stream has nothing pending, e.g. right after the caller synchronised)
        if (hipStreamQuery(caller) != hipSuccess) {
            CR_HIP(hipEventRecord(p->mark, caller));
            for (int k = 0; k < p->depth; ++k) CR_HIP(hipStreamWaitEvent(p->s[k], p->mark, 0));
        }
        p->last_tri = d_tri; p->last_nrm = d_nrm; p->last_T = T; p->last_caller = caller;
        p->synced = true;
    }
    const int k = (int)(p->n % (uint64_t)p->depth);
    const bool timed = p->timing();
    if (timed) CR_HIP(hipEventRecord(p->events[(size_t)p->timed_frames * 2], p->s[k]));
    struct Stamp {      // the closing event, whichever way the frame leaves this function with success
        crender_pipeline *p; int k; bool on;
        int close() {
            if (!on) return CRENDER_OK;
            on = false;
            CR_HIP(hipEventRecord(p->events[(size_t)p->timed_frames * 2 + 1], p->s[k]));
            p->timed_frames++;
            return CRENDER_OK;
        }
    } stamp{p, k, timed};
    if (p->ahead[k] && P16 && T > 0) {
        // One launch per frame: this frame's raster pass together with the binning pass of the
        // slot's NEXT frame — expected to come with the same inputs — into the slot's other plan.
        // A frame whose plan was not binned for exactly these inputs (the first frames after a
        // join, after new inputs) bins first, in a launch of its own.
        if (!d_z || !d_color || !d_normal) return fail(CRENDER_EINVAL, "null framebuffer pointer");
        if (!d_col) return fail(CRENDER_EINVAL, "null triangle array");
        crender_plan *cur = p->sel[k] ? p->ahead[k] : p->plan[k];
        crender_plan *nxt = p->sel[k] ? p->plan[k] : p->ahead[k];
        crender_pipeline::Primed &pr = p->primed[k];
        const bool have = pr.ok && pr.tri == d_tri && pr.nrm == d_nrm && pr.T == T && pr.flags == flags &&
                          std::memcmp(pr.P, P16, sizeof pr.P) == 0;
        pr.ok = false;
        int rc = CRENDER_OK;
        if (!have) rc = bin_pass(cur, true, d_tri, d_nrm, T, P16, flags, p->s[k]);
        if (rc != CRENDER_OK) return rc;
        SetupArgs sa;
        bool deferred = false;
        rc = bin_pass(nxt, true, d_tri, d_nrm, T, P16, flags, p->s[k], &sa, &deferred);
        if (rc != CRENDER_OK) return rc;
        rc = raster_pass(cur, cur->proj(), d_col, d_nrm, T, d_z, d_color, d_normal, d_winner, flags, p->s[k],
                         deferred ? &sa : nullptr);
        if (rc != CRENDER_OK) return rc;
        pr.ok = true; pr.tri = d_tri; pr.nrm = d_nrm; pr.T = T; pr.flags = flags;
        std::memcpy(pr.P, P16, sizeof pr.P);
        p->sel[k] ^= 1;
        p->n++;
        return stamp.close();
    }
    // plan k and framebuffer set k were last used by frame n - depth, earlier on this same stream
    // (a frame that cannot look ahead — no triangles, projected input — takes the slot's first plan
    // whatever was binned ahead: that is void then)
    p->primed[k].ok = false;
    p->sel[k] = 0;
    int rc = crender_render_model_on(p->plan[k], d_tri, d_col, d_nrm, T, P16, d_z, d_color, d_normal,
                                     d_winner, flags, p->s[k]);
    if (rc != CRENDER_OK) return rc;
    p->n++;
    return stamp.close();
}

int crender_pipeline_timing_begin(crender_pipeline *p, int max_frames)
{
    if (!p || max_frames < 0) return fail(CRENDER_EINVAL, "crender_pipeline_timing_begin: bad argument");
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    p->timed_frames = 0;
    p->events.reserve((size_t)max_frames * 2);
    for (int i = 0; i < max_frames * 2; ++i) {
        hipEvent_t e;
        CR_HIP(hipEventCreate(&e));
        p->events.push_back(e);
    }
    return CRENDER_OK;
}

int crender_pipeline_timing_end(crender_pipeline *p, int *frames, double *launch_ms_avg)
{
    if (!p) return fail(CRENDER_EINVAL, "null pipeline");
    for (int k = 0; k < p->depth; ++k) CR_HIP(hipStreamSynchronize(p->s[k]));
    double sum = 0.0;
    const int n = p->timed_frames;
    for (int f = 0; f < n; ++f) {
        float ms = 0.f;
        CR_HIP(hipEventElapsedTime(&ms, p->events[(size_t)f * 2], p->events[(size_t)f * 2 + 1]));
        sum += ms;
    }
    if (frames) *frames = n;
    if (launch_ms_avg) *launch_ms_avg = n ? sum / n : 0.0;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    p->timed_frames = 0;
    return CRENDER_OK;
}

int crender_pipeline_set_lookahead(crender_pipeline *p, crender_plan *const *plans, int n)
{
    if (!p) return fail(CRENDER_EINVAL, "null pipeline");
    if (p->n != 0) return fail(CRENDER_EINVAL, "crender_pipeline_set_lookahead: frames in flight (join first)");
    if (!plans || n == 0) {
        for (int k = 0; k < p->depth; ++k) { p->ahead[k] = nullptr; p->sel[k] = 0; p->primed[k].ok = false; }
        return CRENDER_OK;
    }
    if (n != p->depth) return fail(CRENDER_EINVAL, "crender_pipeline_set_lookahead: one plan per slot");
    for (int i = 0; i < n; ++i) {
        if (!plans[i]) return fail(CRENDER_EINVAL, "crender_pipeline_set_lookahead: null plan");
        for (int j = 0; j < p->depth; ++j)
            if (plans[i] == p->plan[j] || (j < i && plans[i] == plans[j]))
                return fail(CRENDER_EINVAL, "crender_pipeline_set_lookahead: plans must be distinct");
        const Layout &a = plans[i]->L, &b = p->plan[i]->L;
        if (a.ts != b.ts || a.g.W != b.g.W || a.g.H != b.g.H || a.g.y0 != b.g.y0 || a.g.y1 != b.g.y1 ||
            a.max_T != b.max_T)
            return fail(CRENDER_EINVAL, "crender_pipeline_set_lookahead: a slot's two plans must be alike");
    }
    for (int k = 0; k < p->depth; ++k) { p->ahead[k] = plans[k]; plans[k]->stamp_slot = k; p->sel[k] = 0; p->primed[k].ok = false; }
    return CRENDER_OK;
}

int crender_pipeline_bind(crender_pipeline *p, int slot, const float *d_tri, const float *d_col,
                          const float *d_nrm, int64_t T, const float *P16, float *d_z, float *d_color,
                          float *d_normal, int32_t *d_winner, unsigned flags)
{
    if (!p || slot < 0 || slot >= p->depth) return fail(CRENDER_EINVAL, "crender_pipeline_bind: bad slot");
    crender_pipeline::Bound &b = p->bound[slot];
    p->primed[slot].ok = false;          // a new binding may bring new contents at old addresses
    b.tri = d_tri; b.col = d_col; b.nrm = d_nrm; b.T = T;
    b.has_P = P16 != nullptr;
    if (P16) std::memcpy(b.P, P16, sizeof b.P);
    b.z = d_z; b.color = d_color; b.normal = d_normal; b.winner = d_winner; b.flags = flags;
    b.set = true;
    return CRENDER_OK;
}

int crender_pipeline_submit(crender_pipeline *p, void *stream)
{
    if (!p) return fail(CRENDER_EINVAL, "null pipeline");
    const crender_pipeline::Bound &b = p->bound[p->n % (uint64_t)p->depth];
    if (!b.set) return fail(CRENDER_EINVAL, "crender_pipeline_submit: slot not bound");
    return crender_pipeline_frame(p, b.tri, b.col, b.nrm, b.T, b.has_P ? b.P : nullptr, b.z, b.color,
                                  b.normal, b.winner, b.flags, stream);
}

int crender_pipeline_join(crender_pipeline *p, void *stream)
{
    if (!p) return fail(CRENDER_EINVAL, "null pipeline");
    hipStream_t caller = static_cast<hipStream_t>(stream);
    const int used = p->n >= (uint64_t)p->depth ? p->depth : (int)p->n;   // frame j ran on stream j
    for (int k = 0; k < used; ++k) {
        CR_HIP(hipEventRecord(p->done[k], p->s[k]));
        CR_HIP(hipStreamWaitEvent(caller, p->done[k], 0));
    }
    p->n = 0;
    p->synced = false;   // the next frame re-synchronises with the caller's stream
    // the caller may write new inputs behind a join: what was binned ahead from the old ones is void
    // (the abandoned plan starts over like after two crender_prepare calls in a row) — unless the
    // frames came with the promise that the arrays' contents stay (CRENDER_STATIC_INPUTS)
    for (int k = 0; k < p->depth; ++k)
        if (!(p->primed[k].flags & CRENDER_STATIC_INPUTS)) p->primed[k].ok = false;
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
    hipLaunchKernelGGL(k_tile_order_keys, dim3(grid_for((size_t)T, 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), d_tri, T, make_proj(P16, w, h), w, h, d_keys);
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
