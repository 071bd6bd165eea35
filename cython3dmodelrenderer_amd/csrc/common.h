// common.h — what every translation unit of libcrender_hip.so shares: error plumbing, launch
// constants, the strip geometry handed to kernels, LDS staging of [n][9] chunks, wavefront-wide
// reductions and scans (DPP), and the tile-range walkers of the binning passes.
#pragma once
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

namespace crender_detail {
using namespace crender;

// (defined in abi.hip; crender_last_error() hands the text out)
extern thread_local std::string g_last_error;
int fail(int code, const char *what);
int fail_hip(hipError_t e, const char *where);

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
// the 32-pixel kernel with the pixel owners alone (raster.hip, kPathOwners): seven wavefronts (70 registers,
// 19.5 KB of LDS)
#ifndef CRENDER_WAVES_OWNERS
#define CRENDER_WAVES_OWNERS 7
#endif
constexpr int kWavesPerSimdOwners = CRENDER_WAVES_OWNERS;
constexpr int kItemPixels = 2;      // samples per work item of the per-pixel sweep of 16-pixel tiles
#ifndef CRENDER_ITEM32
#define CRENDER_ITEM32 2
#endif
constexpr int kItemPixels32 = CRENDER_ITEM32;    // the same for the small-record batches of 32-pixel tiles
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
inline ProjConst make_proj(const float *P16, int w, int h)
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

// Any lane true?  (HIP's __any goes through a 0 / 1 vector register and a second compare; the ballot
// of the condition is the compare's own scalar result.)
CR_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

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

inline int grid_for(size_t items, int cap)
{
    size_t b = (items + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = (size_t)cap;
    return (int)b;
}

// CRENDER_DEBUG of a development build (-DCRENDER_DEV_KNOBS; see raster.hip), read once; 0 in the product
inline int dev_knobs()
{
#ifdef CRENDER_DEV_KNOBS
    static const int dbg = std::getenv("CRENDER_DEBUG") ? std::atoi(std::getenv("CRENDER_DEBUG")) : 0;
    return dbg;
#else
    return 0;
#endif
}

}  // namespace crender_detail
