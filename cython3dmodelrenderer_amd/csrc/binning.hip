// binning.hip — K1 and the binning passes of a frame (gfx950), and their host side.
//
//   k_project      K1 alone (crender_project; the broadcast-of-projected-vertices variant)
//   k_clear        __cinit__'s buffer state (crender_clear)
//   k_setup_wave   direct bins (scenes <= 65536 triangles): project, cull, box, append 48-byte entries
//   k_count_wave / k_setup -> k_scan -> k_fill_wave / k_fill    scan path (larger scenes)
//
// Build flags (see _build.py): -ffp-contract=off, correctly rounded division, denormals on — float
// parity with the reference depends on them.
#include "plan.h"

#ifndef CRENDER_BIN_PER
#define CRENDER_BIN_PER 1
#endif

using namespace crender_detail;

namespace {

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
                                                      const float *__restrict__ nz3,
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
        if (nz3) {          // the components apart (crender_plan_set_normal_z): 12 contiguous bytes per triangle
            const float *nn = nz3 + (b0 + lane) * 3;
            nz0 = nn[0]; nz1 = nn[1]; nz2 = nn[2];
        } else {
            const float *nn = nrm + (b0 + lane) * 9;
            nz0 = nn[2]; nz1 = nn[5]; nz2 = nn[8];
        }
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

// ---- pair bins: large scenes binned in ONE pass --------------------------------------------------
// k_count_wave's front (project, cull, pixel box, tile range, the projected vertices out) and k_fill_wave's
// back (entries per tile in an LDS histogram over the wavefront's tile bounding box, one returning atomic
// per touched tile, LDS cursors hand out the slots) in one wavefront — against fixed-capacity per-tile
// slabs instead of scanned lists, so that neither the count pass's tile ranges (8 bytes per triangle
// written and read again) nor the scan nor a second pass over the triangles exist.  Entries are
// (position, caller's index) pairs (the position itself without a triangle order).  10 M small triangles:
// k_count_wave 0.152 + k_scan 0.015 + k_fill_wave 0.087 ms -> this kernel alone.  A list that outgrows its
// slab is reported like an overflow of the direct bins (hdr[1], sticky) and the plan returns to the three passes.
// Groups of 64 triangles a wavefront takes through the chain together.  One: more groups lengthen the runs
// of neighbouring positions in the lists (the raster launch gathers its records faster: 0.559 / 0.552 / 0.524 ms
// with 1 / 2 / 4) but slow this pass down by more (0.176 / 0.195 / 0.279 ms; profiles/r05/ab_pair_bins_synth10m.txt).
constexpr int kBinPer = CRENDER_BIN_PER;
template <int TS, bool PROJECT>
__global__ __launch_bounds__(kWave) void k_bin_wave(const float *__restrict__ tri_in,
                                                    const float *__restrict__ nrm,
                                                    const float *__restrict__ nz3,
                                                    const uint32_t *__restrict__ orig_of,
                                                    float *__restrict__ proj_out,
                                                    uint32_t *__restrict__ count,
                                                    uint2 *__restrict__ pairs, uint32_t cap,
                                                    uint32_t *__restrict__ hdr, int64_t T,
                                                    ProjConst P, Geom G)
{
    __shared__ __attribute__((aligned(16))) float sv[kBinPer * kWave * 9];
    __shared__ uint32_t hist[kWaveHistTiles];
    __shared__ uint32_t orig[kBinPer * kWave];            // (any lane writes any owner's entry)
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * (kWave * kBinPer);
    const int n = (int)((T - b0) < kWave * kBinPer ? (T - b0) : kWave * kBinPer);
    stage_in<kWave>(tri_in + b0 * 9, sv, n * 9);
#pragma unroll
    for (int i = 0; i < kWaveHistTiles / kWave; ++i) hist[i * kWave + lane] = 0;   // (while the inputs are on their way)
    float nz[kBinPer][3];                          // .pyx:202 looks at the normals' z only
#pragma unroll
    for (int p = 0; p < kBinPer; ++p) {
        const int k = p * kWave + lane;
        nz[p][0] = nz[p][1] = nz[p][2] = 0.0f;
        if (k < n) {
            if (nz3) {          // the components apart (crender_plan_set_normal_z): 12 contiguous bytes per triangle
                const float *nn = nz3 + (b0 + k) * 3;
                nz[p][0] = nn[0]; nz[p][1] = nn[1]; nz[p][2] = nn[2];
            } else {
                const float *nn = nrm + (b0 + k) * 9;
                nz[p][0] = nn[2]; nz[p][1] = nn[5]; nz[p][2] = nn[8];
            }
            orig[k] = orig_of ? orig_of[b0 + k] : (uint32_t)(b0 + k);
        }
    }
    __syncthreads();
    uint2 r[kBinPer];
    int X0 = 0x7FFFFFFF, X1 = -1, Y0 = 0x7FFFFFFF, Y1 = -1;
#pragma unroll
    for (int p = 0; p < kBinPer; ++p) {
        const int k = p * kWave + lane;
        r[p] = make_uint2(kNoTiles, 0);
        if (k < n) {
            float *v = sv + k * 9;
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
            if (!backface(nz[p][0], nz[p][1], nz[p][2])) r[p] = tile_range<TS>(t, G);
        }
        if (r[p].x != kNoTiles) {
            const int x0 = r[p].x & 0xFFFF, x1 = r[p].x >> 16, y0 = r[p].y & 0xFFFF, y1 = r[p].y >> 16;
            X0 = x0 < X0 ? x0 : X0; X1 = x1 > X1 ? x1 : X1; Y0 = y0 < Y0 ? y0 : Y0; Y1 = y1 > Y1 ? y1 : Y1;
        }
    }
    wave_box(X0, X1, Y0, Y1);
    __syncthreads();
    if (PROJECT) stage_out<kWave>(proj_out + b0 * 9, sv, n * 9);
    if (X1 < 0) return;     // nothing to bin (uniform)
    auto put = [&](uint32_t tile, uint32_t slot, int owner) {      // owner: 0 .. kBinPer * 64 - 1
        if (slot < cap) pairs[(size_t)tile * cap + slot] = make_uint2((uint32_t)(b0 + owner), orig[owner]);
    };
    const int bw = X1 - X0 + 1, area = bw * (Y1 - Y0 + 1);
    if (area > kWaveHistTiles) {
        // (large triangles: the wavefront's box exceeds the histogram) pair by pair
#pragma unroll
        for (int p = 0; p < kBinPer; ++p)
            for_each_tile(r[p], (uint32_t)(p * kWave + lane), G.ntx, [&](int tile, uint32_t owner) {
                const uint32_t slot = atomicAdd(&count[tile], 1u);
                if (slot >= cap) atomicMax(&hdr[1], slot + 1u);
                put((uint32_t)tile, slot, (int)owner);
            });
        return;
    }
#pragma unroll
    for (int p = 0; p < kBinPer; ++p)
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
            base[k] = c[k] ? atomicAdd(&count[t[k]], c[k]) : 0u;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            if (c[k]) {
                hist[k * kWave + lane] = base[k];
                if (base[k] + c[k] > cap) atomicMax(&hdr[1], base[k] + c[k]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kBinPer; ++p)
        for_each_tile_xy(r[p], [&](int tx, int ty, int owner) {
            const uint32_t slot = atomicAdd(&hist[(ty - Y0) * bw + (tx - X0)], 1u);
            put((uint32_t)(ty * G.ntx + tx), slot, p * kWave + owner);
        });
}

// The fill pass in the same shape: (A) the wavefront's entries per tile in LDS, (B) one returning
// atomic per touched tile on the list's cursor reserves a run, (C) LDS cursors hand out its slots.
// trange is read once (k_fill's block histograms need two sweeps).  The pass is a latency chain —
// ranges, then offsets + returning atomics, then the entries: three memory round trips per wavefront,
// 8 192 wavefronts resident — so a wavefront takes kFillPer x 64 triangles through each round trip
// together (10 M triangles: 97 -> 78 us with two, 77 with four; profiles/r03/ab_fill_groups.txt).
// PAIRS (a triangle order is set, crender_plan_set_triangle_order): an entry is the pair (position in
// the arrays, caller's index) — the raster kernel keys its depth comparisons on the caller's index and
// used to gather it per entry from orig_of: for the sixth of a tile's records that live in a
// neighbouring tile's cluster that was a 128-byte line for 4 bytes (10 M triangles: 260 MB per frame).
// Here orig_of is read as a stream, 4 bytes per triangle.
constexpr int kFillPer = 2;
template <bool PAIRS>
__global__ __launch_bounds__(kWave) void k_fill_wave(const uint2 *__restrict__ trange,
                                                     const uint32_t *__restrict__ offs,
                                                     uint32_t *__restrict__ cursor,
                                                     uint32_t *__restrict__ entries,
                                                     const uint32_t *__restrict__ orig_of,
                                                     uint32_t capacity, int64_t T, Geom G)
{
    __shared__ uint32_t hist[kWaveHistTiles];
    __shared__ uint32_t orig[PAIRS ? kWave * kFillPer : 1];     // the wavefront's caller's indices (any lane writes any owner's entry)
    uint2 *const pairs = reinterpret_cast<uint2 *>(entries);
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * (kWave * kFillPer);
    uint2 r[kFillPer];
#pragma unroll
    for (int p = 0; p < kFillPer; ++p) {
        const bool in = b0 + p * kWave + lane < T;
        r[p] = in ? trange[b0 + p * kWave + lane] : make_uint2(kNoTiles, 0);
        if constexpr (PAIRS) orig[p * kWave + lane] = in ? orig_of[b0 + p * kWave + lane] : 0u;
    }
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
    __syncthreads();        // histogram zeroed, the caller's indices in LDS
    const int bw = X1 - X0 + 1, area = bw * (Y1 - Y0 + 1);
    if (area > kWaveHistTiles) {
#pragma unroll
        for (int p = 0; p < kFillPer; ++p)
            for_each_tile(r[p], (uint32_t)(p * kWave + lane), G.ntx, [&](int tile, uint32_t local) {
                const uint32_t pos = offs[tile] + atomicAdd(&cursor[tile], 1u);
                if (pos < capacity) {
                    if constexpr (PAIRS) pairs[pos] = make_uint2((uint32_t)b0 + local, orig[local]);
                    else entries[pos] = (uint32_t)b0 + local;
                }
            });
        return;
    }
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
            if (pos < capacity) {
                if constexpr (PAIRS) pairs[pos] = make_uint2((uint32_t)(b0 + p * kWave + owner), orig[p * kWave + owner]);
                else entries[pos] = (uint32_t)(b0 + p * kWave + owner);
            }
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
                r[u] = none;                    // (not `cond ? trange[t] : none`: a select between two
                if (t < c1) r[u] = trange[t];   // lvalues keeps `none` in scratch memory)
            }
#pragma clang loop unroll(full)
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
            r[u] = none;                    // (not `cond ? trange[t] : none`: a select between two
                if (t < c1) r[u] = trange[t];   // lvalues keeps `none` in scratch memory)
        }
#pragma clang loop unroll(full)
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

// LDS histograms up to this many tiles: k_setup packs 16-bit counters (32 KiB + 18 KiB of
// staging), k_fill needs 32-bit cursors (64 KiB, the dynamic-LDS limit is raised for it).
constexpr int kMaxLdsHistTiles = 16384;
constexpr int64_t kWaveScanBelow = 1 << 18;   // (the filler keeps larger models tile-coherent)

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
    // Which lists are split among several workgroups (binning.h, "heavy tiles"): on a frame rendered alone the
    // plan's own thresholds (32-pixel plans of small frames: every covered tile, in quadrants); on a frame of a
    // swap chain — whose tiles are one workgroup each — only the long ones, from 32 records in halves and from 64
    // in quadrants, at most kMaxHeavyHelped of them: a frame in flight whose longest tiles are shared out ends
    // sooner when it runs with few others (the fill and drain of a burst), and costs the steady state nothing
    // (T-Rex 1024^2: +2.7 % at the driver's 20 steps, +1.9 % at 200; profiles/r06/ab_split_overlapped.txt).
    plan->frame_hmax = 0;
    plan->frame_heavy_at = plan->frame_quad_at = 0;
    if (direct && L.hmax > 0 && !(dbg & 2048)) {
        plan->frame_hmax = plan->frame_lone || L.hmax < kMaxHeavyHelped ? L.hmax : kMaxHeavyHelped;
        plan->frame_heavy_at = plan->frame_lone ? heavy_at(TS) : kHeavyAt;
        plan->frame_quad_at = plan->frame_lone ? quad_at(TS) : kQuadAt;
    }
#if defined(CRENDER_FAULT) && CRENDER_FAULT == 2     // (round 5's defect back in, for the state check's own test: scripts/r6_faults.sh)
    if (plan->awaiting[par]) {
#else
    if (plan->awaiting[par] || plan->unrastered[par ^ 1]) {
#endif
        // this parity was binned into and not zeroed since, or the other one was binned into and never
        // rasterized (the swap chain binned ahead for inputs that then changed; crender_prepare twice in a
        // row): start over from the state crender_plan_create leaves.  (The second case was missed until
        // round 5: the discarded frame's tiles kept their split flags, and the next frame on the plan
        // cleared only half of those it did not cover — test_lone_chain_through_changing_scenes.)
        CR_HIP(hipMemsetAsync(plan->ws + L.off_count, 0, L.off_order - L.off_count, s));
        CR_HIP(hipMemsetAsync(plan->hdr() + 2, 0, 5 * sizeof(uint32_t), s));   // heavy counters, hint_bad
        plan->awaiting[0] = plan->awaiting[1] = false;
        plan->unrastered[0] = plan->unrastered[1] = false;
    }
    plan->awaiting[par] = true;
    plan->unrastered[par] = true;
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
    // large scenes: ONE binning pass into fixed-capacity slabs of (position, index) pairs (k_bin_wave)
    const bool pairbins = !direct && wave_scan && T > 0 && L.pair_cap > 0 && plan->pairbins_ok &&
                          !(flags & CRENDER_NO_DIRECT_BINS) && !(dbg & 16);
    plan->last_frame_pairbins = pairbins;
    plan->last_frame_pairs = pairbins || (!direct && wave_scan && plan->orig_of != nullptr && T > 0);
    if (T > 0 && direct) {
        // direct bins: one wavefront per 64 triangles
        HeavyReg hv;
        if (plan->frame_hmax > 0) {
            hv.ctr = plan->hdr() + 2 + par; hv.flag = plan->hflag(); hv.slots = plan->hslots();
            hv.hmax = (uint32_t)plan->frame_hmax;
            hv.heavy_at = plan->frame_heavy_at;
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
    } else if (pairbins) {
        const unsigned nwg = (unsigned)((T + kWave * kBinPer - 1) / (kWave * kBinPer));
        if (project)
            hipLaunchKernelGGL((k_bin_wave<TS, true>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm, plan->normal_z, plan->orig_of,
                               plan->proj(), count, plan->pairbins(), (uint32_t)L.pair_cap, plan->hdr(), T, P, G);
        else
            hipLaunchKernelGGL((k_bin_wave<TS, false>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm, plan->normal_z, plan->orig_of,
                               plan->proj(), count, plan->pairbins(), (uint32_t)L.pair_cap, plan->hdr(), T, P, G);
        CR_LAUNCH_CHECK("k_bin_wave");
    } else if (T > 0 && wave_scan) {
        const unsigned nwg = (unsigned)((T + kWave - 1) / kWave);
        if (project)
            hipLaunchKernelGGL((k_count_wave<TS, true>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm, plan->normal_z,
                               plan->proj(), plan->trange(), count, T, P, G);
        else
            hipLaunchKernelGGL((k_count_wave<TS, false>), dim3(nwg), dim3(kWave), 0, s, d_tri, d_nrm, plan->normal_z,
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
    if (!direct && !pairbins) {
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
            if (wave_scan && plan->orig_of)
                hipLaunchKernelGGL((k_fill_wave<true>), dim3((unsigned)((T + kWave * kFillPer - 1) / (kWave * kFillPer))), dim3(kWave), 0, s,
                                   plan->trange(), plan->offs(), count, plan->entries(), plan->orig_of,
                                   (uint32_t)L.capacity, T, G);
            else if (wave_scan)
                hipLaunchKernelGGL((k_fill_wave<false>), dim3((unsigned)((T + kWave * kFillPer - 1) / (kWave * kFillPer))), dim3(kWave), 0, s,
                                   plan->trange(), plan->offs(), count, plan->entries(), plan->orig_of,
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

}  // namespace

namespace crender_detail {

int bin_pass(crender_plan *plan, bool project, const float *d_tri, const float *d_nrm, int64_t T,
             const float *P16, unsigned flags, void *stream, SetupArgs *defer,
             bool *deferred)
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

}  // namespace crender_detail

extern "C" {

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

#ifdef CRENDER_STAMPS
CRENDER_API int crender_debug_set_setup_stamps(void *d_buf);
int crender_debug_set_setup_stamps(void *d_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(d_buf);
    CR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_setup_stamps), &p, sizeof p));
    return CRENDER_OK;
}
#endif

}  // extern "C"
