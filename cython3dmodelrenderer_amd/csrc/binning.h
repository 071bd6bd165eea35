// binning.h — device side of the direct-bin setup pass that TWO translation units run: binning.hip
// (k_setup_wave, a launch of its own) and raster.hip (k_frame: the same wavefronts inside a raster
// launch, binning the stream's next frame).  Bin entries, the heavy-tile hand-off, the
// per-wavefront aggregated append.
#pragma once
#include "common.h"

namespace crender_detail {

// Binning mode of k_setup (scan path: scenes of more than 65536 triangles, or direct bins
// switched off): count list lengths in an LDS histogram or with global atomics and store each
// triangle's tile range for k_fill.
enum { kBinCountLds = 0, kBinCountGlobal = 1 };
constexpr int kDirectMaxTilesPerTriangle = 4096;  // beyond this the scan path is used instead

#ifdef CRENDER_STAMPS
// Diagnostic build only: phase timestamps per workgroup of the setup kernels, 8 per workgroup
// (crender_debug_set_setup_stamps); same clock as k_raster's stamps.  One pointer per translation
// unit: the setter lives in binning.hip, so the wavefronts k_frame runs (raster.hip) stamp nothing.
static __device__ unsigned long long *g_setup_stamps = nullptr;
#define CR_SETUP_STAMP(slot)                                                                 \
    do {                                                                                     \
        if (g_setup_stamps && threadIdx.x == 0)                                              \
            g_setup_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64();                \
    } while (0)
#else
#define CR_SETUP_STAMP(slot) do { } while (0)
#endif

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

}  // namespace crender_detail
