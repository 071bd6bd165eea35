// abi.hip — the C ABI of include/crender_hip.h that is not a kernel's own entry point: errors,
// the projection matrix, plans (workspace layout, timing, bin usage), whole frames, and the swap
// chain (crender_pipeline_*).  No kernel is launched from here: a frame's two passes are
// bin_pass (binning.hip) and raster_pass (raster.hip).
#include "plan.h"

#include <atomic>
#include <deque>
#include <mutex>

namespace crender_detail {

thread_local std::string g_last_error;
static std::atomic<int> g_default_path{-1};
int default_raster_path() { return g_default_path.load(std::memory_order_relaxed); }

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

size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

int pick_tile(int H, int W, int tile)
{
    if (tile == 16 || tile == 32 || tile == 64) return tile;
    // measured on MI355X (profiles/r01): small frames are latency-bound per tile and want many
    // small tiles; large frames want 32-pixel tiles (full 128-B rows of z, 7 workgroups per CU)
    return ((int64_t)H * W <= (int64_t)1024 * 1024) ? 16 : 32;
}

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
    L.pair_cap = 0;
    if (L.direct_cap == 0 && max_T > 0) {
#ifndef CRENDER_PAIR_MULT
#define CRENDER_PAIR_MULT 3
#endif
        int64_t pc = (CRENDER_PAIR_MULT * max_T / L.g.ntiles + 64 + 63) / 64 * 64;
        // (a scene whose MEAN list — 1.3 entries per triangle — already fills the clamped slab would overflow
        // it on its first frame and leave up to 2 GB allocated and dead: such plans get no slabs at all)
        const bool hopeless = pc > 8192 && 13 * max_T / 10 / L.g.ntiles > 8192;
        if (pc > 8192) pc = 8192;
        if (!hopeless && pc * (int64_t)sizeof(uint2) * L.g.ntiles <= kPairBinBytes) L.pair_cap = pc;
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
    L.off_stats = o;   o = align_up(o + sizeof(uint32_t) * 2 * kStatWords);         // tile size classes, two parities
    L.off_order = o;   o = align_up(o + sizeof(uint32_t) * 2 * (size_t)(L.ordered ? L.g.ntiles : 0));
    L.off_grouped = o; o = align_up(o + 2 * (size_t)(L.ordered ? L.g.ntiles : 0));
    L.off_offs = o;    o = align_up(o + sizeof(uint32_t) * (size_t)(L.g.ntiles + 1));
    L.off_trange = o;  o = align_up(o + sizeof(uint2) * (size_t)max_T);
    L.off_proj = o;    o = align_up(o + sizeof(float) * 9 * (size_t)max_T);
    // (8 bytes per list entry: with a triangle order the entries are (position, caller's index) pairs)
    L.off_entries = o; o = align_up(o + sizeof(uint2) * (size_t)cap);
    L.off_direct = o;  o = align_up(o + sizeof(BinEntry) * (size_t)L.g.ntiles * (size_t)L.direct_cap);
    L.off_pairbins = o; o = align_up(o + sizeof(uint2) * (size_t)L.g.ntiles * (size_t)L.pair_cap);
    L.total = o;
    return true;
}

}  // namespace crender_detail

using namespace crender_detail;

namespace {

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

// Pinned host memory for the plans' usage records: one allocation of kUsageSlots slots, made at the first
// plan and kept for the life of the process (portable: every device's kernels may write into it); a plan
// takes a slot and gives it back.  Beyond kUsageSlots plans alive at once, a plan gets an allocation of
// its own.
void usage_slot_give(uint32_t *host, int slot);
constexpr int kUsageSlots = 1024;
constexpr size_t kUsageSlotWords = kUsageWords * (kUsageRing + 1);
struct UsagePool {
    std::mutex m;
    uint32_t *host = nullptr;
    // FIFO: a slot given back is the LAST to be handed out again (1 023 other plans first) — a plan may be
    // destroyed with a raster launch still in flight (destroying does not wait: a device synchronisation there
    // cost every other filler's frames 0.3 ms), and that launch's record must not land in a live plan's slot.
    // Each plan also salts the sequence word of its records (crender_plan::usage_salt): a record is only ever
    // taken for the (plan, frame) that wrote it.
    std::deque<int> free_slots;
    uint32_t plans = 0;
};
UsagePool &usage_pool()
{
    // leaked on purpose: crender_plan_destroy may run from a static destructor or a Python __del__ AFTER the
    // process's atexit handlers, and must still find the mutex and the queue alive (the pinned block is never
    // freed anyway)
    static UsagePool &pool = *new UsagePool;
    return pool;
}

hipError_t usage_slot_take(uint32_t **host, uint32_t **dev, int *slot, uint32_t *salt)
{
    UsagePool &P = usage_pool();
    void *h = nullptr;
    *slot = -1;
    {
        std::lock_guard<std::mutex> lock(P.m);
        if (!P.host) {
            void *block = nullptr;
            const hipError_t e = hipHostMalloc(&block, sizeof(uint32_t) * kUsageSlotWords * kUsageSlots,
                                               hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable);
            if (e != hipSuccess) return e;
            P.host = static_cast<uint32_t *>(block);
            for (int i = 0; i < kUsageSlots; ++i) P.free_slots.push_back(i);
        }
        if (!P.free_slots.empty()) {
            *slot = P.free_slots.front();
            P.free_slots.pop_front();
            h = P.host + (size_t)*slot * kUsageSlotWords;
        }
        // (bit 31 set: salted with a frame number below 2^31 the word is never 0, a cleared record's value)
        *salt = (++P.plans * 0x9E3779B1u) | 0x80000000u;
    }
    if (!h) {
        const hipError_t e = hipHostMalloc(&h, sizeof(uint32_t) * kUsageSlotWords, hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) return e;
    }
    std::memset(h, 0, sizeof(uint32_t) * kUsageSlotWords);
    void *d = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
        usage_slot_give(static_cast<uint32_t *>(h), *slot);
        return e;
    }
    *host = static_cast<uint32_t *>(h);
    *dev = static_cast<uint32_t *>(d);
    return hipSuccess;
}

void usage_slot_give(uint32_t *host, int slot)
{
    if (!host) return;
    if (slot < 0) {
        (void)hipHostFree(host);
        return;
    }
    UsagePool &P = usage_pool();
    std::lock_guard<std::mutex> lock(P.m);
    P.free_slots.push_back(slot);
}

// What the header words of a frame mean (crender_plan_last_bin_usage, crender_plan_poll_bin_usage).
void usage_figures(crender_plan *plan, int mode, uint32_t h0, uint32_t h1, uint32_t h4, int64_t *needed,
                   int64_t *capacity)
{
    if (mode != 0) {
        // direct bins / pair bins: per-tile figures.  h1 is sticky: the longest list that did not fit
        // (0xFFFFFFFF = a triangle spans too many tiles).  On overflow this plan switches to
        // the count / scan / fill path for good; the caller renders the frame again.
        const int64_t cap = mode == 1 ? plan->L.direct_cap : plan->L.pair_cap;
        if (h1 > (uint32_t)cap) {                          // h1 stays set: the answer is repeatable
            if (mode == 1) plan->direct_ok = false;
            else plan->pairbins_ok = false;
        }
        if (needed) *needed = h1 > (uint32_t)cap ? (int64_t)h1 : 0;
        if (capacity) *capacity = cap;
        return;
    }
    if (needed) *needed = (int64_t)(((unsigned long long)h4 << 32) | h0);
    if (capacity) *capacity = plan->L.capacity;
}

}  // namespace

// =========================== C ABI ================================================
extern "C" {

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
    // the per-frame usage records: pinned, coherent host memory the raster launches write into — a
    // slot of the process-wide pool (hipHostMalloc / hipHostFree per plan cost milliseconds, and the
    // free synchronises the device: a filler collected in the middle of another one's frames showed as
    // 0.3 ms per call)
    if (e == hipSuccess) e = usage_slot_take(&p->usage, &p->usage_dev, &p->usage_slot, &p->usage_salt);
    if (e != hipSuccess) {
        delete p;
        return fail_hip(e, "crender_plan_create (workspace memset / pinned usage records)");
    }
    *out = p;
    return CRENDER_OK;
}

void crender_plan_destroy(crender_plan *plan)
{
    if (!plan) return;
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    usage_slot_give(plan->usage, plan->usage_slot);
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
    // (into the plan's own pinned memory: a pageable destination makes the runtime stage the copy)
    volatile uint32_t *h = plan->usage + kUsageWords * kUsageRing;       // (the record behind the ring)
    hipStream_t s = static_cast<hipStream_t>(stream);
    CR_HIP(hipMemcpyAsync(const_cast<uint32_t *>(h), plan->hdr(), 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CR_HIP(hipStreamSynchronize(s));
    usage_figures(plan, plan->last_frame_direct ? 1 : plan->last_frame_pairbins ? 2 : 0, h[0], h[1], h[4], needed, capacity);
    return CRENDER_OK;
}

uint64_t crender_plan_frame_ticket(crender_plan *plan) { return plan ? plan->ticket : 0; }

int crender_plan_poll_bin_usage(crender_plan *plan, uint64_t ticket, int64_t *needed, int64_t *capacity)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    if (ticket == 0 || ticket > plan->ticket || plan->ticket - ticket >= (uint64_t)kUsageRing)
        return fail(CRENDER_EINVAL, "crender_plan_poll_bin_usage: no such frame (not launched yet, or more than "
                                    "8 frames ago: its record has been reused)");
    const int slot = (int)(ticket % kUsageRing);
    const volatile uint32_t *rec = plan->usage + kUsageWords * slot;
    // the record is two aligned 16-byte stores of the launch, the frame's sequence word leading the first
    // and trailing the second: taken only when BOTH are there (a record half landed reads as not landed)
    const uint32_t want = (uint32_t)ticket ^ plan->usage_salt;
    if (__atomic_load_n(const_cast<const uint32_t *>(rec), __ATOMIC_ACQUIRE) != want || rec[7] != want)
        return CRENDER_EBUSY;                                                      // (not an error: no text)
    const uint32_t h0 = rec[1], h1 = rec[2], h4 = rec[3];
    if (rec[0] != want || rec[7] != want) return CRENDER_EBUSY;                    // (rewritten under the read: a frame 8 later)
    usage_figures(plan, plan->usage_mode[slot], h0, h1, h4, needed, capacity);
    return CRENDER_OK;
}

// ---- state check between frames (diagnostics) ---------------------------------------------------------------
// A plan carries state from frame to frame that no single frame's pixels show: two parities of per-tile
// counters, the split tiles' flag and helper-slot words, the registration and hint_bad counters, the dispatch
// order and its header, the ring of usage records.  Four defects of rounds 4-5 were words of that state left
// behind by one frame and found, frames later, by their pixel symptom.  This reads the state back and checks
// the invariants themselves.
int crender_plan_debug_check(crender_plan *plan, void *stream, char *msg, size_t msg_bytes)
{
    if (!plan) return fail(CRENDER_EINVAL, "null plan");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Layout &L = plan->L;
    const int nt = L.g.ntiles;
    std::vector<unsigned char> host(L.off_offs);
    CR_HIP(hipMemcpyAsync(host.data(), plan->ws, L.off_offs, hipMemcpyDeviceToHost, s));
    CR_HIP(hipStreamSynchronize(s));
    auto words = [&](size_t off) { return reinterpret_cast<const uint32_t *>(host.data() + off); };
    const uint32_t *hdr = words(L.off_hdr), *hflag = words(L.off_hflag), *hslots = words(L.off_hslots);
    const uint32_t *cnt[2] = {words(L.off_count), words(L.off_count) + L.count_stride};
    std::string bad;
    char line[256];
    auto say = [&](const char *fmt, auto... a) {
        if (bad.size() > 1500) return;
        std::snprintf(line, sizeof line, fmt, a...);
        bad += line;
        bad += "; ";
    };
    const int par = plan->parity;
    const bool binned_only = plan->unrastered[par];          // binned, its raster launch not issued yet
    const bool virgin = plan->frame_no == 0;
    if (!virgin && !binned_only && plan->unrastered[par ^ 1])
        say("parity %d holds a binned frame that was never rasterized although a later frame was binned and rasterized: "
            "that frame's binning pass did not start the plan over", par ^ 1);
    if (!virgin && !plan->unrastered[0] && !plan->unrastered[1]) {
        // (A) the last frame was rasterized: the other parity is ready for the next frame, nothing is handed off
        int nz = 0, first = -1;
        for (int i = 0; i < nt; ++i)
            if (cnt[par ^ 1][i]) { if (first < 0) first = i; ++nz; }
        if (nz) say("%d counters of parity %d (the NEXT frame's) are not zero, first tile %d = %u", nz, par ^ 1, first, cnt[par ^ 1][first]);
        if (plan->awaiting[par ^ 1]) say("host flag awaiting[%d] still set after the raster launch", par ^ 1);
        if (hdr[2 + (par ^ 1)]) say("registration counter of the next frame hdr[%d] = %u", 2 + (par ^ 1), hdr[2 + (par ^ 1)]);
        if (hdr[5 + (par ^ 1)]) say("hint_bad of the next frame hdr[%d] = %u", 5 + (par ^ 1), hdr[5 + (par ^ 1)]);
        int nf = 0, ff = -1;
        for (int i = 0; i < nt; ++i)
            if (hflag[i]) { if (ff < 0) ff = i; ++nf; }
        if (nf) say("%d split flags left up after the raster launch, first tile %d", nf, ff);
        int nsl = 0, fs = -1;
        for (int i = 0; i < 3 * (L.hmax > 0 ? L.hmax : 1); ++i)
            if (hslots[i]) { if (fs < 0) fs = i; ++nsl; }
        if (nsl) say("%d helper-slot words left set after the raster launch, first slot %d = tile %u", nsl, fs, hslots[fs] - 1);
    } else if (binned_only) {
        // (B) binned, not rasterized yet: the registrations must match the flags, the slots and the lists —
        // and the OTHER parity must not hold a binned frame that was never rasterized: a binning pass that finds
        // one starts the plan over (flags, slots and the order hint exist once, not per parity)
        if (plan->unrastered[par ^ 1])
            say("%s", "both parities hold a binned frame that was never rasterized: the binning pass did not start the plan over");
        const bool split = plan->last_frame_direct && plan->frame_hmax > 0;
        const uint32_t at = plan->frame_heavy_at;
        if (split) {
            const uint32_t reg = hdr[2 + par];
            const uint32_t hm = (uint32_t)plan->frame_hmax;
            const uint32_t used = reg < hm ? reg : hm;
            std::vector<int> slot_of(nt, -1);
            for (uint32_t k = 0; k < (uint32_t)L.hmax; ++k) {
                const uint32_t a = hslots[3 * k], b = hslots[3 * k + 1], c = hslots[3 * k + 2];
                if (k >= used) {
                    if (a | b | c) say("helper triple %u beyond the %u registered holds tiles %u %u %u", k, used, a, b, c);
                    continue;
                }
                if (a == 0 || a != b || a != c || a > (uint32_t)nt) { say("helper triple %u of %u registered holds %u %u %u", k, used, a, b, c); continue; }
                const int tile = (int)a - 1;
                if (slot_of[tile] >= 0) say("tile %d registered twice (triples %d and %u)", tile, slot_of[tile], k);
                slot_of[tile] = (int)k;
                if (!hflag[tile]) say("tile %d has helper triple %u but no flag", tile, k);
                if (cnt[par][tile] < at) say("tile %d registered with a list of %u < %u", tile, cnt[par][tile], at);
            }
            uint32_t nheavy = 0;
            for (int i = 0; i < nt; ++i) {
                if (cnt[par][i] >= at) ++nheavy;
                if (hflag[i] && slot_of[i] < 0) say("tile %d flagged without a helper triple", i);
            }
            if (nheavy != reg) say("%u tiles reach %u entries but %u registered", nheavy, at, reg);
        } else {
            for (int i = 0; i < nt; ++i)
                if (hflag[i]) { say("split flag of tile %d up on a frame that is not split", i); break; }
        }
    }
    // the dispatch order the next ordered launch reads: a permutation, its grouped section as the header says
    if (L.ordered && plan->last_ordered) {
        const int hp = plan->hint_par;
        const uint32_t *hint = words(L.off_hint) + 4 * hp;
        const uint32_t *order = words(L.off_order) + (size_t)hp * nt;
        const unsigned char *grouped = host.data() + L.off_grouped + (size_t)hp * nt;
        std::vector<unsigned char> seen(nt, 0);
        int dup = 0, oob = 0;
        for (int i = 0; i < nt; ++i) {
            if (order[i] >= (uint32_t)nt) { ++oob; continue; }
            if (seen[order[i]]++) ++dup;
        }
        if (dup || oob) say("the dispatch order is no permutation (%d repeated, %d out of range)", dup, oob);
        const uint32_t ns = hint[1];
        // (hint[0] == 0: the launch reads no order — an empty frame's, or a header a binning pass that started the
        // plan over has zeroed: nothing to agree with)
        if (hint[0] == 0) { }
        else if (ns > (uint32_t)nt) say("order header: %u tiles with a workgroup of their own of %d", ns, nt);
        else if (!dup && !oob) {
            int wrong = 0;
            for (int i = 0; i < nt; ++i)
                if ((grouped[order[i]] != 0) != ((uint32_t)i >= ns)) ++wrong;
            if (wrong) say("%d tiles whose `grouped` byte disagrees with their place in the order", wrong);
            const int g = L.ts >= 32 ? 2 : 8;
            if (hint[2] != ((uint32_t)nt - ns + g - 1) / g) say("order header: %u groups for %u grouped tiles", hint[2], (uint32_t)nt - ns);
        }
    }
    // the usage ring: every landed record is one of this plan's launches, in its own slot, whole
    for (int k = 0; k < kUsageRing; ++k) {
        const volatile uint32_t *rec = plan->usage + kUsageWords * k;
        const uint32_t a = rec[0], z = rec[7];
        if (a == 0 && z == 0) continue;
        const uint64_t t = (uint64_t)(a ^ plan->usage_salt);
        if (a != z) say("usage record %d: leading word %08x, trailing word %08x", k, a, z);
        else if (t == 0 || t > plan->ticket || (int)(t % kUsageRing) != k) say("usage record %d names launch %llu of %llu", k, (unsigned long long)t, (unsigned long long)plan->ticket);
    }
    if (msg && msg_bytes) {
        std::snprintf(msg, msg_bytes, "%s", bad.c_str());
    }
    if (!bad.empty()) return fail(CRENDER_ESTATE, "crender_plan_debug_check: the plan's cross-frame state is inconsistent (text in msg)");
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

int crender_plan_set_normal_z(crender_plan *plan, const float *d_nz)
{
    if (!plan) return fail(CRENDER_EINVAL, "crender_plan_set_normal_z: null plan");
    plan->normal_z = d_nz;
    return CRENDER_OK;
}

int crender_plan_set_light(crender_plan *plan, const float *light3)
{
    if (!plan || !light3) return fail(CRENDER_EINVAL, "crender_plan_set_light: bad argument");
    plan->light[0] = light3[0]; plan->light[1] = light3[1]; plan->light[2] = light3[2];
    return CRENDER_OK;
}

int crender_plan_set_raster_path(crender_plan *plan, int path)
{
    if (!plan || path < -1 || path > 1) return fail(CRENDER_EINVAL, "crender_plan_set_raster_path: path is -1 (automatic), 0 or 1");
    plan->forced_path = path;
    return CRENDER_OK;
}

int crender_plan_last_raster_path(crender_plan *plan) { return plan ? plan->last_path : 0; }

int crender_set_default_raster_path(int path)
{
    if (path < -1 || path > 1) return fail(CRENDER_EINVAL, "crender_set_default_raster_path: path is -1 (automatic), 0 or 1");
    g_default_path.store(path, std::memory_order_relaxed);
    return CRENDER_OK;
}

int crender_plan_last_frame_direct(crender_plan *plan)
{
    return plan && (plan->last_frame_direct || plan->last_frame_pairbins) ? 1 : 0;
}

int crender_plan_last_frame_binning(crender_plan *plan)
{
    return !plan ? 0 : plan->last_frame_direct ? 1 : plan->last_frame_pairbins ? 2 : 0;
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
    {   // the chain's plans share what any of them has learnt about the frames' size class (raster_path_hint)
        crender_plan *use = (p->ahead[k] && P16 && T > 0 && p->sel[k]) ? p->ahead[k] : p->plan[k];
        if (raster_path_hint(use)) { p->shared_path = use->auto_path; p->shared_known = true; }
        else if (p->shared_known) { use->auto_path = p->shared_path; use->auto_known = true; }
    }
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

}  // extern "C"
