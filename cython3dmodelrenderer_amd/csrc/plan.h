// plan.h — host-side state behind the opaque handles of include/crender_hip.h: the workspace layout
// of a plan, the plan, the swap chain; and the two passes of a frame, which live with their kernels
// (bin_pass in binning.hip, raster_pass in raster.hip) and are called from abi.hip.
#pragma once
#include "binning.h"

namespace crender_detail {

constexpr size_t kAlign = 256;
size_t align_up(size_t v);
int pick_tile(int H, int W, int tile);

struct Layout {
    int ts;
    Geom g;
    int64_t max_T;
    int64_t capacity;
    int64_t direct_cap;   // entries per tile of the direct bins, 0 = scene too large for them
    int64_t pair_cap;     // entries per tile of the PAIR bins (large scenes binned in one pass, k_bin_wave), 0 = none
    int hmax;             // helper triples of a raster launch (heavy tiles split in four), 0 = none
    bool ordered;         // raster launches leave a dispatch order for the next one (build_order)
    size_t count_stride;  // u32 words between the two parities of the per-tile counters
    size_t off_hdr, off_count, off_hflag, off_hslots, off_hint, off_stats, off_order, off_grouped, off_offs,
           off_trange, off_proj, off_entries, off_direct, off_pairbins, total;
};
constexpr int kUsageRing = 8;
constexpr int kUsageWords = 8;         // words per usage record: {seq, hdr[0], hdr[1], hdr[4]} {large, small, path, seq}
constexpr int kStatWords = 2 * 16;     // per parity: kStatSlots (large, small) pairs (raster.hip, TileLists::stats)
int default_raster_path();             // crender_set_default_raster_path (-1: none)
constexpr int kOrderMaxTiles = 8192;   // ordered dispatch: the builder keeps one byte per tile in the batch queue's LDS
constexpr int kMaxHeavyHelped32 = 512;
constexpr int kMaxHeavyHelped = 128;   // +384 workgroups per raster launch (9 % at 1024 x 1024)

// Direct bins are for small scenes (the README benchmark): one launch fewer than the
// count / scan / fill path matters when a frame takes tens of microseconds.
constexpr int64_t kDirectMaxTriangles = 1 << 16;
constexpr int kDirectMaxTiles = 1 << 16;    // beyond: count / scan / fill
constexpr int64_t kDirectBinBytes = 512ll << 20;   // per-tile capacity = this budget / 48 B / tiles, <= 1024
// Pair bins (scenes beyond the direct bins): fixed-capacity per-tile slabs of 8-byte (position, index) entries
// that ONE binning pass appends to (k_bin_wave) — no count pass, no scan, no fill pass.  Sized at three
// times the mean list (entries per triangle ~ 1.3), within 64 .. 8192 entries and kPairBinBytes in all; a
// frame that does not fit reports so like the direct bins and the plan goes back to count / scan / fill.
constexpr int64_t kPairBinBytes = 2048ll << 20;
bool make_layout(int H, int W, int y0, int y1, int64_t max_T, int64_t cap, int tile, Layout &L);

}  // namespace crender_detail

using namespace crender_detail;     // (the handles are global types: extern "C" names them)

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
    bool last_frame_pairs = false;    // its list entries are (position, caller's index) pairs (k_fill_wave<true>, k_bin_wave)
    bool pairbins_ok = true;          // cleared once a frame overflowed the pair bins
    bool last_frame_pairbins = false; // binned in ONE pass into fixed-capacity per-tile slabs of pairs (k_bin_wave)
    int64_t last_T = -1;          // triangle count of the last bin pass (crender_draw must match)
    // The per-tile counters exist twice.  Frame f bins into parity f & 1 and its raster pass
    // zeroes the OTHER parity for frame f + 1, so no raster workgroup ever writes a counter that
    // another workgroup of the same launch reads (the four workgroups of a heavy tile all read
    // its count).  awaiting[p]: parity p was binned into and not zeroed since.  unrastered[p]: parity p was
    // binned into and its raster pass has not been launched — a binning pass that finds the OTHER parity so
    // (bins filled ahead for inputs that then changed, crender_prepare twice) starts over as well: the flag
    // and helper-slot words of the split tiles and the order hint exist once, not per parity.
    unsigned frame_no = 0;
    int parity = 0;               // of the last bin pass
    bool awaiting[2] = {false, false};
    bool unrastered[2] = {false, false};
    uint32_t *count(int par) const { return reinterpret_cast<uint32_t *>(ws + L.off_count) + (size_t)par * L.count_stride; }
    uint32_t *hflag() const { return reinterpret_cast<uint32_t *>(ws + L.off_hflag); }
    uint32_t *hslots() const { return reinterpret_cast<uint32_t *>(ws + L.off_hslots); }
    int hint_par = 0;             // order / hint buffer the next raster launch reads (it writes the other)
    float light[3] = {0.f, 0.f, 0.f};   // crender_plan_set_light (CRENDER_FUSED_GURO)
    const uint32_t *orig_of = nullptr, *pos_of = nullptr;   // crender_plan_set_triangle_order
    const float *normal_z = nullptr;                        // crender_plan_set_normal_z
    bool frame_lone = true;       // the last bin pass belonged to a frame rendered for latency (no
                                  // CRENDER_OVERLAPPED_FRAMES): ordered dispatch and split heavy tiles
    // how the last bin pass split long lists (run_bin_pass): helper triples in use (0: none), the list length that
    // registers a tile, the length from which its parts are quadrants instead of halves
    int frame_hmax = 0;
    uint32_t frame_heavy_at = 0, frame_quad_at = 0;
    // Bin-list usage of every frame without a host round trip (crender_plan_poll_bin_usage): each
    // raster launch copies the header words crender_plan_last_bin_usage reads into a record of its
    // own in PINNED host memory the plan owns — two 16-byte stores by one thread of the launch,
    // {frame number, hdr[0], hdr[1], hdr[4]} {large tiles, small tiles, kernel, frame number} — so that the host
    // learns of an overflow (and of the frames' size class) by reading its own memory: no copy command, no event,
    // no synchronisation.  Ring of kUsageRing frames.
    uint32_t *usage = nullptr;            // [kUsageRing + 1][kUsageWords] (the last one: staging of the blocking query)
    uint32_t *usage_dev = nullptr;        // the same memory as the device addresses it
    int usage_slot = -1;                  // its slot in the process-wide pool of pinned records (-1: an allocation of its own)
    uint32_t usage_salt = 0;              // XORed into the sequence word of this plan's records (slots are recycled)
    uint64_t ticket = 0;                  // raster launches so far: the last frame's number
    unsigned char usage_mode[kUsageRing] = {};   // how the frame of each record was binned: 0 scan, 1 direct bins, 2 pair bins
    // which raster kernel a 32-pixel plan's frames get (raster.hip, kPath*): forced by the caller (-1: not),
    // else suggested by the size classes its last reported frame counted; what the last launch was
    int forced_path = -1, auto_path = 0, last_path = 0;
    bool last_ordered = false;            // the last raster launch left a dispatch order (crender_plan_debug_check)
    bool auto_known = false;              // auto_path comes from a record (else from the triangle count per tile)
    uint64_t hint_ticket = 0;             // the launch whose record was taken last
    uint32_t *stats(int par) const { return reinterpret_cast<uint32_t *>(ws + L.off_stats) + (size_t)par * kStatWords; }
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
    uint2 *entry_pairs() const { return reinterpret_cast<uint2 *>(ws + L.off_entries); }   // (with a triangle order)
    uint2 *pairbins() const { return reinterpret_cast<uint2 *>(ws + L.off_pairbins); }
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
    // which raster kernel the chain's frames get when left to the plans (kPath*, raster.hip): the newest
    // opinion any of the chain's plans has read from its records — they all render the same stream of frames
    int shared_path = 0;
    bool shared_known = false;
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

namespace crender_detail {

#define CR_BY_TILE(call16, call32, call64) \
    (plan->L.ts == 16 ? (call16) : plan->L.ts == 32 ? (call32) : (call64))
// Frame = bin pass (K1 + binning into the plan) + raster pass (K2 from the plan's bins).
// `defer` (crender_pipeline's look-ahead): when the pass is the one-launch direct-bin kernel its
// arguments are handed back instead of launched, for k_frame to run it inside a raster launch.
int bin_pass(crender_plan *plan, bool project, const float *d_tri, const float *d_nrm, int64_t T,
             const float *P16, unsigned flags, void *stream, SetupArgs *defer = nullptr,
             bool *deferred = nullptr);
bool raster_path_hint(crender_plan *plan);
int raster_pass(crender_plan *plan, const float *proj, const float *d_col, const float *d_nrm, int64_t T,
                float *d_z, float *d_color, float *d_normal, int32_t *d_winner, unsigned flags,
                void *stream, const SetupArgs *with_setup = nullptr);

}  // namespace crender_detail
