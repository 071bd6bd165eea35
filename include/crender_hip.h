/*
 * crender_hip.h — C ABI of the MI355X (gfx950) rasterizer.
 *
 * Drop-in boundary for ONE path of oKatanaaa/Cython3DModelRenderer: the Version-C
 * filler `AdvancedPixelBufferFiller.render_model(model)`.  Reference file cited
 * below as ".pyx" = crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx,
 * "mu.pyx" = crender/cy/pixel_buffer_filler/math_utils.pyx.
 *
 * Conventions
 *   - plain C: raw pointers, sizes, an opaque plan handle; no torch / C++ types.
 *   - every `d_*` pointer is DEVICE memory (hipMalloc'ed or a torch-ROCm tensor's
 *     data_ptr()); `P16` and every other pointer is HOST memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All
 *     work is enqueued on it; nothing synchronises unless stated.
 *   - triangle arrays are C-contiguous float32 [T][3][3] (the layout the reference
 *     binds at .pyx:94-96 after `.copy()`): 9 floats per triangle, 36 bytes.
 *   - framebuffers are C-contiguous float32: z [H][W], colour [H][W][3] (BGR 0..255),
 *     normal [H][W][3]  (.pyx:65-67).  They always address the FULL frame; a call only
 *     touches rows y0 <= y < y1 (row strips for multi-GPU; y0 = 0, y1 = H otherwise).
 *   - every entry point returns CRENDER_OK (0) or an error code and never throws;
 *     crender_last_error() gives the text for the calling thread.
 *
 * Result contract: after crender_raster / crender_render_model the buffers hold, bit
 * for bit, what the reference's 1-thread loop leaves:
 * per pixel the fragment with the smallest z among all fragments and the prior buffer
 * value; equal z -> the highest triangle index; a fragment equal to the prior value
 * overwrites it.
 */
#ifndef CRENDER_HIP_H
#define CRENDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRENDER_ABI_VERSION 6
#define CRENDER_API __attribute__((visibility("default")))

enum {
    CRENDER_OK = 0,
    CRENDER_EINVAL = 1,  /* bad argument (null pointer, negative size, bad strip) */
    CRENDER_EHIP = 2,    /* HIP runtime error; text in crender_last_error()       */
    CRENDER_ENOMEM = 3,  /* workspace smaller than crender_plan_workspace_bytes   */
    CRENDER_EBUSY = 4,   /* crender_plan_poll_bin_usage: that frame's record has not landed yet */
    CRENDER_ESTATE = 5   /* crender_plan_debug_check: the plan's cross-frame state is inconsistent */
};

/* flags of crender_raster / crender_render_model / crender_raster_atomic */
enum {
    /* Treat the strip as freshly initialised (z = 1e6, colour = normal = 0, the state
     * __cinit__ leaves, .pyx:65-67) instead of reading it, and write every pixel of
     * the strip exactly once.  Equivalent to crender_clear followed by a flag-less
     * call, without the extra pass over the framebuffer. */
    CRENDER_FUSED_CLEAR = 1u,
    /* Always bin through the count / scan / fill passes.  Without it, scenes are binned in ONE pass
     * by appending straight into fixed-capacity per-tile lists — of 48-byte entries that carry the
     * projected triangle for scenes of up to 65536 triangles (the "direct bins"), of 8-byte (position,
     * index) pairs for larger ones (the "pair bins", three times the mean list per tile) — one launch
     * instead of three; a frame that does not fit reports so through crender_plan_last_bin_usage /
     * crender_plan_poll_bin_usage, and the plan then uses the general path by itself. */
    CRENDER_NO_DIRECT_BINS = 2u,
    /* This frame is one of several in flight on the GPU (a swap chain: crender_pipeline_*).  On
     * frames up to 1024 x 1024 a lone frame is rendered for latency: its raster launch starts the
     * tiles the previous frame on this plan found covered first, clears the others in groups, and
     * splits tiles with long lists over several workgroups — all covered tiles then hold a
     * workgroup slot from the first microsecond to the last, which is the shortest a single frame
     * gets and leaves no room for a second frame's launch to overlap it.  With this flag the
     * launch keeps plain raster order and one workgroup per tile — but for its LONGEST lists (from 32
     * records on in two halves, from 64 in four quadrants, at most 128 tiles), so that a frame that
     * runs with few others, at the start and the end of a burst, does not end on them (more frames per
     * second, longer frame).  Pass the same value to crender_prepare and crender_draw (the draw follows
     * what the prepare decided).  Results do not depend on it. */
    CRENDER_OVERLAPPED_FRAMES = 4u,
    /* next row f1 fused: apply GuroIllumination.draw_illumination
     * (guro_illumination.py:20-27) to every pixel as it is stored — colour *= clip(n.l / (|n| +
     * 1e-6), 0, 1) with the light vector of crender_plan_set_light, in numpy's float32 operation
     * order — instead of in a second pass over the colour and normal planes (36 B/pixel).  Only
     * with CRENDER_FUSED_CLEAR: the reference shades the WHOLE buffer after every render, which a
     * frame can only reproduce pixel by pixel when it starts from cleared buffers (background
     * colour 0 stays 0).  Same result as the render followed by crender_guro_illumination. */
    CRENDER_FUSED_GURO = 8u,
    /* crender_pipeline_* with look-ahead only: the CONTENTS of the triangle arrays do not change
     * for as long as the same arguments keep coming (e.g. a private device copy of a model), so
     * what was binned ahead for a slot stays valid across crender_pipeline_join.  Without it a join
     * voids it (the caller may have written new inputs behind the join) and every slot's first frame
     * afterwards bins in a launch of its own.  crender_pipeline_bind always voids its slot. */
    CRENDER_STATIC_INPUTS = 16u
};

CRENDER_API int crender_abi_version(void);
CRENDER_API const char *crender_last_error(void);

/* replaces: __cinit__ scalars + _init_projection_matrix (.pyx:54-59, 83-90).
 * Writes the row-major float32 4x4 matrix P = [[f/a,0,0,0],[0,f,0,0],[0,0,q,1],
 * [0,0,-z_near*q,0]] with the reference's float32 rounding steps. */
CRENDER_API int crender_projection_matrix(double fov_deg, double z_near, double z_far,
                              int h, int w, float *P16);

/* replaces: project_on_screen_multithread (K1, .pyx:106-130).
 * d_tri_out may equal d_tri_in (the reference always projects in place, .pyx:99). */
CRENDER_API int crender_project(const float *d_tri_in, float *d_tri_out, int64_t T,
                    const float *P16, int w, int h, void *stream);

/* replaces: the buffer initialisation of __cinit__ (.pyx:65-67) for rows [y0, y1).
 * d_winner (int32 [H][W], optional) is set to -1. */
CRENDER_API int crender_clear(float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                  int H, int W, int y0, int y1, void *stream);

/* ---- plan: frame geometry + carve-up of a caller-owned device workspace ----------
 * The tile rasterizer bins triangles into screen tiles; a plan fixes the strip, the
 * tile size and the capacity of the bin lists inside `d_workspace`.  The workspace
 * must stay allocated and untouched by others for the plan's lifetime.  A plan (and a
 * pipeline) belongs to the device that is current when it is created; calls using it must
 * be made with that device current. */
typedef struct crender_plan crender_plan;

/* tile: 0 = automatic (16 for frames up to 1024 x 1024, else 32), or 16, 32, 64.  bin_capacity: number of (tile, triangle) list
 * entries to reserve; 0 = automatic (4 per triangle + slack). */
CRENDER_API size_t crender_plan_workspace_bytes(int H, int W, int y0, int y1, int64_t max_T,
                                    int64_t bin_capacity, int tile);
CRENDER_API int crender_plan_create(crender_plan **out, int H, int W, int y0, int y1,
                        int64_t max_T, int64_t bin_capacity, int tile,
                        void *d_workspace, size_t workspace_bytes, void *stream);
/* Destroying a plan does not wait for the device: launches of the plan still in flight finish on their own
 * (the caller keeps the workspace valid until then, e.g. by freeing it in stream order); the pinned records
 * they write into (crender_plan_poll_bin_usage) belong to a process-wide pool and are not handed to another plan
 * until 1 023 others have been served. */
CRENDER_API void crender_plan_destroy(crender_plan *plan);

/* Synchronises `stream`, then reports the number of bin-list entries the most recent
 * crender_raster / crender_render_model on this plan needed and the capacity it had.
 * needed > capacity means that frame dropped fragments and must be rendered again:
 *   - if crender_plan_last_frame_direct(plan) is 1 the frame used fixed-capacity bins (direct bins or
 *     pair bins: figures are then per tile); the plan has switched itself to the general path, just
 *     re-render;
 *   - otherwise recreate the plan with bin_capacity >= needed first. */
CRENDER_API int crender_plan_last_bin_usage(crender_plan *plan, void *stream,
                                int64_t *needed, int64_t *capacity);

CRENDER_API int crender_plan_last_frame_direct(crender_plan *plan);
/* How the most recent frame was binned: 0 = count / scan / fill, 1 = direct bins, 2 = pair bins
 * (crender_plan_last_frame_direct is 1 for both kinds of fixed-capacity bins). */
CRENDER_API int crender_plan_last_frame_binning(crender_plan *plan);

/* The same figures WITHOUT a host round trip, per frame (no reference counterpart: the reference's
 * render_model, .pyx:92-104, returns when the buffers are written; a caller of this library that wants
 * to return before the GPU is done needs to learn of a dropped frame afterwards).  Every raster launch
 * on a plan (crender_raster / crender_render_model / crender_draw / a pipeline's frames) is numbered,
 * from 1, and leaves its figures in pinned host memory the plan owns, written by the launch itself —
 * no copy command, no event.
 *   crender_plan_frame_ticket    number of the most recent raster launch on the plan (0: none yet)
 *   crender_plan_poll_bin_usage  CRENDER_OK and the figures of frame `ticket` (as crender_plan_last_bin_usage
 *                                reports them, with the same switch of an overflowed direct-bin plan to
 *                                the general path) once its record has landed; CRENDER_EBUSY while it
 *                                has not — nothing is waited for, nothing is enqueued; CRENDER_EINVAL
 *                                for a frame never launched or older than the last 8.
 * Once `stream` of that launch has been synchronised by any means the record is there. */
/* CRENDER_OK means "this frame's binning figures are known" — the record is written when the raster launch
 * STARTS its last workgroup (the binning pass ran in an earlier launch), not when the framebuffers are
 * written: it is no completion signal.  A record is two aligned 16-byte stores with the frame's sequence
 * word first and last; it is taken only when both are there.  On a swap chain with look-ahead the NEXT
 * frame's binning wavefronts run inside this launch and raise the same sticky overflow word: an overflow
 * of theirs may be reported one frame early (conservative: the frame is rendered again). */
CRENDER_API uint64_t crender_plan_frame_ticket(crender_plan *plan);
CRENDER_API int crender_plan_poll_bin_usage(crender_plan *plan, uint64_t ticket, int64_t *needed,
                                            int64_t *capacity);

/* Tile-coherent triangle order (no reference counterpart: a data-layout choice for scenes of
 * millions of small triangles, where gathering unsorted 36-byte records dominates the frame).
 * The caller may hand the kernels a PERMUTATION of its triangle arrays (all three alike), sorted
 * so that triangles of one screen region are neighbours in memory — crender_tile_order_keys
 * writes a sort key per triangle (Morton code of the 32-pixel tile of its projected centroid) —
 * (the key is in fact the Morton code of the 4-pixel CELL, whose prefixes are the codes of the 8- to
 * 64-pixel tiles: neighbours in the frame stay neighbours in memory within a tile too) —
 * and tell the plan about it: d_orig_of[position] = index in the caller's own arrays,
 * d_pos_of[index] = position (uint32 [T] each, device memory that outlives the plan's frames;
 * NULL, NULL = no permutation).  Results are unchanged: depth ties still go to the highest index
 * of the caller's arrays and the winner plane reports those indices. */
CRENDER_API int crender_tile_order_keys(const float *d_tri, int64_t T, const float *P16, int w, int h,
                                        uint32_t *d_keys, void *stream);
CRENDER_API int crender_plan_set_triangle_order(crender_plan *plan, const uint32_t *d_orig_of,
                                                const uint32_t *d_pos_of);

/* Normal z components apart (no reference counterpart: a data-layout choice like the triangle order).
 * The back-face test of .pyx:202 reads the z component of a triangle's three vertex normals — 12 of
 * the 36 bytes of its d_nrm record, at a stride that drags every line of the array through the
 * binning pass (360 MB per frame of 10 M triangles, a third of that pass's traffic).  A caller that
 * keeps a resident model may hand the plan those components as an array of their own:
 * d_nz float32 [T][3] = d_nrm[:, :, 2], in the order of the triangle arrays the frames are given
 * (device memory that outlives the plan's frames; NULL = read d_nrm).  The test itself — the float
 * sum and its comparison — still runs every frame, on the same values: results are unchanged.
 * Used by the scan path's binning pass (scenes beyond the direct bins). */
CRENDER_API int crender_plan_set_normal_z(crender_plan *plan, const float *d_nz);

/* Which raster kernel a plan's frames get (no reference counterpart: the reference has one loop nest,
 * .pyx:196-244, and so has every kernel here — the choice is about speed only).  Plans on 32-pixel tiles
 * have two kernels that both render EVERY tile exactly (the parity tests force each through every scene):
 *   0  the general one, every sweep in it (6 wavefronts per SIMD);
 *   1  the pixel owners alone — frames of large triangles (bunny 4096^2, T-Rex 8192^2): no key plane, 19.5 KB
 *      of LDS, 7 wavefronts per SIMD.
 * -1 (the default) lets the plan choose: every raster launch counts its covered tiles by the size of their
 * records and leaves the counts in the launch's usage record (crender_plan_poll_bin_usage's pinned memory);
 * a plan reads its own (a swap chain: its plans') last landed record before a launch — no copy, no
 * synchronisation — and takes 1 when three quarters of the covered tiles hold large records, else 0; until a
 * record has landed, the triangle count per tile decides (few triangles on many tiles are large ones).
 * Other tile sizes have the general kernel only and ignore the setting.
 *   crender_plan_set_raster_path      fix the kernel of this plan's frames, or hand the choice back (-1)
 *   crender_plan_last_raster_path     which kernel the most recent raster launch of the plan was
 *   crender_set_default_raster_path   process-wide: what plans on -1 take instead of choosing (-1: choose);
 *                                     the parity tests run the whole suite under each value */
CRENDER_API int crender_plan_set_raster_path(crender_plan *plan, int path);
CRENDER_API int crender_plan_last_raster_path(crender_plan *plan);
CRENDER_API int crender_set_default_raster_path(int path);

/* Light direction (host pointer to 3 floats, copied) for frames rendered with CRENDER_FUSED_GURO:
 * GuroIllumination.__init__'s light_direction (guro_illumination.py:6-18). */
CRENDER_API int crender_plan_set_light(crender_plan *plan, const float *light3);

/* Diagnostics (no reference counterpart: the reference keeps no state between render_model calls but its
 * buffers, .pyx:92-104).  A plan does: two parities of per-tile counters, the split tiles' flag and helper-slot
 * words, registration / hint_bad counters, the dispatch order, the ring of usage records.  This synchronises
 * `stream`, reads that state back and checks its invariants for the moment between two frames — after a
 * raster launch: the next frame's counters, registration counter and hint_bad all zero, no split flag and
 * no helper slot left set; after a binning pass alone (crender_prepare, a swap chain's look-ahead): helper
 * triples = registrations = tiles whose list reached the threshold, each flagged once; always: the dispatch
 * order is a permutation that agrees with its header and `grouped` bytes, every landed usage record is one of
 * this plan's launches, in its slot, leading and trailing sequence words alike.  CRENDER_OK, or CRENDER_ESTATE
 * with the findings as text in `msg` (may be NULL).  Not for the frame path: it copies the plan's head. */
CRENDER_API int crender_plan_debug_check(crender_plan *plan, void *stream, char *msg, size_t msg_bytes);

/* Measurement aid (no reference counterpart): record HIP events on the frame's own
 * stream around the binning passes and around the raster kernel of each of the next
 * `max_frames` frames; crender_plan_timing_end synchronises and returns the averages. */
CRENDER_API int crender_plan_timing_begin(crender_plan *plan, int max_frames);
CRENDER_API int crender_plan_timing_end(crender_plan *plan, void *stream, int *frames,
                            double *bin_ms_avg, double *raster_ms_avg);

/* replaces: compute_triangle_statistics_multithread (K2, .pyx:177-244) including
 * _compute_pixel_coords_c (.pyx:132-175) and compute_bar_coords_single_pixel
 * (mu.pyx:8-34).  d_tri_proj holds K1's output.  d_winner (optional, int32 [H][W])
 * receives the index of the triangle whose fragment each written pixel holds.
 * Tile-binned LDS rasterizer (the production path). */
CRENDER_API int crender_raster(crender_plan *plan, const float *d_tri_proj, const float *d_col,
                   const float *d_nrm, int64_t T,
                   float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                   unsigned flags, void *stream);

/* replaces: render_model (.pyx:92-104) = K1 into the plan's scratch + K2, with the
 * projection fused into the binning pass.  d_tri is the UNPROJECTED vertex array and
 * is not modified. */
CRENDER_API int crender_render_model(crender_plan *plan, const float *d_tri, const float *d_col,
                         const float *d_nrm, int64_t T, const float *P16,
                         float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                         unsigned flags, void *stream);

/* crender_render_model in two halves, so that a caller can overlap the first half of the
 * next frame with the second half of the current one on another stream (two plans, double
 * buffering; crender_pipeline_* below is the ready-made version with whole frames in flight):
 *   crender_prepare  K1 + binning into the plan.  P16 == NULL: d_tri is already projected.
 *   crender_draw     K2 from the plan's bins.  d_tri_proj == NULL: use the vertices
 *                    crender_prepare projected into the plan; T must equal the prepared T.
 * The caller orders the two calls (same stream, or an event between streams) and must not
 * prepare into a plan whose previous crender_draw has not finished. */
CRENDER_API int crender_prepare(crender_plan *plan, const float *d_tri, const float *d_nrm, int64_t T,
                    const float *P16, unsigned flags, void *stream);
CRENDER_API int crender_draw(crender_plan *plan, const float *d_tri_proj, const float *d_col,
                 const float *d_nrm, int64_t T, float *d_z, float *d_color, float *d_normal,
                 int32_t *d_winner, unsigned flags, void *stream);

/* A swap chain, ready-made: one call per frame, up to `depth` (1..8) frames overlap on the GPU
 * (depth 1: consecutive frames on ONE stream — with look-ahead each of them a single launch).
 * The pipeline owns `depth` streams; frame i runs (bin pass + raster pass) on stream i % depth
 * with plans[i % depth].  FRAMES IN FLIGHT MUST TARGET DIFFERENT FRAMEBUFFER SETS (a swap
 * chain): nothing orders frame i against frames i+1 .. i+depth-1.  Frames i and i + depth share
 * a stream and may share buffers.  Each frame's result equals crender_render_model's with the
 * same arguments.
 *   - `stream` is the CALLER's stream: at the first frame after creation or a join, and when the
 *     input pointers, T or the stream change, the pipeline's streams first wait for everything
 *     enqueued on it (uploads, earlier use of the framebuffers).  If inputs are overwritten IN
 *     PLACE, call crender_pipeline_join first.
 *   - crender_pipeline_join makes `stream` wait for all submitted frames and restarts the frame
 *     counter at 0.  Call it before a framebuffer is read or written by anything else, before
 *     using the plans directly and before destroying them.  Bin-list overflow is queried per
 *     plan as usual, after a join.
 * No HIP event separates frames: an event record + cross-stream wait costs a 7-12 us bubble on
 * MI355X, which is why frames are decoupled by buffers instead of chained. */
typedef struct crender_pipeline crender_pipeline;
CRENDER_API int crender_pipeline_create(crender_pipeline **out, crender_plan *const *plans, int depth);
CRENDER_API void crender_pipeline_destroy(crender_pipeline *pipeline);
CRENDER_API int crender_pipeline_frame(crender_pipeline *pipeline, const float *d_tri, const float *d_col,
                           const float *d_nrm, int64_t T, const float *P16,
                           float *d_z, float *d_color, float *d_normal, int32_t *d_winner,
                           unsigned flags, void *stream);
CRENDER_API int crender_pipeline_join(crender_pipeline *pipeline, void *stream);
/* Look-ahead: give every slot a second plan (`plans`: `depth` plans, each like the slot's first,
 * distinct from all others; n = 0 or plans = NULL switches it off; only between joins).  Slot k then
 * alternates between its two plans, and the launch that rasterizes frame i also bins the slot's next
 * frame (i + depth) — expected to arrive with the same inputs, projection matrix and flags — into the
 * other plan: one launch per frame instead of two, and the binning pass (a latency chain of a few
 * hundred wavefronts) costs no launch of its own.  A frame that arrives with other arguments, and
 * every slot's first frame after a join (the caller may have written new inputs), bins in a launch
 * of its own first, exactly as without look-ahead.  Scenes on the count / scan / fill path gain
 * nothing (their passes are launched as before).  Bin-list usage is queried per plan, all 2 x depth. */
CRENDER_API int crender_pipeline_set_lookahead(crender_pipeline *pipeline, crender_plan *const *plans, int n);
/* The same with the per-frame arguments bound up front, for callers whose call overhead grows with
 * the argument count (ctypes: ~4 us for crender_pipeline_frame's twelve): bind slot k = 0..depth-1
 * (the arguments frame i with i % depth == k is to use; P16 is copied), then every
 * crender_pipeline_submit renders the next frame of the rotation exactly as crender_pipeline_frame
 * would with that slot's arguments.  Submitting an unbound slot is CRENDER_EINVAL. */
CRENDER_API int crender_pipeline_bind(crender_pipeline *pipeline, int slot, const float *d_tri,
                                      const float *d_col, const float *d_nrm, int64_t T,
                                      const float *P16, float *d_z, float *d_color, float *d_normal,
                                      int32_t *d_winner, unsigned flags);
CRENDER_API int crender_pipeline_submit(crender_pipeline *pipeline, void *stream);
/* Measurement aid (no reference counterpart), the swap chain's counterpart of
 * crender_plan_timing_*: HIP events on the frame's OWN stream right before and right after the
 * launch(es) of each of the next `max_frames` frames; crender_pipeline_timing_end synchronises
 * the pipeline's streams and returns the average.  With look-ahead and depth 1 that is the
 * duration of one k_frame launch running alone — the kernel a pipelined stream of frames runs. */
CRENDER_API int crender_pipeline_timing_begin(crender_pipeline *pipeline, int max_frames);
CRENDER_API int crender_pipeline_timing_end(crender_pipeline *pipeline, int *frames, double *launch_ms_avg);

/* Same contract as crender_raster, computed a second, independent way: one wavefront
 * per triangle, 64-bit global atomics on a packed (z, index) key plane, then a
 * per-pixel resolve.  Needs no plan; d_keys is caller-owned scratch of
 * crender_atomic_scratch_bytes(H, W) bytes.  Used to cross-check the tile path at
 * sizes where the CPU oracle is slow. */
CRENDER_API size_t crender_atomic_scratch_bytes(int H, int W);
CRENDER_API int crender_raster_atomic(const float *d_tri_proj, const float *d_col, const float *d_nrm,
                          int64_t T, float *d_z, float *d_color, float *d_normal,
                          int32_t *d_winner, int H, int W, int y0, int y1,
                          unsigned flags, void *d_keys, void *stream);

/* Self-check hook (no reference counterpart).  The sweep divides many numerators by one
 * per-triangle denominator and hoists the reciprocal refinement of hipcc's own division
 * expansion out of the loop (csrc/raster_math.h, "shortcut (2)").  This entry point evaluates
 * that shortcut (out_tail; plain division outside its operand window) and the plain `/`
 * (out_div) element-wise so a test can require the two to be bit-identical. */
CRENDER_API int crender_selfcheck_division(const float *d_num, const float *d_den, float *d_out_tail,
                               float *d_out_div, int64_t n, void *stream);

/* next row f1: GuroIllumination.draw_illumination
 * (crender/cy/illumination/guro_illumination.py:20-27) on device, in place on the
 * colour buffer: colour *= clip(n.l / (|n| + 1e-6), 0, 1), rows [y0, y1). */
CRENDER_API int crender_guro_illumination(float *d_color, const float *d_normal, const float *light3,
                              int H, int W, int y0, int y1, void *stream);

/* next row f3: presentation — run.py:26 `image[::-1].astype('uint8')`: float32 colour plane
 * [H][W][3] -> uint8 [H][W][3] with numpy's C-cast semantics, rows flipped if flip_rows. */
CRENDER_API int crender_present_u8(const float *d_color, unsigned char *d_out, int H, int W,
                       int flip_rows, void *stream);

/* next row f2: the Model's reproducible transforms on a device-resident
 * vertex buffer d_vertices = float32 [V][3] — crender/cy/data_structures/model.py:153-236.
 *   crender_model_shift   Model.shift (:213-216): vertices + shift.  shift_is_float32 = 1: the
 *                         caller's shift was a float32 array (float32 sum); 0: a Python list / a
 *                         float64 array (numpy promotes the sum to float64, stored as float32).
 *   crender_model_scale   Model.scale (:218-236): vtx -= mean; vtx *= coef; vtx += mean in float32
 *                         (keep_position), or vtx *= coef.  d_mean3: DEVICE pointer to 3 floats.
 *   crender_model_stats   _update_vertices_and_normals (:153-160): mean vertex (numpy's float32
 *                         row-after-row sum, float64 division) and max span (max float32 norm of
 *                         vertices - mean) into DEVICE memory d_mean3 [3], d_max_span [1].
 *   crender_model_gather  attr[index] (:156, :172, :151): d_attr float32 [N][3], d_index int32
 *                         [T][3] -> d_out float32 [T][3][3]: the filler's input arrays.
 *   crender_model_texture_colors  next row f4, its device part (:143-151): one float32 BGR
 *                         colour per texture coordinate — nearest texel, v axis flipped, numpy's
 *                         float32 arithmetic and truncating int32 cast.  d_uv float32 [n][uv_cols]
 *                         (u, v first; uv_cols >= 2), d_texture uint8 [th][tw][3] -> d_out float32
 *                         [n][3]; crender_model_gather with the faces' texture indices then gives
 *                         colors_by_triangles.  (Parsing .obj / .mtl text stays on the host.)
 *   crender_model_rotate  Model.rotate (:238-256), the matrix product: every coordinate the
 *                         three-term float64 dot product of the float32 vertex with a row of R9 (the
 *                         host composes mat_rot in float64 as the reference does; row-major 3 x 3),
 *                         rounded to float32, in place.
 *   crender_model_vertex_normals  Model._compute_normals_by_vertex (:175-208): unit face normals
 *                         into d_face_normals [T][3], then per vertex the normalised mean of the
 *                         distinct ones among the faces that name it, visited in face order
 *                         (d_offsets int32 [V + 1], d_occurrences int32 [3 T]: per vertex the faces of
 *                         its (face, corner) occurrences, ascending; d_taken: 3 T bytes of scratch).
 *                         d_faces int32 [T][3] with non-negative indices.
 * numpy takes np.linalg.norm / np.dot / matmul from its BLAS build, so "what the reference computes"
 * is a property of that build, not of numpy.  The kernels match numpy built on OpenBLAS (verified:
 * scipy-openblas 0.3.29, x86-64): float32 products added in a double accumulator and rounded once
 * (its sdot's tail loop), a three-term float64 sum for the rotation — spelled out as such, and bit-identical
 * to the host Model on every mesh of tests/test_hip_parity_gpu.py::test_device_model_rotate_and_normals
 * under that build (tests/test_host_cpu.py::test_numpy_dot_of_3_vectors guards the assumption; on
 * another BLAS parity of this row is unpinned and the GPU test falls back to a bounded check)
 * (the de-duplication test `dot >= 1` is discontinuous: a 1-ulp difference in the dot would change
 * which face normals a vertex averages, as rounds 1-3's plain float32 sum did on 0.8 % of T-Rex's
 * vertices). */
CRENDER_API int crender_model_rotate(float *d_vertices, int64_t V, const double *R9, void *stream);
CRENDER_API int crender_model_vertex_normals(const float *d_vertices, int64_t V, const int32_t *d_faces, int64_t T,
                                             const int32_t *d_offsets, const int32_t *d_occurrences,
                                             float *d_face_normals, unsigned char *d_taken, float *d_normals,
                                             void *stream);
CRENDER_API int crender_model_shift(float *d_vertices, int64_t V, const double *shift3,
                                    int shift_is_float32, void *stream);
CRENDER_API int crender_model_scale(float *d_vertices, int64_t V, const float *d_mean3, float coef,
                                    int keep_position, void *stream);
CRENDER_API int crender_model_stats(const float *d_vertices, int64_t V, float *d_mean3,
                                    float *d_max_span, void *stream);
CRENDER_API int crender_model_gather(const float *d_attr, const int32_t *d_index, float *d_out,
                                     int64_t T, void *stream);
CRENDER_API int crender_model_texture_colors(const float *d_uv, int uv_cols, int64_t n,
                                             const unsigned char *d_texture, int th, int tw,
                                             float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CRENDER_HIP_H */
