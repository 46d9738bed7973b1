/*
 * gsplat_amd.h -- C ABI of the MI355X-native (gfx950) differentiable 3D-Gaussian rasterizer.
 *
 * Drop-in boundary
 * ----------------
 * The reference has no FFI of its own on this path: its whole hot path is one Python call,
 *     gsplat.rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
 *                          packed=False, render_mode=..., sh_degree=..., backgrounds=...)
 * made at /root/reference/src/backends/gsplat_backend.py:168-182 (live) and :83-97 (dead),
 * behind the plugin method GSplatBackend.render_with_depth (gsplat_backend.py:129-210).
 * The entry points below are the stages of that one call;
 * the Python host side (3d-gaussian-splatting_amd/gsplat_amd/) binds them with ctypes and
 * re-assembles the reference's operator and plugin surface on top.
 *
 * Conventions
 * -----------
 *  - extern "C", plain C types only; no torch / C++ types cross this boundary.
 *  - Every pointer is a DEVICE pointer unless its comment says "host".
 *  - Every function takes the hipStream_t to launch on (as void*; pass
 *    torch.cuda.current_stream().cuda_stream) and is asynchronous with respect to the host.
 *  - Every function returns int: 0 = success, > 0 = a hipError_t, < 0 = a GSA_ERR_* code.
 *    Nothing throws, nothing calls exit(), nothing allocates: all scratch is caller-provided
 *    (sizes from the *_workspace_bytes queries) so the caller's caching allocator owns it.
 *  - Re-entrant and thread-compatible: no mutable globals.
 *  - All floating tensors are float32 and contiguous; indices are int32/uint32.
 *
 * Per-gaussian "splat record" (GSA_SPLAT_STRIDE floats = 64 B, one cache-line-sized row per gaussian;
 * written by gsa_project_fwd, gathered by the compositor; the gradient record has the same layout):
 *    [0] mean2d.x  [1] mean2d.y  [2] conic a  [3] conic b  [4] conic c  [5] opacity
 *    [6] r  [7] g  [8] b  [9] depth (camera z)  [10] ln(255 * opacity) (the largest sigma with alpha >= 1/255;
 *    zero in the gradient record)  [11..15] unused (zero)
 */
#ifndef GSPLAT_AMD_H
#define GSPLAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "gsplat_amd_constants.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GSA_ABI_VERSION 14
#define GSA_SPLAT_STRIDE 16

#define GSA_OK 0
#define GSA_ERR_INVALID_ARGUMENT (-1) /* null pointer, negative extent, unsupported SH degree ...   */
#define GSA_ERR_WORKSPACE_TOO_SMALL (-2)
#define GSA_ERR_UNSUPPORTED (-3)      /* e.g. more tiles than the sort key holds                    */

typedef void *gsa_stream_t; /* hipStream_t */

/* ABI version of the loaded library (== GSA_ABI_VERSION of the header it was built from). */
int gsa_abi_version(void);

/* Static, human-readable text for a return code of this library (host pointer, never freed). */
const char *gsa_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Stage 1 (SURVEY 8a rows a3.1 + a3.2): projection + EWA covariance + SH colour + tile count.
 * Replaces gsplat's projection and spherical-harmonics kernels behind gsplat_backend.py:168-182.
 *
 *  means[N,3] quats[N,4](w first) scales[N,3] opacities[N]
 *  colors: sh_degree >= 0 -> SH coefficients [N, sh_coeffs, 3] (first (sh_degree+1)^2 used);
 *          sh_degree == -1 -> post-activation colours [N,3] used as they are.
 *  viewmat[16] row-major world->camera, intrin[9] row-major K (both on the device).
 *  cull_mode: 0 = tile rectangle of (mean2d +- radius), exactly the published rule;
 *             1 = additionally drop tiles whose pixel centres the alpha >= 1/255 ellipse cannot reach
 *                 (same image and gradients, fewer records).
 * Outputs: splat[N,16] (record above; zero rows for culled gaussians), radii[N,2] int32 (0 = culled),
 *          tiles_touched[N] int32, depth_keys[N] uint32 (float bits of depth; 0xFFFFFFFF if culled),
 *          tile_rects[N,2] uint32 (optional, may be NULL; 8-byte aligned): per gaussian the tile rectangle packed
 *          as x0 | y0 << 10 | width << 20 (what record word 11 holds; 0 without records or for tile grids of 1024+
 *          tiles a side) and the record count (== tiles_touched) -- the dense input of the fused binning
 *          (gsa_scan_tiles_rect).
 *          sh_aux[N,12] float (optional, may be NULL; 16-byte aligned; written for sh_degree >= 1, zero rows for
 *          culled gaussians): what stage 5 needs of the 192-B coefficient row, in 48 B -- per colour channel
 *          d colour / d mean through the view direction (x, y, z) and the clamp flag (1 = the channel was not clamped
 *          at 0).  Handed to gsa_project_bwd*, it spares that stage the coefficient rows.
 *  bin_tile: 0 / 16 -- tile_rects describes the 16 x 16 tiles; 32 (cull_mode 0 only) -- tile_rects describes 32 x 32 BINS
 *          (rectangle and count on the grid of ceil(width / 32) x ceil(height / 32) bins): about half as many records to
 *          sort, for a compositor that is told so (bin_tile of gsa_tile_bin16 / gsa_tile_schedule / gsa_composite_*:
 *          the four tiles of a bin walk the bin's list; a wave re-tests every record against its 8 x 8 quadrant anyway).
 *          splat word 11, radii and tiles_touched always describe the 16 x 16 tiles.
 *  key_range (ABI 7; optional, may be NULL; 16-byte aligned): 4 * ceil(n / 64) uint32 -- per wave of 64 consecutive
 *          gaussians the (min, max) of the depth keys of its VISIBLE members ((0xFFFFFFFF, 0) when it has none) and the
 *          keys of its members 0 and 32 (a sample; 0xFFFFFFFF if culled or beyond n); every entry is written.
 *          gsa_depth_order_msd(key_range_dev) builds the view's depth table from them on the device.
 *  qbox (ABI 10; optional, may be NULL): n uint32 -- per gaussian the 8 x 8 quadrants (pixel centres 8 q + 0.5 .. 8 q + 7.5
 *          a side) the bounding box of its alpha >= 1/255 ellipse covers: first column | first row << 10 | (columns - 1) << 20
 *          | (rows - 1) << 26; 0xFFFFFFFF = unbounded (more than 64 quadrants a side, grids beyond 1024); unspecified for
 *          culled gaussians.  gsa_tile_bin16 turns it into per-record quadrant boxes for gsa_composite_fwd.
 */
int gsa_project_fwd(int n, const float *means, const float *quats, const float *scales,
                    const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                    const float *viewmat, const float *intrin, int width, int height, int cull_mode,
                    float *splat, int32_t *radii, int32_t *tiles_touched, uint32_t *depth_keys,
                    uint32_t *tile_rects, float *sh_aux, int bin_tile, uint32_t *key_range, uint32_t *qbox,
                    gsa_stream_t stream);

/* ABI 12 -- the SAME gaussians seen by n_cams cameras in one launch: the views of one optimisation step (the reference
 * renders the B views of a batch one after the other from the same parameters, src/training/trainer.py:458-506; gsplat's
 * own operator takes viewmats [C,4,4]).  A gaussian's geometry and its 192-B coefficient row are read ONCE for all
 * cameras (236 of the 376 bytes a single-camera launch moves per gaussian).  viewmats: n_cams x 16 floats; intrins:
 * n_cams x 9 floats with intrin_stride = 9, or ONE 3 x 3 matrix shared by all cameras with intrin_stride = 0.  Every
 * output is an [n_cams, ...] array of per-camera slabs laid out exactly like gsa_project_fwd's outputs (camera c's slab
 * starts c * n rows in; key_range: c * 4 * ceil(n / 64) words in); optional outputs as there.  Camera c's slabs hold the
 * same bits a gsa_project_fwd call with camera c writes. */
int gsa_project_fwd_batch(int n, const float *means, const float *quats, const float *scales,
                          const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                          int n_cams, const float *viewmats, const float *intrins, int intrin_stride, int width,
                          int height, int cull_mode, float *splat, int32_t *radii, int32_t *tiles_touched,
                          uint32_t *depth_keys, uint32_t *tile_rects, float *sh_aux, int bin_tile,
                          uint32_t *key_range, uint32_t *qbox, gsa_stream_t stream);

/* ABI 13 -- the stage in PARTS, for callers that overlap them (n_cams >= 1; arguments as gsa_project_fwd_batch; n_cams = 1 with
 * intrin_stride 0 is gsa_project_fwd).
 *   GSA_PROJECT_GEOMETRY: only what the BINNING reads -- radii, tiles_touched, depth_keys, tile_rects, key_range, qbox
 *       (44 B in, 28 B out per gaussian and camera; colors / splat / sh_aux are not touched and may be NULL);
 *   GSA_PROJECT_RECORDS:  only what the SWEEPS and stage 5 read -- splat (with the colour) and sh_aux (the six geometry
 *       outputs are not touched and may be NULL);
 *   GSA_PROJECT_GEOMETRY | GSA_PROJECT_RECORDS: the whole stage.
 * Each part recomputes the projection from the inputs, so the two launches are independent: the records part (which streams
 * the 192-B coefficient rows) can run on a second stream beside the latency-bound binning chain of the same view, which only
 * waits for the short geometry part.  Together the parts write the bits the whole stage writes. */
#define GSA_PROJECT_GEOMETRY 1
#define GSA_PROJECT_RECORDS 2
int gsa_project_fwd_parts(int parts, int n, const float *means, const float *quats, const float *scales,
                          const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                          int n_cams, const float *viewmats, const float *intrins, int intrin_stride, int width,
                          int height, int cull_mode, float *splat, int32_t *radii, int32_t *tiles_touched,
                          uint32_t *depth_keys, uint32_t *tile_rects, float *sh_aux, int bin_tile,
                          uint32_t *key_range, uint32_t *qbox, gsa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 2 (row a3.3): tile binning.  Replaces gsplat's isect_tiles + 64-bit radix sort +
 * isect_offset_encode.  Same resulting order (tile, then depth bits, then gaussian index), built
 * as: stable sort of the N gaussians by depth bits; emit records in that order; stable sort of the
 * M records by tile id only.
 */
size_t gsa_sort_workspace_bytes(int n);

/* Stable LSD radix sort of n (key,value) uint32 pairs on bits [begin_bit, end_bit) of the key.
 * vals_in == NULL means value i = i.  keys_tmp/vals_tmp: n elements each.  in/out/tmp must not alias. */
int gsa_sort_pairs_u32(int n, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                       uint32_t *vals_out, uint32_t *keys_tmp, uint32_t *vals_tmp, int begin_bit,
                       int end_bit, void *workspace, size_t workspace_bytes, gsa_stream_t stream);

/* offsets[k] = sum_{j<k} tiles_touched[order[j]] (exclusive, in depth order); *total = M.
 * workspace: gsa_sort_workspace_bytes(n). */
int gsa_scan_tiles(int n, const int32_t *tiles_touched, const uint32_t *order, uint32_t *offsets,
                   uint32_t *total, void *workspace, size_t workspace_bytes, gsa_stream_t stream);

/* Fused binning for cull_mode 0 and 16-bit tile ids (gsa_tile_bin16_supported: tile grid below 1024 x 1024 and at
 * most 65535 tiles).  gsa_scan_tiles_rect is gsa_scan_tiles on the counts in gsa_project_fwd's tile_rects[N,2] that
 * also brings the packed rectangles into depth order -- rects[k] for g = order[k] -- and notes which gaussian owns the
 * first record slot of every block of GSA_BIN_BLOCK slots: block_owner[b] = last k with offsets[k] <= b *
 * GSA_BIN_BLOCK, for b < n_blocks (entries of blocks at or beyond the record count are not written), and
 * block_owner[n_blocks] = the last gaussian (in depth order) that has records: n_blocks + 1 entries.
 * gsa_tile_bin16 then produces what gsa_tile_emit + gsa_sort_pairs_u32 + gsa_tile_offsets produce -- sorted_tile
 * (as uint16), sorted_gid, tile_offsets[n_tiles + 1] -- without ever writing the records in emission order: the
 * first pass of the tile sort generates them from the rectangles.  m_dev: live record count on the device (NULL:
 * m_capacity is the count); block_owner: from a scan with n_blocks == ceil(m_capacity / GSA_BIN_BLOCK); tile_tmp / gid_tmp:
 * m_capacity elements each (unused for at most 256 tiles); workspace: gsa_sort_workspace_bytes(m_capacity).
 * ABI 8 -- the direct path.  For grids of 257 .. 2047 tiles / bins (gsa_tile_bin16_direct: the 60 x 34 bins of 32 x 32
 * pixels at 1920 x 1080) a caller that passes tile_tmp == gid_tmp == NULL and a workspace of
 * gsa_tile_bin16_workspace_bytes gets the lists from ONE placement pass: a per-block histogram over the bins
 * themselves, a column scan that also writes tile_offsets, and every gaussian id stored straight at its place in its
 * bin's list (no digit passes, no search for the list boundaries).  sorted_tile may then be NULL (not written): the
 * lists are sorted_gid + tile_offsets.  Same output, bit for bit, as the two-pass path.
 * ABI 10 -- quadrant boxes from the placement pass.  Where gsa_tile_bin16_boxes says 1 (the direct path, at most 2^24
 * gaussians) a caller may pass qbox (gsa_project_fwd's quadrant boxes): the TOP BYTE of every sorted_gid entry then holds
 * the part of the gaussian's box that lies inside the record's bin, in quadrants of the bin (bins of 16 or 32 pixels) or
 * in PAIRS of quadrants (64-pixel bins: 8 x 8 quadrants; ABI 11) -- since ABI 12 as two MASKS: bit 24 + c set: the box
 * covers column unit c (0..3) of the bin, bit 28 + r: row unit r; both nibbles zero: none -- so a wave tests "my quadrant"
 * with one AND and one compare -- and the id is the low 24 bits.  gsa_composite_fwd / _bwd / _bwd_lists must be told (gid_boxes = 1): the forward sweep then gathers only
 * the records whose box covers a wave's quadrant (a superset of the per-pixel rule: at the bench scene 78 % of the pairs it
 * lets through reach a pixel, where 22 % of a bin's records reach a given quadrant) -- for about fifteen integer
 * instructions per record in the placement and not one store more.  qbox == NULL: plain ids (gid_boxes = 0). */
#define GSA_BIN_BLOCK 4096
int gsa_tile_bin16_supported(int width, int height);
int gsa_tile_bin16_direct(int width, int height, int bin_tile);
int gsa_tile_bin16_boxes(int width, int height, int bin_tile, int n);
size_t gsa_tile_bin16_workspace_bytes(int width, int height, int bin_tile, uint32_t m_capacity);
/* ... for bins of bin_tile pixels (16 / 32 / 64): an 8K image has too many 16 x 16 tiles for 16-bit ids, but not 32 x 32 bins */
int gsa_tile_bins_supported(int width, int height, int bin_tile);
int gsa_scan_tiles_rect(int n, const uint32_t *order, const uint32_t *tile_rects, uint32_t *offsets,
                        uint32_t *rects, int32_t *block_owner, int n_blocks, uint32_t *total, void *workspace,
                        size_t workspace_bytes, gsa_stream_t stream);
/* ABI 5 -- depth ordering for the fused binning.  gsa_depth_order: stable LSD radix sort of the n gaussians by bits
 * [0, key_bits) of their depth keys (key_bits in 1..32, rounded up to whole 8-bit passes): key_bits is how many low
 * bits the caller expects the keys of the VISIBLE gaussians to differ in (culled gaussians carry 0xFFFFFFFF and fall
 * behind every visible key in each digit; they have no records, so their place among equal low bits is immaterial).
 * Besides order[n] it leaves, in depth order, rects[k] / counts[k] = tile_rects[order[k]] (what gsa_scan_tiles_rect
 * gathers), and key_range[0..1] = min / max key over the visible gaussians (0xFFFFFFFF / 0 when there is none): the
 * order is the full 32-bit order iff ((key_range[0] ^ key_range[1]) >> passes * 8) == 0 -- the caller checks that
 * after the fact (like the record capacity) and repeats with key_bits = 32 on a miss.
 * gsa_scan_tiles_sorted: the exclusive scan + block-owner table of gsa_scan_tiles_rect on counts that are already in
 * depth order (in place: counts in, offsets out).  workspace: gsa_depth_order_workspace_bytes(n) /
 * gsa_sort_workspace_bytes(n). */
size_t gsa_depth_order_workspace_bytes(int n);
int gsa_depth_order(int n, const uint32_t *depth_keys, const uint32_t *tile_rects, int key_bits, uint32_t *order,
                    uint32_t *rects, uint32_t *counts, uint32_t *key_range, void *workspace, size_t workspace_bytes,
                    gsa_stream_t stream);
/* gsa_depth_order_msd: the same outputs as gsa_depth_order + gsa_scan_tiles_sorted -- order, rects, offsets (exclusive
 * scan of the record counts in depth order), block_owner (n_blocks + 1 entries as gsa_scan_tiles_rect leaves them) and
 * ctl[0] = M -- in four launches whatever the key width, for callers that know the range of the visible depth keys
 * beforehand: ONE global radix pass buckets the keys by depth range (bucket = ((key - key_min) * scale) >> 32, clamped to
 * 254 -- 255 buckets of equal key width 2^32 / scale <= gsa_depth_order_msd_max_width() = 65536, i.e. scale >= 65536;
 * the culled sentinel alone in bucket 255), then one workgroup per bucket sorts its keys in LDS on their offset inside
 * the bucket, gathers rectangles and counts, scans them and adds the records of the buckets in front (published per
 * bucket through one 8-byte granule; bounded spin).
 * ctl[1..2] = min / max key over the visible gaussians as measured, ctl[3] = status (0 ok; bit 0: a bucket exceeded the
 * LDS capacity of 8192 keys; bit 1: a wait timed out).  The result is valid iff ctl[3] == 0, ctl[1] >= key_min and
 * ((ctl[2] - key_min) * scale) >> 32 <= 254 -- checked by the caller after the fact; otherwise it repeats with
 * gsa_depth_order(key_bits = 32).  Entries of order / rects / offsets that belong to culled gaussians (they sort behind
 * every visible one and have no records) are not written.  workspace: gsa_depth_order_msd_workspace_bytes(n).
 * ctl_host (optional): four words of PINNED, device-mapped HOST memory (hipHostMalloc).  The last workgroup stores
 * ctl[0..2] there and then, released at system scope, word 3 = status | token << 8 (token < 2^24, chosen by the caller,
 * different from what the word holds): the caller polls word 3 for its token instead of queueing a 16-byte copy and an
 * event behind the kernel (on the bench step those cost a 4-us copy kernel and a 6-us bubble on the stream).
 * key_range_dev (ABI 7; optional, may be NULL; 16-byte aligned): the 4 * ceil(n / 64) words gsa_project_fwd left for THIS
 * view.  key_min / scale are then ignored: the first launch builds the view's own DEPTH TABLE on the device -- the exact
 * key range cut into <= 4096 aligned cells, the sample keys counted per cell, consecutive cells grouped into <= 255
 * buckets of about equal key count and at most 2^24 keys of width -- so the buckets fit the view at hand whatever views
 * came before, and however unevenly depth is distributed (with 255 EQUAL buckets the diagonal cameras of the bench ring
 * put 3.5 x the mean into one bucket, beyond the local sorts' capacity).  The order produced does not depend on the
 * table (any monotone table gives the same order).
 * ABI 10: a bucket's local sort covers 24 bits of key offset (16 before: about two octaves of depth per VIEW) -- a bucket
 * wider than 65536 keys takes a third local pass -- so any range of positive float depths fits 255 buckets (scale > 256;
 * gsa_depth_order_msd_max_width() = 2^24).  ctl[3] bit 2 (the view's key range is wider than 250 x 2^24 keys) can only be
 * raised by keys that are no depths. */
size_t gsa_depth_order_msd_workspace_bytes(int n);
int gsa_depth_order_msd_max_width(void);
/* (diagnostics) where gsa_depth_order_msd(key_range_dev) leaves the view's depth table inside its workspace: byte offset from
 * a 16-byte-aligned workspace base, and its size in uint32 -- bytes [0, 4096): bucket of every cell; words [1024, 1281):
 * first cell of every bucket, then the end; words 1284..1287: first cell's index (key >> shift), shift, cells, flags. */
size_t gsa_depth_order_msd_table_offset(int n);
int gsa_depth_order_msd_table_words(void);
int gsa_depth_order_msd(int n, const uint32_t *depth_keys, const uint32_t *tile_rects, uint32_t key_min, uint32_t scale,
                        const uint32_t *key_range_dev, uint32_t *order, uint32_t *rects, uint32_t *offsets, int32_t *block_owner,
                        int n_blocks, uint32_t *ctl, uint32_t *ctl_host, uint32_t token, void *workspace,
                        size_t workspace_bytes, gsa_stream_t stream);
int gsa_scan_tiles_sorted(int n, uint32_t *counts_offsets, int32_t *block_owner, int n_blocks, uint32_t *total,
                          void *workspace, size_t workspace_bytes, gsa_stream_t stream);
/* ABI 13 -- gsa_scan_tiles_sorted behind gsa_depth_order(key_range = ctl + 1) that also REPORTS like gsa_depth_order_msd: ctl[0]
 * = M, and the single workgroup of its middle launch stores ctl[0..2] into ctl_host (four words of pinned, device-mapped host
 * memory) and releases word 3 = token << 8 (status 0) at system scope -- the caller polls for its token instead of queueing a
 * 16-byte copy and an event behind the scan (19 + 6 us on the stream at 5 M gaussians). */
int gsa_scan_tiles_sorted_report(int n, uint32_t *counts_offsets, int32_t *block_owner, int n_blocks, uint32_t *ctl,
                                 uint32_t *ctl_host, uint32_t token, void *workspace, size_t workspace_bytes,
                                 gsa_stream_t stream);
int gsa_tile_bin16(int n, const uint32_t *order, const uint32_t *offsets, const uint32_t *rects,
                   const int32_t *block_owner, int width, int height, int bin_tile, uint32_t m_capacity,
                   const uint32_t *m_dev, uint16_t *tile_tmp,
                   uint32_t *gid_tmp, uint16_t *sorted_tile, uint32_t *sorted_gid, int32_t *tile_offsets,
                   const uint32_t *qbox,
                   void *workspace, size_t workspace_bytes, gsa_stream_t stream);

/* ABI 12 -- gsa_tile_bin16 on the one-pass placement path (gsa_tile_bin16_direct, n > 0, m_capacity > 0) that ALSO writes
 * gsa_tile_schedule's output: eight extra workgroups of the placement launch order every XCD band's 16 x 16 tiles by the
 * length of the lists they walk (tile_order[8 * ceil(n_tiles16 / 8)], n_tiles16 = the frame's 16 x 16 tiles) while the
 * records are being placed -- one launch and one kernel boundary less per view than gsa_tile_bin16 + gsa_tile_schedule,
 * the same tile_order up to the order inside a bucket (which only moves work in time). */
int gsa_tile_bin16_scheduled(int n, const uint32_t *order, const uint32_t *offsets, const uint32_t *rects,
                             const int32_t *block_owner, int width, int height, int bin_tile, uint32_t m_capacity,
                             const uint32_t *m_dev, uint16_t *sorted_tile, uint32_t *sorted_gid, int32_t *tile_offsets,
                             const uint32_t *qbox, int32_t *tile_order, void *workspace, size_t workspace_bytes,
                             gsa_stream_t stream);

/* For k in depth order, g = order[k]: write that gaussian's tile ids and its index g at
 * [offsets[k], offsets[k] + tiles_touched[g]).  Must use the cull_mode of gsa_project_fwd. */
int gsa_tile_emit(int n, const uint32_t *order, const uint32_t *offsets, const float *splat,
                  const int32_t *radii, int width, int height, int cull_mode, uint32_t m_capacity,
                  uint32_t *isect_tile, uint32_t *isect_gid, gsa_stream_t stream);

/* Device-side record count.  The record count M is produced on the device (gsa_scan_tiles' `total`).  These
 * variants take a CAPACITY the buffers were sized for plus a pointer to the live count, so that the host can
 * queue the whole forward pass without waiting for M; every kernel clamps to min(capacity, *count).  If the
 * count turns out to exceed the capacity the results are truncated -- the caller checks (asynchronously) and
 * re-runs with larger buffers. */
int gsa_sort_pairs_u32_dev(int capacity, const uint32_t *n_dev, const uint32_t *keys_in,
                           const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                           uint32_t *keys_tmp, uint32_t *vals_tmp, int begin_bit, int end_bit,
                           void *workspace, size_t workspace_bytes, gsa_stream_t stream);
int gsa_tile_emit_dev(int n, const uint32_t *order, const uint32_t *offsets, const float *splat,
                      const int32_t *radii, int width, int height, int cull_mode, uint32_t m_capacity,
                      const uint32_t *m_dev, uint32_t *isect_tile, uint32_t *isect_gid, gsa_stream_t stream);
int gsa_tile_offsets_dev(uint32_t m_capacity, const uint32_t *m_dev, const uint32_t *sorted_tile,
                         int n_tiles, int32_t *tile_offsets, gsa_stream_t stream);

/* tile_offsets[t] = first record of tile t in the tile-sorted list, t in [0, n_tiles]; [n_tiles] = m. */
int gsa_tile_offsets(uint32_t m, const uint32_t *sorted_tile, int n_tiles, int32_t *tile_offsets,
                     gsa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 3 (row a3.4): front-to-back alpha compositing of RGB (channels == 3) or RGB+depth
 * (channels == 4) per 16x16 tile.  background: [channels] or NULL (= zeros).
 * Outputs: render[H,W,channels], alpha[H,W] = 1 - T, final_T[H,W] = T (saved for the backward: recomputing
 * 1 - alpha would lose the low bits of a small T), last_id[H,W] int32: position in the sorted list of the last
 * record the pixel's 8x8 quadrant walked while the pixel was still accumulating (it may itself have failed the
 * alpha test there), -1 if none -- where stage 4 starts its reverse sweep for that pixel.
 * quad_list / quad_count (optional, both or neither): per 8x8 quadrant the list positions of the records that were
 * BLENDED at one of its pixels at least, ascending, and their number: quad_count[4 * tile + q] entries at
 * quad_list[4 * tile_offsets[tile] + q * (tile_offsets[tile + 1] - tile_offsets[tile])] (quad_list holds 4 M
 * uint32, mostly unwritten).  gsa_composite_bwd_lists sweeps exactly those records.
 * v_splat_zero / n_zero_rows (optional; NULL / 0): a [n_zero_rows,16] float buffer (16-byte aligned) the kernel fills with
 * zeros on the side -- the gradient records of the backward sweep to come, which then needs no fill pass of its own
 * (gsa_composite_bwd_prepare with n = 0).
 * bin_tile (here and in gsa_tile_bin16 / gsa_tile_schedule / gsa_composite_bwd*): 0 / 16 -- tile_offsets[n_tiles + 1]
 * indexes the 16 x 16 tiles; 32 -- the records were binned into 32 x 32 bins (gsa_project_fwd's bin_tile):
 * tile_offsets[n_bins + 1] indexes the bins (row-major on ceil(width / 32) x ceil(height / 32)), the four tiles of a bin
 * walk its list, and quad_list holds 16 M uint32 (16 quadrant slices per bin, at 16 * offsets[bin] + (4 * sub-tile +
 * quadrant) * list length).
 * gid_boxes (ABI 10; 0 or 1; here and in gsa_composite_bwd / _bwd_lists): 1 -- the top byte of every sorted_gid entry is
 * the quadrant box gsa_tile_bin16 left there (its qbox argument) and the gaussian id is the low 24 bits: a quadrant's
 * wave gathers and stages only the records whose box covers it.  Results do not depend on it (a record whose bounding box
 * misses a quadrant fails the alpha test at every pixel of it).  bin_tile 16 or 32 only.  0: plain 32-bit ids, every
 * record of a list is a candidate for every quadrant.
 */
int gsa_composite_fwd(int width, int height, int channels, const float *splat,
                      const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                      const float *background,
                      const int32_t *tile_order, float *render, float *alpha, float *final_T, int32_t *last_id,
                      uint32_t *quad_list, int32_t *quad_count, float *v_splat_zero, long long n_zero_rows,
                      int bin_tile, gsa_stream_t stream);

/* Launch order for stages 3 and 4 (optional; speed only, never results): tile_order[8 * ceil(n_tiles / 8)] receives,
 * for each of the 8 tile bands, the band's tiles with the longest record lists first (-1 pads the last band). Pass it
 * as `tile_order` to gsa_composite_fwd / gsa_composite_bwd, or pass NULL there for index order. */
int gsa_tile_schedule(int n_tiles, const int32_t *tile_offsets, int32_t *tile_order, int tile_w, int bin_tile,
                      gsa_stream_t stream);

/* Stage 4 (row a3.5): reverse-order gradient sweep.  v_splat[N,16] must be zero on entry and receives, per
 * gaussian, the raw moments of u = d loss / d sigma over d = mean2d - pixel:
 *   [0] sum u dx  [1] sum u dy  [2] sum u dx^2  [3] sum u dx dy  [4] sum u dy^2  [5] sum u
 *   [6..8] sum fac * v_render(rgb)  [9] sum fac * v_render(depth)   (fac = alpha * T in front)
 * d loss / d (mean2d, conic, opacity) are linear in them; gsa_project_bwd finishes that per gaussian. */
int gsa_composite_bwd(int width, int height, int channels, const float *splat,
                      const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets, const float *background,
                      const int32_t *tile_order, const float *final_T, const int32_t *last_id,
                      const float *v_render, const float *v_alpha, float *v_splat, int bin_tile, gsa_stream_t stream);

/* First launch of a backward pass over the forward's lists: zero-fills v_splat[N,16] (the sweep accumulates into it) and
 * writes the launch order for gsa_composite_bwd_lists -- per tile band, the tiles whose quadrants blended the most
 * records first (tile_order[8 * ceil(n_tiles / 8)], as gsa_tile_schedule).  n = 0: the records are clear already
 * (gsa_composite_fwd's v_splat_zero); only the order is written. */
int gsa_composite_bwd_prepare(long long n, float *v_splat, int n_tiles, const int32_t *quad_count,
                              int32_t *tile_order, gsa_stream_t stream);

/* gsa_composite_bwd over the per-quadrant lists of blended records the forward pass left (quad_list / quad_count of
 * gsa_composite_fwd): same result, but no reach test and no record that blends nowhere in the quadrant (in a dense
 * scene ~10 % of what the plain sweep walks), in full batches of 64. */
int gsa_composite_bwd_lists(int width, int height, int channels, const float *splat,
                            const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets, const float *background,
                            const int32_t *tile_order, const float *final_T, const int32_t *last_id,
                            const float *v_render, const float *v_alpha, const uint32_t *quad_list,
                            const int32_t *quad_count, float *v_splat, int bin_tile, gsa_stream_t stream);

/* ABI 13 -- gsa_composite_bwd_lists with the COLOUR moments going to v_rgb [N,3] (zeroed by the caller) instead of fields 6..8
 * of the gradient record, already masked by the colour clamp (a channel the forward clamped at 0 -- its record colour is
 * exactly 0 -- gets no gradient): d loss / d rgb per gaussian in the 12-byte layout the view-parallel exchange all-gathers,
 * straight out of the sweep, so a rank's collective for view j is queued behind view j's SWEEP and the projection backward
 * of all its views runs once per step (gsa_project_bwd_batch_geometry).  Fields 0..5 and 9 go to v_splat as before. */
int gsa_composite_bwd_lists_rgb(int width, int height, int channels, const float *splat,
                                const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets, const float *background,
                                const int32_t *tile_order, const float *final_T, const int32_t *last_id,
                                const float *v_render, const float *v_alpha, const uint32_t *quad_list,
                                const int32_t *quad_count, float *v_splat, float *v_rgb, int bin_tile, gsa_stream_t stream);

/* ABI 14 -- HOT gaussians.  Every (quadrant, record) visit of the reverse sweep is one 64-byte float-atomic request on the
 * gaussian's gradient record; a camera inside the scene sees splats that cover thousands of tiles, and tens of thousands of
 * requests on one line serialise (measured: the sweep 0.535 ms with them, 0.158 ms without).  gsa_composite_bwd_lists_hot
 * is gsa_composite_bwd_lists for a v_splat buffer of n + gsa_hot_rows() rows, ALL zero on entry (gsa_composite_fwd's
 * n_zero_rows / gsa_composite_bwd_prepare's n take the sum): behind the n gradient records lies the HOT REGION,
 * GSA_HOT_REPLICAS x GSA_HOT_ENTRIES replica rows (replica k of entry e is row n + k GSA_HOT_ENTRIES + e: the replicas of
 * one gaussian lie 256 KB apart, on different memory channels) and GSA_HOT_ENTRIES tag words.  A gaussian whose tile rectangle (word 11 of
 * its splat record) is at least GSA_HOT_MIN_WIDTH tiles wide claims an entry on first touch (hash of its index, compare-and-
 * swap on the tag, GSA_HOT_PROBES probes; none free: it adds to its own record as before) and every quadrant's wave adds
 * into replica (4 tile + quadrant) mod GSA_HOT_REPLICAS; the claiming wave leaves entry + 1 (an integer) in word
 * GSA_HOT_MARK_FIELD of the gaussian's own gradient record.  gsa_project_bwd / _acc / _factored / _batch* read that word with
 * the record and add the replicas of a marked gaussian, taking the hot region to lie behind row n of the v_splat they are
 * given -- so WORDS 10 .. 15 OF A GRADIENT RECORD ARE RESERVED: zero, or anything that is not the bit pattern of an integer in
 * 1 .. GSA_HOT_ENTRIES, unless the record comes out of gsa_composite_bwd_lists_hot.  Sums are the same up to the order of the
 * float additions (the atomics' order is not fixed either way). */
#ifndef GSA_HOT_LOG_ENTRIES   /* (developer builds override these three: tools/wide_ab.sh) */
#define GSA_HOT_LOG_ENTRIES 12
#endif
#define GSA_HOT_ENTRIES (1 << GSA_HOT_LOG_ENTRIES)
#ifndef GSA_HOT_REPLICAS
#define GSA_HOT_REPLICAS 16
#endif
#ifndef GSA_HOT_MIN_WIDTH
#define GSA_HOT_MIN_WIDTH 24
#endif
#define GSA_HOT_PROBES 8
#define GSA_HOT_MARK_FIELD 11
#define GSA_HOT_ROWS ((long long)GSA_HOT_ENTRIES * GSA_HOT_REPLICAS + GSA_HOT_ENTRIES / GSA_SPLAT_STRIDE)
long long gsa_hot_rows(void);   /* = GSA_HOT_ROWS, for hosts that do not compile against this header */
int gsa_composite_bwd_lists_hot(int width, int height, int channels, const float *splat,
                                const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets, const float *background,
                                const int32_t *tile_order, const float *final_T, const int32_t *last_id,
                                const float *v_render, const float *v_alpha, const uint32_t *quad_list,
                                const int32_t *quad_count, long long n, float *v_splat, int bin_tile, gsa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 5 (row a3.6): projection + SH backward.  Reads the sweep's moment records (stage 4), the raw
 * parameters and the saved radii; writes EVERY element of v_means[N,3], v_quats[N,4], v_scales[N,3],
 * v_opacities[N] and v_colors (same shape as colors; zeros above the active degree and for
 * culled gaussians).  `splat` is not read (the projection is recomputed, the colour clamp mask re-derived);
 * it stays in the signature and may be NULL.  sh_aux (may be NULL): the [N,12] digest gsa_project_fwd left for THIS
 * view and these parameters; with it `colors` is not read for sh_degree >= 1 (same gradients to rounding: the
 * direction term is summed per channel first instead of per basis function first).
 */
int gsa_project_bwd(int n, const float *means, const float *quats, const float *scales,
                    const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                    const float *viewmat, const float *intrin, int width, int height,
                    const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                    float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                    const float *sh_aux, gsa_stream_t stream);

/* The same with `accumulate` != 0: ADD to what the five output buffers hold (which must then be fully initialised)
 * instead of overwriting them -- the views of a batch sum into one gradient bucket without a separate pass. */
int gsa_project_bwd_acc(int n, const float *means, const float *quats, const float *scales,
                        const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                        const float *viewmat, const float *intrin, int width, int height,
                        const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                        float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                        int accumulate, const float *sh_aux, gsa_stream_t stream);

/* Factored colour gradient for the view-parallel exchange.  The SH-coefficient gradient of one view is the outer
 * product  v_colors[g][k][c] = basis_k(dir(g, view)) * v_rgb[g][c]  (48 of the 59 gradient floats per gaussian), and every
 * rank can evaluate the basis of every view.  gsa_project_bwd_factored writes v_rgb[N,3] (d loss / d rgb after the
 * clamp mask; zeros for culled gaussians) instead of v_colors; gsa_sh_grad_expand rebuilds
 *   v_colors (+)= sum over n_views of basis(viewmats[v]) (x) v_rgb[v]        (v_rgb: [n_views, N, 3])
 * so ranks exchange 11 N floats by all-reduce plus 3 N floats per rank by all-gather instead of all-reducing 59 N. */
int gsa_project_bwd_factored(int n, const float *means, const float *quats, const float *scales,
                             const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                             const float *viewmat, const float *intrin, int width, int height,
                             const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                             float *v_quats, float *v_scales, float *v_opacities, float *v_rgb,
                             int accumulate, const float *sh_aux, gsa_stream_t stream);
int gsa_sh_grad_expand(int n, const float *means, int sh_degree, int sh_coeffs, int n_views,
                       const float *viewmats, const float *intrin, const float *v_rgb, float *v_colors,
                       int accumulate, gsa_stream_t stream);

/* ABI 12 -- the backward projection for the n_cams cameras of one optimisation step in ONE launch, the counterpart of
 * gsa_project_fwd_batch: per camera the chain rule of gsa_project_bwd on that camera's gradient record and coefficient
 * digest, the geometry gradients SUMMED over the cameras in camera order (the sums a chain of gsa_project_bwd_acc calls
 * leaves) and the SH-coefficient gradient  sum_c basis(dir_c) (x) d loss / d rgb_c  -- what gsa_project_bwd_factored per
 * camera + gsa_sh_grad_expand rebuild -- every output element written once (accumulate != 0: added to what the buffers hold).
 * radii / v_splat / sh_aux: HOST arrays of n_cams DEVICE pointers (camera c's radii [N,2], gradient records [N,16] from
 * gsa_composite_bwd*, digests [N,12] from gsa_project_fwd*); viewmats / intrins / intrin_stride as gsa_project_fwd_batch.
 * A gaussian's geometry is read once and its gradients are written once: 760 B per gaussian at four cameras where four
 * launches and the expansion move 1308.  GSA_ERR_UNSUPPORTED unless 1 <= sh_degree <= 3, sh_coeffs == 16 and n_cams <= 8
 * (the caller then runs the per-camera entry points). */
int gsa_project_bwd_batch(int n, const float *means, const float *quats, const float *scales,
                          const float *opacities, int sh_degree, int sh_coeffs, int n_cams, const float *viewmats,
                          const float *intrins, int intrin_stride, int width, int height,
                          const int32_t *const *radii, const float *const *v_splat, const float *const *sh_aux,
                          float *v_means, float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                          int accumulate, gsa_stream_t stream);

/* ABI 13 -- gsa_project_bwd_batch for the step of a view-parallel RANK: the sweeps of the rank's n_cams views left their
 * masked colour gradients in v_rgb[c] ([N,3] each: gsa_composite_bwd_lists_rgb; all-gathered right behind the sweeps, the
 * coefficient gradient of ALL views is rebuilt from the gathered slabs by gsa_sh_grad_expand), so this launch only sums the
 * GEOMETRY gradients of the rank's cameras into v_means / v_quats / v_scales / v_opacities (written once; accumulate != 0:
 * added) -- one launch per step instead of one gsa_project_bwd_factored per view: the geometry is read once and the sums are
 * not read back.  v_rgb: a HOST array of n_cams DEVICE pointers, like radii / v_splat / sh_aux. */
int gsa_project_bwd_batch_geometry(int n, const float *means, const float *quats, const float *scales,
                                   const float *opacities, int sh_degree, int n_cams, const float *viewmats,
                                   const float *intrins, int intrin_stride, int width, int height,
                                   const int32_t *const *radii, const float *const *v_splat, const float *const *sh_aux,
                                   const float *const *v_rgb, float *v_means, float *v_quats, float *v_scales,
                                   float *v_opacities, int accumulate, gsa_stream_t stream);

/* gsa_sh_grad_expand with the output split the way the reference model stores the coefficients: v_dc [N,1,3] and
 * v_rest [N, sh_coeffs - 1, 3] (both overwritten). */
int gsa_sh_grad_expand_split(int n, const float *means, int sh_degree, int sh_coeffs, int n_views,
                             const float *viewmats, const float *intrin, const float *v_rgb, float *v_dc,
                             float *v_rest, gsa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ABI 13 -- ONE call per view forward and ONE per view backward.  The steady-state chain of a view behind its projection is
 * eight to ten launches: gsa_depth_order_msd -> gsa_tile_bin16_scheduled -> gsa_composite_fwd, then
 * gsa_composite_bwd_prepare -> gsa_composite_bwd_lists (-> gsa_project_bwd_acc / _factored).  A host in the reference's
 * language (Python, src/backends/gsplat_backend.py:168-182 makes ONE call per view) pays per C-ABI call and per argument;
 * gsa_view_fwd / gsa_view_bwd take the whole chain's arguments in one plain struct and queue the launches back to back.
 * Every field means what the argument of the same name means in the stage it is passed to; all scratch stays
 * caller-provided; the results are the stage-by-stage path's, bit for bit.  What the fused calls do NOT cover stays with
 * the stages: the first call of a frame size / a repair (exact sizes, LSD ordering), grids without the one-pass placement
 * (gsa_tile_bin16_direct == 0: GSA_ERR_UNSUPPORTED), cull_mode 1.
 *   struct_bytes   sizeof(gsa_view_args) as the caller sees it (checked: a layout mismatch is an error, not a crash)
 *   wait_event     (optional) a hipEvent_t the stream is made to wait for between the binning and the forward sweep: the
 *                  RECORDS part of a split projection (gsa_project_fwd_parts) running on another stream
 *   v_splat        the gradient records [n,16]: cleared by the forward sweep on the side, accumulated into by the backward
 *   bwd_tile_order the backward sweep's own launch order [8 * ceil(n_tiles16 / 8)] (written by gsa_view_bwd)
 *   project_mode   gsa_view_bwd: 0 = stop behind the sweep (the caller finishes all views with gsa_project_bwd_batch);
 *                  1 = gsa_project_bwd_acc into v_means .. v_colors; 2 = gsa_project_bwd_factored (v_rgb instead of v_colors)
 *   wait_bucket    (optional) a hipEvent_t waited for in front of the projection backward (the bucket is summed in view order)
 *   sweep_rgb      (optional, [n,3], zeroed by the caller) gsa_view_bwd's sweep is gsa_composite_bwd_lists_rgb: the masked colour
 *                  gradient leaves the sweep in the exchange layout (with project_mode 0: the rank's step ends in ONE
 *                  gsa_project_bwd_batch_geometry)
 */
typedef struct gsa_view_args {
    uint32_t struct_bytes;
    int32_t n, width, height, channels, bin_tile, gid_boxes, n_blocks;
    uint32_t m_capacity, key_min, scale, token;
    /* forward inputs */
    const float *splat;
    const uint32_t *depth_keys, *tile_rects, *key_range, *qbox;
    const float *background;
    /* binning: outputs and scratch */
    uint32_t *order, *rects, *offsets;
    int32_t *block_owner;
    uint32_t *ctl, *ctl_host;
    void *depth_ws;
    size_t depth_ws_bytes;
    uint32_t *sorted_gid;
    int32_t *tile_offsets, *tile_order;
    void *bin_ws;
    size_t bin_ws_bytes;
    void *wait_event;
    /* forward sweep outputs */
    float *render, *alpha, *final_T;
    int32_t *last_id;
    uint32_t *quad_list;
    int32_t *quad_count;
    float *v_splat;
    /* backward */
    const float *v_render, *v_alpha;
    int32_t *bwd_tile_order;
    int32_t project_mode, accumulate, sh_degree, sh_coeffs;
    void *wait_bucket;
    const float *means, *quats, *scales, *opacities, *colors, *viewmat, *intrin, *sh_aux;
    const int32_t *radii;
    float *v_means, *v_quats, *v_scales, *v_opacities, *v_colors, *v_rgb;
    float *sweep_rgb;
    int hot;   /* ABI 14: 1 -- v_splat has n + gsa_hot_rows() rows; the forward clears them all, the sweep is gsa_composite_bwd_lists_hot */
} gsa_view_args;
size_t gsa_view_args_size(void);
int gsa_view_fwd(const gsa_view_args *args, gsa_stream_t stream);
int gsa_view_bwd(const gsa_view_args *args, gsa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step stages around the rasterizer.
 *
 * gsa_view_loss: the reference's per-view loss, src/training/trainer.py:485-492 --
 *   loss = (1 - lambda) * L1(img, target) + lambda * (1 - SSIM_3x3(img, target)),  img = clamp(render, 0, 1) with a
 *   straight-through gradient (:102-107), SSIM with a 3x3 zero-padded box window (:354-371) -- and its gradient with
 *   respect to the render.  render/target: [H,W,3] fp32.  sums[3] (double, ACCUMULATED: zero them first) receive
 *   sum |img - target|, sum SSIM map, sum (img - target)^2 over the 3 H W elements (L1 = sums[0] / 3HW, SSIM =
 *   sums[1] / 3HW, PSNR's mse = sums[2] / 3HW, :70-75).  v_render (may be NULL) = grad_scale * d loss / d render.
 *   workspace: gsa_view_loss_workspace_bytes(width, height).  Images of 4 GiB and more (12 H W bytes; rows are addressed
 *   with 32-bit byte offsets) are refused with GSA_ERR_INVALID_ARGUMENT.
 *
 * gsa_adam_step: one torch.optim.Adam step (no weight decay, no amsgrad) over a flat bucket of n floats split into
 *   n_segments (<= 8) consecutive segments, segment k = [segment_end[k-1], segment_end[k]) with learning rate
 *   segment_lr[k] (host arrays; segment_end[n_segments-1] == n) -- the six parameter groups of
 *   trainer.py:331-340 laid out in one 59 N buffer.  step = 1 for the first update.  skip_if_nonzero: see
 *   gsa_grad_unscale. */
/* gsa_densify_gather: the data movement of GaussianModel.densify_and_prune
 * (src/gaussians/gaussian_model.py:110-366: clone, split into two, prune, with the Adam moments spliced through,
 * :9-50) as ONE pass.  Buckets are field-major (field f of width field_width[f] occupies a contiguous block of
 * width * n floats, fields in order).  Output row j copies old row src_row[j]; kind[j] = 0 survivor (moments kept),
 * 1 clone (moments zero), 2 split child (moments zero; log-scale field lowered by log_split = log 1.6, xyz moved by
 * noise[j] * exp(new log-scale), :296-303).  noise: [n_new,3].  The three moment pointers may all be NULL. */
int gsa_densify_gather(long long n_old, long long n_new, int n_fields, const int *field_width, int xyz_field,
                       int scaling_field, float log_split, const int32_t *src_row, const uint8_t *kind,
                       const float *noise, const float *old_param, const float *old_exp_avg,
                       const float *old_exp_avg_sq, float *new_param, float *new_exp_avg, float *new_exp_avg_sq,
                       gsa_stream_t stream);

size_t gsa_view_loss_workspace_bytes(int width, int height);
int gsa_view_loss(int width, int height, const float *render, const float *target, float lambda_dssim,
                  float grad_scale, void *workspace, size_t workspace_bytes, float *v_render, double *sums,
                  gsa_stream_t stream);
int gsa_adam_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                  int n_segments, const long long *segment_end, const float *segment_lr, double beta1,
                  double beta2, double eps, int step, const uint32_t *skip_if_nonzero, gsa_stream_t stream);
/* The same step with STRIPED segments: segment k's elements repeat with segment_period[k] (0: a plain segment; else its
 * length must be a multiple), and the first segment_head[k] elements of every period take segment_head_lr[k] instead of
 * segment_lr[k].  What a model that keeps the SH table as ONE [N,16,3] segment needs: the 3 floats of coefficient 0 are
 * the reference's "f_dc" optimizer group, the other 45 its "f_rest" group (src/training/trainer.py:333-334).  The three
 * arrays may all be NULL (= gsa_adam_step). */
int gsa_adam_step_striped(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                          int n_segments, const long long *segment_end, const float *segment_lr,
                          const int *segment_period, const int *segment_head, const float *segment_head_lr,
                          double beta1, double beta2, double eps, int step, const uint32_t *skip_if_nonzero,
                          gsa_stream_t stream);
/* Mixed-precision bookkeeping (the reference trains with GradScaler, src/training/trainer.py:31,505-511: the upstream
 * gradient of the rasterizer arrives multiplied by the loss scale): grad *= inv_scale in place, and *found_inf
 * (uint32 on the device, ACCUMULATED: zero it first) becomes 1 if any element is not finite -- torch's
 * GradScaler.unscale_.  gsa_adam_step's skip_if_nonzero (may be NULL) then skips the whole update, as
 * GradScaler.step does. */
int gsa_grad_unscale(long long n, float *grad, float inv_scale, uint32_t *found_inf, gsa_stream_t stream);

/* The metrics of a step in one launch (the bookkeeping of src/training/trainer.py:485-519 and :70-75): view_sums
 * [n_local_views,3] (device, the sums gsa_view_loss leaves per view), view_inv_count[n_local_views] (HOST array: 1 / (3 H W)
 * of every view; at most 16 views), reg_partials[n_partials] (device: gsa_activate_bwd's partial sums; the regulariser is
 * reg_factor times their sum).  report[5] (device, double, overwritten) = { sum l1, sum ssim, sum psnr,
 * sum((1 - lambda) l1 + lambda (1 - ssim)) / n_views_total + reg, reg } with psnr = mse < 1e-10 ? 100 : -10 log10(mse);
 * the two reg terms only with add_reg (rank 0: the ranks sum their reports). */
int gsa_step_report(int n_local_views, const double *view_sums, const double *view_inv_count, double lambda_dssim,
                    int n_views_total, const double *reg_partials, long long n_partials, double reg_factor, int add_reg,
                    double *report, gsa_stream_t stream);

/* Activations of the reference model's raw parameters (src/gaussians/gaussian_model.py:86-108 as the trainer calls them,
 * trainer.py:399-406) in one pass: scales = exp(scaling) [N,3], quats = rotation / max(|rotation|, 1e-12) [N,4],
 * opacities = sigmoid(opacity) [N], shs = cat(features_dc [N,1,3], features_rest [N,15,3]) [N,16,3].
 * quats and shs must be 16-byte aligned.  shs may be NULL (features_dc / features_rest then too): geometry only, for a
 * model whose coefficients already are [N,16,3] rows. */
int gsa_activate_fwd(long long n, const float *scaling, const float *rotation, const float *opacity,
                     const float *features_dc, const float *features_rest, float *scales, float *quats,
                     float *opacities, float *shs, gsa_stream_t stream);

/* The derivatives of those activations applied to the gradients of the activated parameters (v_scales [N,3],
 * v_quats [N,4], v_opacities [N]) -> v_scaling [N,3], v_rotation [N,4], v_opacity [N] (overwritten), together with
 * the scale-ratio regulariser of trainer.py:434-445,
 *   reg = reg_weight * mean(max(ratio, reg_max_ratio) - reg_max_ratio),  ratio = max(scales) / max(min(scales), 1e-8):
 * its gradient is added to v_scaling when add_reg_grad != 0, and the per-block sums of the hinge term are written to
 * reg_partials[gsa_activate_partials(n)] (doubles; reg = reg_weight * sum(reg_partials) / n). */
size_t gsa_activate_partials(long long n);
int gsa_activate_bwd(long long n, const float *scaling, const float *rotation, const float *opacity,
                     const float *v_scales, const float *v_quats, const float *v_opacities, float reg_max_ratio,
                     float reg_weight, int add_reg_grad, float *v_scaling, float *v_rotation, float *v_opacity,
                     double *reg_partials, gsa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_AMD_H */
