/*
 * svamd.h - C ABI of libsvamd.so, the MI355X (gfx950) implementation of the
 * normalised cross-correlation hot path of jessefr0st999/stereo-vision.
 *
 * The reference is pure Python and has no FFI of its own; each entry point
 * below replaces the Python function cited beside it (paths relative to the
 * reference checkout) and is what a ctypes binding in that function's module
 * would call.
 *
 * Conventions
 *  - Every array argument is a DEVICE pointer (hipMalloc'd or a torch tensor's
 *    data_ptr()) unless its comment says "host".  Nothing is allocated or freed
 *    by the library: callers pass outputs and, where a function needs scratch
 *    memory, a workspace of at least <fn>_workspace_bytes(...) bytes, 256-byte
 *    aligned.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *    only enqueue work; they do not synchronise.
 *  - Images and templates are row-major; sizes are in elements.
 *  - Return value: SVAMD_OK or an SVAMD_ERR_* code; svamd_last_error() returns
 *    a message for the calling thread's most recent failure.
 */
#ifndef SVAMD_H
#define SVAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVAMD_OK             0
#define SVAMD_ERR_INVALID    1   /* bad argument (NULL, non-positive size, unsupported dtype) */
#define SVAMD_ERR_SHAPE      2   /* shape rule of the reference violated (it raises Exception) */
#define SVAMD_ERR_WORKSPACE  3   /* workspace missing, too small or misaligned */
#define SVAMD_ERR_HIP        4   /* HIP runtime error */
#define SVAMD_ERR_LIMIT      5   /* size beyond what the kernel supports */

/* element types of image arguments */
#define SVAMD_U8   0
#define SVAMD_F32  1
#define SVAMD_F64  2

/* precision of the multiply-accumulate in svamd_scan_pairs */
#define SVAMD_ACC_F32 0
#define SVAMD_ACC_F64 1
/* svamd_scan_grid only: float32, then float64 for every window whose selection came
 * within the guard band of flipping, and for the value each window reports */
#define SVAMD_ACC_MIXED 2

/* selection rule of svamd_band_scan_u8 */
#define SVAMD_BAND_ARGMAX 0     /* first maximum in ascending offset (utils.py:17-26) */
#define SVAMD_BAND_FOLD   1     /* image_scan's rule (sv_image_comparison.py:249-255) */

#define SVAMD_BAND_NO_MATCH INT32_MIN

/* flags of svamd_band_scan_u8 */
#define SVAMD_BAND_PREPARED 1   /* the workspace already holds svamd_band_prepare_u8's tables */
#define SVAMD_BAND_BOX_SUMS 2   /* svamd_band_scan_u8: cross terms from per-offset summed-area tables of the
                                   product image instead of one multiply-add per window pixel; identical
                                   results (svamd_band_box_sums_supported says whether a shape fits) */

int svamd_version(void);
const char *svamd_last_error(void);

/* ---- K1: summed-area tables -------------------------------------------------
 * No counterpart function in the reference: replaces the per-output
 * region_snapshot.mean() / .std() of correlation.py:106,112.
 * s1, s2: (h+1) x (w+1) tables of sum(v - offset) and sum((v - offset)^2) with a
 * zero first row and column.  U8 input -> int64 tables (exact); F32/F64 input
 * -> double tables. */
int svamd_sat2d(const void *img, int dtype, int h, int w, int64_t pitch,
                double offset, void *s1, void *s2, void *stream);

/* ---- a11: utils.max_pos(array, step_x, step_y) ---------------------------------
 * utils.py:17-26: index of the first maximum ignoring NaNs (numpy.nanargmax);
 * *out_index = -1 when every element is NaN (numpy raises ValueError there).
 * The (x, y) scaling by the steps is left to the caller. */
int svamd_nanargmax(const double *data, int64_t n, int64_t *out_index, void *stream);

/* ---- a1: cross_correlate_2d(template, region, step_x, step_y) ----------------
 * correlation.py:77-117.  tmpl: th x tw float64, NaN = masked pixel.  region:
 * rh x rw of `region_dtype`.  out: ceil(rh/step_y) x ceil(rw/step_x) float64.
 * SVAMD_ERR_SHAPE when the template exceeds the region (correlation.py:81-82). */
size_t svamd_cross_correlate_2d_workspace_bytes(int th, int tw, int rh, int rw);
int svamd_cross_correlate_2d(const double *tmpl, int th, int tw,
                             const void *region, int region_dtype, int rh, int rw,
                             int step_x, int step_y, double *out,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ---- a2: cross_correlate_2d_spectral(template, region) ------------------------
 * correlation_spectral.py:32-55.  out: rh x rw float64 (the circular,
 * globally normalised surface).  min_std / min_mean_diff are the module
 * constants MIN_ST_DEV / MIN_MEAN_DIFF (correlation_spectral.py:6,10).
 * Any shapes.  Evaluated directly (a sliding dot product over the wrapped region)
 * or, when that is the slower way by the two routes' measured rates (from a few 1e9
 * products) and the template (cropped to the region) covers at least a quarter of the
 * region, through FFTs of both
 * operands zero-padded to powers of two (up to 4096 per side), which keeps the
 * circular semantics of the original shape exactly. */
size_t svamd_cross_correlate_2d_spectral_workspace_bytes(int th, int tw, int rh, int rw);
int svamd_cross_correlate_2d_spectral(const void *tmpl, int tmpl_dtype, int th, int tw,
                                      const void *region, int region_dtype, int rh, int rw,
                                      double min_std, double min_mean_diff, double *out,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- a2, batched, power-of-two shapes: the FFT route of correlation_spectral.py:51-54 --
 * `batch` independent pairs: templates [batch][th][tw] and regions [batch][H][W],
 * float32, H and W powers of two in 4..4096 (svamd_fft_supported).  Same
 * semantics as svamd_cross_correlate_2d_spectral (statistics over the full
 * arrays, template zero-padded or cropped top-left to H x W, low-deviation
 * branch); evaluated by a radix-4 FFT in LDS: packed row transforms, column
 * transforms fused with the conjugate multiply and the inverse column transform,
 * inverse row transforms (1024 x 1024 has kernels of its own: one wavefront
 * per transform).  out: [batch][H][W] of out_dtype (SVAMD_F32 or SVAMD_F64). */
int svamd_fft_supported(int H, int W);
size_t svamd_xcorr2d_fft_workspace_bytes(int batch, int H, int W);
int svamd_xcorr2d_fft(const float *tmpl, int th, int tw, const float *region, int H, int W,
                      int batch, double min_std, double min_mean_diff, void *out, int out_dtype,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- a3: cross_correlate_1d(vec_1, vec_2, normalised) -------------------------
 * correlation.py:44-75 (and its list twin cross_correlate_1d_raw, :4-42).
 * v1, v2: n float64.  out: 2n-1 float64. */
size_t svamd_cross_correlate_1d_workspace_bytes(int64_t n);
int svamd_cross_correlate_1d(const double *v1, const double *v2, int64_t n, int normalised,
                             double *out, void *workspace, size_t workspace_bytes, void *stream);

/* The same correlation for callers that cannot allocate or pass a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  The workspace is an array inside
 * the library's code object (about 31 MB per device, held until the process
 * ends); the call allocates and frees nothing.  n is limited to
 * SVAMD_XCORR1D_STATIC_MAX_N samples (SVAMD_ERR_LIMIT beyond).  All calls on one
 * device share that array: calls that overlap in time on one device (different
 * streams or threads) race; calls on one stream, or on different devices, do
 * not.  A non-NULL stream must belong to the current device (SVAMD_ERR_INVALID
 * otherwise).  The first call on each device looks up the array's address and must
 * be made outside stream capture; later calls on that device are plain kernel
 * launches and memsets and can be captured.  Results equal svamd_cross_correlate_1d's
 * bit for bit. */
#define SVAMD_XCORR1D_STATIC_MAX_N 65536
int svamd_cross_correlate_1d_static(const double *v1, const double *v2, int64_t n, int normalised,
                                    double *out, void *stream);

/* ---- a5: cross_correlate_1d_spectral(template, signal) ------------------------
 * correlation_spectral.py:12-30.  out: ns + nt - 1 float64.  The shorter array
 * slides over the other (the circular correlation reads the same either way).
 * Above 2e9 products both 1D functions evaluate the sum through the library's 2D
 * float32 transforms (the arrays cut into rows: long recordings against snippets,
 * music_patterns_africa.py:24-45) as long as the two arrays together stay below
 * 8.3 million samples; the workspace size reflects the route. */
size_t svamd_cross_correlate_1d_spectral_workspace_bytes(int64_t nt, int64_t ns);
int svamd_cross_correlate_1d_spectral(const double *tmpl, int64_t nt, const double *signal, int64_t ns,
                                      double *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a6 inner loop: x_corr_2d + np.max over many (template, region) pairs ------
 * sv_image_comparison.py:243-244 for every pair of a scan at once.
 * tbase / rbase: float64 arrays (the left / right image, or one packed buffer).
 * desc: npairs x 8 int64 = {t_off, t_pitch, t_h, t_w, r_off, r_pitch, r_h, r_w},
 * offsets and pitches in elements.  max_side: an upper bound of every t_h, t_w,
 * r_h, r_w of the call, at most SVAMD_PAIR_MAX_SIDE (it sizes the LDS tile; a
 * pair that exceeds it yields NaN).
 * out_max[p]: numpy.max of the pair's surface (NaN-propagating); pairs with an
 * empty region give NaN.  out_argmax (nullable): numpy.nanargmax of the surface
 * (-1 if all NaN).  surfaces (nullable): the r_h x r_w surfaces back to back in
 * pair order, starting at surf_off[p] (int64 element offsets). */
#define SVAMD_PAIR_MAX_SIDE 64
int svamd_scan_pairs(const double *tbase, const double *rbase,
                     const int64_t *desc, int64_t npairs, int max_side,
                     double min_std, double min_mean_diff, int acc_precision,
                     double *out_max, int64_t *out_argmax,
                     double *surfaces, const int64_t *surf_off, void *stream);

/* ---- a6 selection rule ---------------------------------------------------------
 * sv_image_comparison.py:236-255 for every window at once.  win_off: nwin+1
 * int64 offsets into the pair arrays (regions of one window are contiguous and
 * in the reference's visiting order); is_centre[p] != 0 where the region's
 * centre equals the window's; skip[p] != 0 for empty regions (:241-242).
 * best_pair[w]: winning pair index or -1; best_val[w]: its correlation or 0.
 * min_margin (nullable): per window, the smallest distance of any comparison
 * the rule made from flipping (used to re-check float32 decisions in float64). */
int svamd_scan_fold(const double *corr_max, const uint8_t *is_centre, const uint8_t *skip,
                    const int64_t *win_off, int64_t nwin, double threshold, double centre_tol,
                    int64_t *best_pair, double *best_val, double *min_margin, void *stream);

/* ---- a7 stage 0 + a8: whole_image_search_regions / region_pairs + image_scan per window
 * sv_image_comparison.py:35-47,74-79 with :143-191, :119-141, :283-298 and :229-255,
 * for window rows [row_begin, row_end) of the first stage, without a host-side list
 * of pairs.  The windows tile the left image on an nrow x ncol grid; a window's
 * candidate regions are slot 0 = its own position, then slot 1 + iy * nx + ix = the
 * scheme grid row by row (region_pairs' order), each the cross product of a per-column
 * and a per-row candidate.  The caller supplies per-axis tables (int32, device):
 *   xcand [ncol][1 + nx][SVAMD_GRID_CAND_FIELDS] = {begin, end, flags, rank, centre_rank}
 *     begin/end: the candidate's clipped box along the axis (get_window_boundaries,
 *     :294-297); flags: 1 = rejected by :288-292, 2 = equals the window's own centre
 *     (region_pairs drops that grid candidate, :136-140); rank / centre_rank: how many
 *     grid candidates before this one in its column are not rejected / are not
 *     rejected and at the centre.  Entry 0 is the window's own position.
 *   xwin [ncol][SVAMD_GRID_XWIN_FIELDS] = {t_begin, t_end, own_ok, valid, centred,
 *     own_before, valid_before, centred_before}: the template's extent along the axis
 *     (:170), whether the own position survives, the number of surviving grid
 *     candidates, how many of them sit at the centre, and the sums of those three over
 *     the columns to the left.
 *   ycand [nrow][1 + ny][...] likewise; ywin [nrow][SVAMD_GRID_YWIN_FIELDS] = the first
 *     five fields.
 *   row_first [nrow + 1] int64: index of each window row's first pair in the whole
 *     stage (a window (r, c) has own_ok + valid_y * valid_x - centred_y * centred_x pairs).
 * npairs = row_first[row_end] - row_first[row_begin] sizes the workspace (at most 2^31 - 1
 * per call: scan fewer rows at a time).  left / right: float64 images with row pitches in
 * elements.  precision: SVAMD_ACC_F32, SVAMD_ACC_F64 or SVAMD_ACC_MIXED.
 * best_slot / best_val: nrow * ncol entries indexed by window id (row-major); the call
 * writes rows [row_begin, row_end): the selected candidate slot or -1, and its
 * correlation or 0 (image_scan's return value, :281). */
#define SVAMD_GRID_CAND_FIELDS 5
#define SVAMD_GRID_XWIN_FIELDS 8
#define SVAMD_GRID_YWIN_FIELDS 5
size_t svamd_scan_grid_workspace_bytes(int64_t nwin, int64_t npairs);
int svamd_scan_grid(const double *left, int64_t left_pitch, const double *right, int64_t right_pitch,
                    const int32_t *xcand, const int32_t *xwin, int ncol, int nx,
                    const int32_t *ycand, const int32_t *ywin, int nrow, int ny,
                    const int64_t *row_first, int row_begin, int row_end, int64_t npairs,
                    int max_side, double min_std, double min_mean_diff, double threshold,
                    double centre_tol, int precision, double guard_band, int32_t *best_slot,
                    double *best_val, void *workspace, size_t workspace_bytes, void *stream);

/* ---- dense band scan ------------------------------------------------------------
 * For every win_w x win_h window of the left image (origins on a stride grid,
 * windows fully inside the image) and every horizontal offset d in
 * [-max_disp, +max_disp] whose box lies inside the right image, the spatial NCC
 * of correlation.py:104-116 at full overlap, i.e. row 0 of
 * cross_correlate_2d(window, band); then per window either utils.max_pos's
 * arg-max (utils.py:17-26) or image_scan's selection rule
 * (sv_image_comparison.py:249-255, centre offset first, then ascending d).
 * left, right: H x W uint8.  Windows: nwy x nwx = ((H-win_h)/stride_y+1) x
 * ((W-win_w)/stride_x+1).  out_disp: int32 (SVAMD_BAND_NO_MATCH when arg-max
 * found no finite score; 0 when the rule selected nothing).  out_val: float64.
 * scores (nullable): nwy x nwx x (2*max_disp+1) float64, NaN where invalid.
 * Rows [row_begin, row_end) of the window grid are processed (for sharding
 * over devices); outputs are indexed by absolute window row.
 * flags: 0, or SVAMD_BAND_PREPARED when svamd_band_prepare_u8 has already run on
 * these images into this workspace (e.g. one prepare, then one scan per row
 * shard).  Limits: win_w * win_h <= 1024 (exact float32 accumulation);
 * stride_x == 1; win_h rows of a 256-window-wide tile must fit in LDS (about 70
 * rows for narrow windows).  Window widths 3 ... 32 each have a fully unrolled kernel;
 * any other width runs through a kernel that takes it at run time. */
size_t svamd_band_scan_u8_workspace_bytes(int H, int W, int win_w, int win_h);
/* 1 when svamd_band_scan_u8 can run this window / row stride / offset range with
 * SVAMD_BAND_BOX_SUMS (a tile's tables and the right band under all offsets must fit the
 * LDS; win_w <= 257; window area up to 4096, where the multiply-add kernel stops at
 * 1024), else 0.  No counterpart in the reference. */
int svamd_band_box_sums_supported(int win_w, int win_h, int stride_y, int max_disp);

/* K1 for the band scan: summed-area tables of both images and, from them, the
 * per-position window sums and inverse deviations, into the workspace. */
int svamd_band_prepare_u8(const uint8_t *left, const uint8_t *right, int H, int W,
                          int win_w, int win_h, void *workspace, size_t workspace_bytes,
                          void *stream);
int svamd_band_scan_u8(const uint8_t *left, const uint8_t *right, int H, int W,
                       int win_w, int win_h, int stride_x, int stride_y, int max_disp,
                       int mode, double threshold, double centre_tol,
                       int row_begin, int row_end, int flags,
                       int32_t *out_disp, double *out_val, double *scores,
                       void *workspace, size_t workspace_bytes, void *stream);

/* The dense band scan for float64 images (what utils.downsample, utils.py:28-35, returns
 * for factors above 1, where pixels are no longer 8-bit levels).  Same windows, offsets,
 * modes and outputs as svamd_band_scan_u8.  The cross terms are accumulated in float32
 * from pixels centred so that the products are products of deviations (scores within
 * about 1e-6 of a float64 evaluation); every window whose closest decision came within
 * guard_band of going the other way is then evaluated again in float64, all offsets,
 * and its result replaced, so the selected offsets are those of a float64 evaluation
 * (guard_band 0 skips that).  Window statistics come from float64 summed-area tables of
 * the centred pixels; windows too flat for the tables' rounding take them from their own
 * pixels, and exactly constant windows give NaN scores (the reference's 0 / 0).
 * Images with NaN pixels are not supported.  Limits: win_w * win_h <= 4096,
 * stride_x == 1, the LDS tile as above.  Widths 5, 7, 9-12, 15, 16, 20, 21, 24, 25 and 31 have unrolled kernels. */
size_t svamd_band_scan_f64_workspace_bytes(int H, int W, int win_w, int win_h);
/* byte offset, inside that workspace, of an int32 that holds after a scan the number of
 * windows that were evaluated again in float64 (a diagnostic) */
size_t svamd_band_scan_f64_recheck_count_offset(int H, int W, int win_w, int win_h);
int svamd_band_prepare_f64(const double *left, const double *right, int H, int W,
                           int win_w, int win_h, void *workspace, size_t workspace_bytes,
                           void *stream);
int svamd_band_scan_f64(const double *left, const double *right, int H, int W,
                        int win_w, int win_h, int stride_x, int stride_y, int max_disp,
                        int mode, double threshold, double centre_tol, double guard_band,
                        int row_begin, int row_end, int flags,
                        int32_t *out_disp, double *out_val, double *scores,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- displacement grids: the accumulation after a scan ---------------------------
 * test_image_comparison.py:61-92.  Every window of a sequence paints dp_x, dp_y,
 * |dp| and (if |dp| > 0) a count of 1 over its box of four H x W float64 grids;
 * a sequence's grids start from zero and are then added to the running totals
 * (:89-92).  boxes: n x 4 int32 {x_start, x_end, y_start, y_end} (half open, as
 * the slices at :84-88; empty boxes paint nothing), vals: n x 3 float64 {dp_x,
 * dp_y, |dp|}, both in the reference's visiting order; per pixel the values are
 * summed in that order, so the totals match the reference bit for bit.
 * max_box_w / max_box_h: upper bounds of the box sides (they size the
 * workspace).  depth, dp_x, dp_y, contrib: the running totals, updated in
 * place (zero them before the first sequence). */
size_t svamd_accumulate_boxes_workspace_bytes(int64_t n, int H, int W, int max_box_w, int max_box_h);
int svamd_accumulate_boxes(const int32_t *boxes, const double *vals, int64_t n, int H, int W,
                           int max_box_w, int max_box_h, double *depth, double *dp_x, double *dp_y,
                           double *contrib, void *workspace, size_t workspace_bytes, void *stream);

/* The same accumulation for the dense band scan's disparity map: window
 * (wy, wx) of the nwy x nwx grid paints disp[wy][wx] (SVAMD_BAND_NO_MATCH counts
 * as 0; dp_y is 0 throughout) over rows [wy*stride_y, wy*stride_y+win_h) and
 * columns [wx, wx+win_w).  Sums of integers: exact. */
size_t svamd_band_accumulate_workspace_bytes(int H, int W, int win_h, int stride_y);
int svamd_band_accumulate(const int32_t *disp, int H, int W, int win_w, int win_h, int stride_y,
                          double *depth, double *dp_x, double *contrib, void *workspace,
                          size_t workspace_bytes, void *stream);

/* test_image_comparison.py:95-98: contributions of 0 become infinity, then the
 * three grids are divided by the contributions, in place. */
int svamd_grid_average(double *depth, double *dp_x, double *dp_y, double *contrib, int64_t total,
                       void *stream);

/* ---- calibration peak finder: local maxima of a correlation surface -------------
 * sv_calibration.py:125-129.  mask[y][x] = 1 where data equals the maximum of its
 * size x size neighbourhood (scipy.ndimage.maximum_filter's window and reflected
 * borders) and that maximum exceeds the neighbourhood's minimum by more than
 * `threshold`, else 0.  data: H x W float64 without NaNs; mask: H x W uint8. */
size_t svamd_peak_mask_workspace_bytes(int H, int W);
int svamd_peak_mask(const double *data, int H, int W, int size, double threshold, uint8_t *mask,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---- image preparation ahead of the scans --------------------------------------
 * utils.py:6-15 (and the `.convert('L')` calls of the drivers, e.g.
 * test_image_comparison.py:36-44).  pixels: H x W x channels uint8, interleaved
 * (1 = grey, 2 = grey + alpha, 3 = RGB, 4 = RGBA).  Grey level = PIL's "L":
 * (19595 R + 38470 G + 7471 B + 32768) >> 16.  grey_u8 (H x W uint8) and / or
 * grey_f64 (H x W float64) receive it; with nan_alpha != 0 the float64 output is
 * NaN where alpha == 0 (utils.py:13-14).  Either output may be NULL, not both. */
int svamd_greyscale(const uint8_t *pixels, int channels, int H, int W, int nan_alpha, uint8_t *grey_u8,
                    double *grey_f64, void *stream);

/* utils.py:28-35: scikit-image 0.19 `resize(image, (out_h, out_w))` with its
 * defaults, i.e. uint8 values times 1/255 (float64 images unchanged), a Gaussian
 * filter of sigma (in/out - 1)/2 along each axis that shrinks (radius
 * int(4 sigma + 0.5), borders mirrored about the edge pixel), linear
 * interpolation at (k + 0.5) in/out - 0.5, and a clip to the input's range.
 * image: H x W of `dtype` (SVAMD_U8 or SVAMD_F64); out: out_h x out_w float64.
 * NaNs spread to every output the filter or the interpolation draws them into,
 * as they do in the reference.  Reduction factors up to 128. */
size_t svamd_downsample_workspace_bytes(int H, int W);
int svamd_downsample(const void *image, int dtype, int H, int W, int out_h, int out_w, double *out,
                     void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVAMD_H */
