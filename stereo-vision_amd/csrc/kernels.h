// Internal launch helpers shared between the translation units of libsvamd.
#pragma once

#include "common.h"

namespace svamd {

// sat.hip
// Summed-area tables of (img - offset) and its square, (h + 1) x (w + 1) each.  With
// a scratch buffer (any 256-byte aligned device memory that is free until the call's
// kernels have run; 2 * 32 * (w + 1) table elements are as much as is ever used) tall
// tables are scanned down their columns in row blocks; without one, by one thread per
// column.  `device_offset`, when given, replaces `offset` by a double read on the device.
template <typename TIn, typename TAcc>
int sat2d_launch(const TIn *img, int h, int w, int64_t pitch, TAcc offset, TAcc *s1, TAcc *s2,
                 hipStream_t stream, void *scratch = nullptr, size_t scratch_bytes = 0,
                 const double *device_offset = nullptr);
// ... for an image of dtype SVAMD_U8 (int64 tables), SVAMD_F32 or SVAMD_F64 (float64 tables)
int sat2d_any(const void *img, int dtype, int h, int w, int64_t pitch, double offset, void *s1, void *s2,
              hipStream_t stream, void *scratch, size_t scratch_bytes, const double *device_offset = nullptr);

// xcorr_slide.hip
// Two-pass mean / population std of an h x w array (NaNs optionally ignored).
// `scratch`: stats_scratch_doubles() doubles, reusable by later calls on the stream.
size_t stats_scratch_doubles();
int stats_launch(const void *src, int dtype, int h, int w, int64_t pitch, bool ignore_nan, Stats *out,
                 double *scratch, hipStream_t stream);

// Sliding dot product of a float32 template over a float32 image:
//   out[i][j] = sum_{m<th, n<tw} t[m][n] * img[i*sy + m][j*sx + n]
// for i < oh, j < ow.  The image has eh valid rows of ew valid columns (row pitch
// img_pitch, a multiple of 4 unless eh == 1); anything beyond them reads as 0
// (rows: only with sx == 1; with sx > 1 the image must hold every row touched).
// Products are accumulated in float32 over short runs and the runs in float64.
// `pack`: slide_pack_floats() floats.  `partial`: with sx == 1
// slide_partial_doubles() doubles; with sx > 1 `partial_cap` doubles, which
// bounds how many ways the strided kernel may split the template columns
// (slide_strided_partial_doubles() is the customary amount; 0 = no split).
// `planes_left`: when given, partial planes are not added up: their number is
// returned (1: the sums are in `out`; more: plane z is partial + z * oh * ow) and
// the caller's next kernel adds them.
size_t slide_pack_floats(int th, int tw);
size_t slide_partial_doubles(int tw, int64_t oh, int64_t ow);
size_t slide_strided_partial_doubles(int64_t rh, int64_t rw);
int slide_launch(const float *t, int th, int tw, const float *img, int64_t img_pitch, int eh, int ew,
                 int oh, int ow, int sx, int sy, double *out, float *pack, double *partial,
                 hipStream_t stream, size_t partial_cap = 0, int *planes_left = nullptr);

// fft_xcorr.hip
// Plain circular correlation of two P x Q float32 arrays (P, Q powers of two within
// svamd_fft_supported) through the LDS FFT kernels, float32 result:
//   out[k][l] = sum_{m,n} a[m][n] * b[(m + k) mod P][(n + l) mod Q]
size_t fft_raw_workspace_bytes(int P, int Q);
int fft_raw_correlate(const float *a, const float *b, int P, int Q, float *out, void *workspace,
                      hipStream_t stream);

// scan_pairs.hip
// The pair kernels behind svamd_scan_pairs (same arguments).  `list` / `list_count`
// (float64 accumulation only): evaluate the pairs list[0 .. *list_count), a count that
// an earlier kernel of the stream left on the device, with `list_grid` workgroups, and
// leave each result at its pair's own index; without a list, pairs 0 .. npairs - 1.
int launch_pairs(const double *tbase, const double *rbase, const int64_t *desc, int64_t npairs, int max_side,
                 double min_std, double min_mean_diff, int acc_precision, double *out_max, int64_t *out_argmax,
                 double *surfaces, const int64_t *surf_off, const int32_t *list, const int32_t *list_count,
                 unsigned list_grid, hipStream_t st);

// band_sums.hip
// svamd_band_scan_u8's scan by box sums of the per-offset product images (same tables, same
// results as the multiply-add kernel).  band_sums_fits: whether a tile of this window and
// offset range fits the LDS.
bool band_sums_fits(int win_w, int win_h, int stride_y, int max_disp);
int band_sums_launch(const uint8_t *left, const uint8_t *right, int H, int W, int win_w, int win_h, int stride_y,
                     int max_disp, int mode, double thr, double tol, int row_begin, int row_end,
                     const int32_t *wl, const double *il, const int32_t *wr, const double *ir, int32_t *out_disp,
                     double *out_val, double *scores, hipStream_t st);

}  // namespace svamd
