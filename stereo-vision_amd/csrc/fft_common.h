// Pieces shared by the transform kernels of fft_xcorr.hip and fft_wide.hip.
#pragma once

#include "common.h"

namespace svamd {

struct FftStats {                    // per pair
    double t_sum, t_sq, r_sum, r_sq;  // sums of (x - centre), (x - centre)^2 (float64 atomics)
    float t_centre, r_centre;         // constants subtracted before the transforms
    float pad[2];
};

// Mean and population standard deviation from sums of (x - c) and (x - c)^2.
__device__ __forceinline__ void moments(double sum, double sq, double n, double c, double &mean, double &var)
{
    const double m = sum / n;
    mean = c + m;
    var = sq / n - m * m;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)   // a * conj(b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }   // a * (+i)

// radix-4 butterflies: decimation in frequency (twiddles after) and in time (conjugate
// twiddles before), and their twiddle-free forms
__device__ __forceinline__ void dif4(float2 &a, float2 &b, float2 &c, float2 &d, float2 w1, float2 w2, float2 w3)
{
    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    a = cadd(apc, bpd);
    b = cmul(cadd(amc, mul_mi(bmd)), w1);
    c = cmul(csub(apc, bpd), w2);
    d = cmul(cadd(amc, mul_pi(bmd)), w3);
}
__device__ __forceinline__ void dif4_plain(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    a = cadd(apc, bpd);
    b = cadd(amc, mul_mi(bmd));
    c = csub(apc, bpd);
    d = cadd(amc, mul_pi(bmd));
}
__device__ __forceinline__ void dit4(float2 &a, float2 &b, float2 &c, float2 &d, float2 w1, float2 w2, float2 w3)
{
    b = cmul_conj(b, w1);
    c = cmul_conj(c, w2);
    d = cmul_conj(d, w3);
    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    a = cadd(apc, bpd);
    b = cadd(amc, mul_pi(bmd));
    c = csub(apc, bpd);
    d = cadd(amc, mul_mi(bmd));
}
__device__ __forceinline__ void dit4_plain(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    a = cadd(apc, bpd);
    b = cadd(amc, mul_pi(bmd));
    c = csub(apc, bpd);
    d = cadd(amc, mul_mi(bmd));
}

inline int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return l;
}

// ---- fft_wide.hip: transforms of 256 ... 4096 points, N / 16 threads per transform ----
bool wide_size(int n);
// partial sums per image row that the wide pass 1 leaves for fft_row_sums_kernel
int wide_row_parts(int W);
// the whole correlation of `batch` pairs (buffers carved from the caller's workspace)
int wide_correlate(const float *tmpl, int th, int tw, const float *region, int H, int W, int batch, double min_std,
                   double min_mean_diff, void *out, int out_dtype, FftStats *stats, float2 *table_w,
                   float2 *table_h, float2 *spec_t, float2 *spec_r, double *row_sums, int hp, int raw,
                   hipStream_t st);

}  // namespace svamd
