// K3-large: batched circular 2D cross-correlation through FFTs held in LDS, for
// power-of-two region shapes (rows and columns 16..1024).  Per pair:
//
//   pass 1 (rows)     z = (t - mean t) + i (r - mean r) per image row, one complex
//                     FFT of length W in LDS, split into the two half spectra
//                     T(y, kx), R(y, kx), kx = 0..W/2 (two real transforms for the
//                     price of one complex one);
//   pass 2 (columns)  a tile of 8 spectrum columns of T and of R in LDS: forward
//                     FFT along y of both, P = conj(T) * R, inverse FFT along y of
//                     P - conjugate-multiply and inverse transform fused, nothing
//                     but P' goes back to memory;
//   pass 3 (rows)     two rows of P' per workgroup, Hermitian completion, one
//                     complex inverse FFT whose real / imaginary parts are the two
//                     output rows; scaling 1 / (H W std t * size t * std r) or the
//                     constant surface of the low-deviation branch.
//
// Transforms are radix-4 (one radix-2 stage when log2 N is odd), in place:
// decimation in frequency forward (natural order in, digit-reversed out) and
// decimation in time backward (digit-reversed in, natural out), so the pointwise
// product in pass 2 needs no reordering at all.  Twiddles come from a table built
// in double precision.
#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int FFT_MAX_N = 1024;
constexpr int FFT_COLS = 8;          // spectrum columns per tile in pass 2
constexpr int FFT_THREADS = 256;

struct FftStats {                    // per pair, accumulated with float64 atomics
    double t_sum, t_sq, r_sum, r_sq;
};

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

// tw[j] = exp(-2 pi i j / n), j < n
__global__ void __launch_bounds__(256)
twiddle_kernel(float2 *__restrict__ tw, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        double s, c;
        sincospi(-2.0 * (double)j / (double)n, &s, &c);
        tw[j] = make_float2((float)c, (float)s);
    }
}

// base-4 digit reversal of k (with the radix-2 digit, if any, as the leading one):
// the position at which the forward transform leaves frequency k.
__device__ __forceinline__ int digit_reverse(int k, int log2n)
{
    int out = 0, bits = log2n;
    if (bits & 1) {                   // the radix-2 stage runs first: it decides the top bit
        out = (k & 1) << (bits - 1);  // frequency parity -> upper / lower half
        k >>= 1;
        bits -= 1;
    }
    for (int b = 0; b < bits; b += 2) {
        out |= (k & 3) << (bits - 2 - b);
        k >>= 2;
    }
    return out;
}

// In-place forward FFT (DIF) of `count` interleaved sequences of length n:
// element j of sequence s lives at x[j * stride + s].  All threads of the block
// take part; ends with a barrier.
__device__ __forceinline__ void fft_forward(float2 *x, int n, int log2n, int count, int stride,
                                            const float2 *__restrict__ tw)
{
    int len = n;
    if (log2n & 1) {
        const int half = n / 2;
        for (int i = threadIdx.x; i < half * count; i += blockDim.x) {
            const int s = i % count, j = i / count;
            float2 *p = x + j * stride + s;
            const float2 a = p[0], b = p[half * stride];
            p[0] = cadd(a, b);
            p[half * stride] = cmul(csub(a, b), tw[j]);
        }
        len = half;
        __syncthreads();
    }
    for (; len >= 4; len >>= 2) {
        const int q = len / 4, tstep = n / len;
        const int groups = n / len;
        for (int i = threadIdx.x; i < groups * q * count; i += blockDim.x) {
            const int s = i % count, rest = i / count;
            const int j = rest % q, g = rest / q;
            float2 *p = x + (g * len + j) * stride + s;
            const float2 a = p[0], b = p[q * stride], c = p[2 * q * stride], d = p[3 * q * stride];
            const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
            p[0] = cadd(apc, bpd);
            p[q * stride] = cmul(cadd(amc, mul_mi(bmd)), tw[j * tstep]);
            p[2 * q * stride] = cmul(csub(apc, bpd), tw[2 * j * tstep]);
            p[3 * q * stride] = cmul(cadd(amc, mul_pi(bmd)), tw[3 * j * tstep]);
        }
        __syncthreads();
    }
}

// In-place inverse FFT (DIT) from digit-reversed order, unscaled.
__device__ __forceinline__ void fft_inverse(float2 *x, int n, int log2n, int count, int stride,
                                            const float2 *__restrict__ tw)
{
    const int top = (log2n & 1) ? n / 2 : n;
    for (int len = 4; len <= top; len <<= 2) {
        const int q = len / 4, tstep = n / len;
        const int groups = n / len;
        for (int i = threadIdx.x; i < groups * q * count; i += blockDim.x) {
            const int s = i % count, rest = i / count;
            const int j = rest % q, g = rest / q;
            float2 *p = x + (g * len + j) * stride + s;
            const float2 a = p[0];
            const float2 b = cmul_conj(p[q * stride], tw[j * tstep]);
            const float2 c = cmul_conj(p[2 * q * stride], tw[2 * j * tstep]);
            const float2 d = cmul_conj(p[3 * q * stride], tw[3 * j * tstep]);
            const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
            p[0] = cadd(apc, bpd);
            p[q * stride] = cadd(amc, mul_pi(bmd));
            p[2 * q * stride] = csub(apc, bpd);
            p[3 * q * stride] = cadd(amc, mul_mi(bmd));
        }
        __syncthreads();
    }
    if (log2n & 1) {
        const int half = n / 2;
        for (int i = threadIdx.x; i < half * count; i += blockDim.x) {
            const int s = i % count, j = i / count;
            float2 *p = x + j * stride + s;
            const float2 a = p[0], b = cmul_conj(p[half * stride], tw[j]);
            p[0] = cadd(a, b);
            p[half * stride] = csub(a, b);
        }
        __syncthreads();
    }
}

// ---- statistics: sum and sum of squares of every template and region ---------------
__global__ void __launch_bounds__(256)
fft_stats_kernel(const float *__restrict__ t, int64_t t_elems, const float *__restrict__ r, int64_t r_elems,
                 FftStats *__restrict__ stats)
{
    __shared__ double red[17];
    const int pair = blockIdx.y;
    const float *tp = t + (int64_t)pair * t_elems;
    const float *rp = r + (int64_t)pair * r_elems;
    double ts = 0, tq = 0, rs = 0, rq = 0;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < t_elems; i += step) {
        const double v = tp[i];
        ts += v;
        tq += v * v;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < r_elems; i += step) {
        const double v = rp[i];
        rs += v;
        rq += v * v;
    }
    ts = block_sum(ts, red);
    tq = block_sum(tq, red);
    rs = block_sum(rs, red);
    rq = block_sum(rq, red);
    if (threadIdx.x == 0) {
        atomicAdd(&stats[pair].t_sum, ts);
        atomicAdd(&stats[pair].t_sq, tq);
        atomicAdd(&stats[pair].r_sum, rs);
        atomicAdd(&stats[pair].r_sq, rq);
    }
}

// ---- pass 1: packed row transform ---------------------------------------------------
// One workgroup per (row, pair).  spec_t / spec_r: [pair][H][hp] complex, hp = W/2 + 8.
__global__ void __launch_bounds__(FFT_THREADS)
fft_rows_forward_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H,
                        int W, int log2w, const FftStats *__restrict__ stats,
                        const float2 *__restrict__ twiddle, float2 *__restrict__ spec_t,
                        float2 *__restrict__ spec_r, int hp)
{
    __shared__ float2 buf[FFT_MAX_N];
    __shared__ float2 twl[FFT_MAX_N];
    const int y = blockIdx.x, pair = blockIdx.y;
    const FftStats st = stats[pair];
    const float tmean = (float)(st.t_sum / ((double)th * tw_));
    const float rmean = (float)(st.r_sum / ((double)H * W));
    const float *trow = t + ((int64_t)pair * th + y) * tw_;
    const float *rrow = r + ((int64_t)pair * H + y) * W;
    const int mh = min(th, H), mw = min(tw_, W);   // template zero-padded or cropped to the region
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const float tv = (y < mh && x < mw) ? trow[x] - tmean : 0.0f;
        buf[x] = make_float2(tv, rrow[x] - rmean);
        twl[x] = twiddle[x];
    }
    __syncthreads();
    fft_forward(buf, W, log2w, 1, 1, twl);
    // Z(k) = T(k) + i R(k) with T, R Hermitian:  T(k) = (Z(k) + conj Z(-k)) / 2,
    // R(k) = (Z(k) - conj Z(-k)) / (2i)
    float2 *ot = spec_t + ((int64_t)pair * H + y) * hp;
    float2 *orr = spec_r + ((int64_t)pair * H + y) * hp;
    for (int k = threadIdx.x; k <= W / 2; k += blockDim.x) {
        const float2 zk = buf[digit_reverse(k, log2w)];
        const float2 zm = buf[digit_reverse((W - k) & (W - 1), log2w)];
        ot[k] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        orr[k] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
    }
}

// ---- pass 2: column transforms, conjugate multiply, inverse column transform --------
// One workgroup per (tile of FFT_COLS spectrum columns, pair).  The T and R tiles
// are interleaved in LDS as [H][2 * FFT_COLS] so that one call transforms all 16
// sequences; P = conj(T) R overwrites the T half, whose inverse transform is then
// stored over spec_t.
constexpr int FFT_COL_THREADS = 1024;

__global__ void __launch_bounds__(FFT_COL_THREADS)
fft_cols_fused_kernel(float2 *__restrict__ spec_t, const float2 *__restrict__ spec_r, int H, int log2h,
                      int ncols, int hp, const float2 *__restrict__ twiddle)
{
    extern __shared__ __align__(16) float2 lds[];
    constexpr int S = 2 * FFT_COLS;
    float2 *tile = lds;                     // [H][S]
    float2 *twl = lds + H * S;              // [H]
    const int c0 = blockIdx.x * FFT_COLS, pair = blockIdx.y;
    float2 *gt = spec_t + (int64_t)pair * H * hp;
    const float2 *gr = spec_r + (int64_t)pair * H * hp;
    for (int i = threadIdx.x; i < H * FFT_COLS; i += blockDim.x) {
        const int y = i / FFT_COLS, c = i % FFT_COLS;
        const bool in = c0 + c < ncols;
        tile[y * S + c] = in ? gt[(int64_t)y * hp + c0 + c] : make_float2(0.f, 0.f);
        tile[y * S + FFT_COLS + c] = in ? gr[(int64_t)y * hp + c0 + c] : make_float2(0.f, 0.f);
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x)
        twl[i] = twiddle[i];
    __syncthreads();
    fft_forward(tile, H, log2h, S, S, twl);
    for (int i = threadIdx.x; i < H * FFT_COLS; i += blockDim.x) {
        const int y = i / FFT_COLS, c = i % FFT_COLS;
        const float2 tv = tile[y * S + c], rv = tile[y * S + FFT_COLS + c];
        tile[y * S + c] = cmul_conj(rv, tv);            // conj(T) * R
    }
    __syncthreads();
    fft_inverse(tile, H, log2h, FFT_COLS, S, twl);
    for (int i = threadIdx.x; i < H * FFT_COLS; i += blockDim.x) {
        const int y = i / FFT_COLS, c = i % FFT_COLS;
        if (c0 + c < ncols)
            gt[(int64_t)y * hp + c0 + c] = tile[y * S + c];
    }
}

// ---- pass 3: inverse row transform of two rows at a time, scaling ---------------------
template <typename TOut>
__global__ void __launch_bounds__(FFT_THREADS)
fft_rows_inverse_kernel(const float2 *__restrict__ spec, int th, int tw_, int H, int W, int log2w, int hp,
                        const FftStats *__restrict__ stats, double min_std, double min_mean_diff,
                        const float2 *__restrict__ twiddle, TOut *__restrict__ out)
{
    __shared__ float2 buf[FFT_MAX_N];
    __shared__ float2 twl[FFT_MAX_N];
    const int y0 = 2 * blockIdx.x, pair = blockIdx.y;
    const FftStats st = stats[pair];
    const double tn = (double)th * tw_, rn = (double)H * W;
    const double tmean = st.t_sum / tn, rmean = st.r_sum / rn;
    const double tvar = st.t_sq / tn - tmean * tmean, rvar = st.r_sq / rn - rmean * rmean;
    const double tstd = sqrt(fmax(tvar, 0.0));
    // a one-pass variance cannot resolve a relative spread below ~1e-14: constant region
    const double rstd = rvar > 1e-14 * (st.r_sq / rn) ? sqrt(rvar) : 0.0;
    TOut *o0 = out + ((int64_t)pair * H + y0) * W;
    TOut *o1 = o0 + W;
    const bool second = y0 + 1 < H;
    if (tstd < min_std || rstd == 0.0) {
        // homogeneous template: constant surface; constant region: the reference
        // divides 0 by 0 and every element is NaN
        const TOut fill = tstd < min_std ? (fabs(tmean - rmean) < min_mean_diff ? TOut(1) : TOut(0))
                                         : TOut(NAN);
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            o0[x] = fill;
            if (second)
                o1[x] = fill;
        }
        return;
    }
    const float2 *p0 = spec + ((int64_t)pair * H + y0) * hp;
    const float2 *p1 = second ? p0 + hp : p0;
    for (int k = threadIdx.x; k < W; k += blockDim.x) {
        float2 a, b;
        if (k <= W / 2) {
            a = p0[k];
            b = p1[k];
        } else {               // Hermitian completion
            a = p0[W - k];
            b = p1[W - k];
            a.y = -a.y;
            b.y = -b.y;
        }
        buf[digit_reverse(k, log2w)] = make_float2(a.x - b.y, a.y + b.x);   // a + i b
        twl[k] = twiddle[k];
    }
    __syncthreads();
    fft_inverse(buf, W, log2w, 1, 1, twl);
    const double scale = 1.0 / ((double)H * (double)W) / (tstd * tn) / rstd;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        o0[x] = (TOut)((double)buf[x].x * scale);
        if (second)
            o1[x] = (TOut)((double)buf[x].y * scale);
    }
}

inline int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return l;
}

}  // namespace svamd

using namespace svamd;

extern "C" int svamd_fft_supported(int H, int W)
{
    auto pow2 = [](int v) { return v >= 4 && v <= FFT_MAX_N && (v & (v - 1)) == 0; };
    return pow2(H) && pow2(W) ? 1 : 0;
}

extern "C" size_t svamd_xcorr2d_fft_workspace_bytes(int batch, int H, int W)
{
    if (batch <= 0 || !svamd_fft_supported(H, W))
        return 0;
    const size_t hp = (size_t)W / 2 + 8;
    return align_up((size_t)batch * sizeof(FftStats), 256) + 2 * align_up(FFT_MAX_N * sizeof(float2), 256) +
           2 * align_up((size_t)batch * H * hp * sizeof(float2), 256);
}

extern "C" int svamd_xcorr2d_fft(const float *tmpl, int th, int tw, const float *region, int H, int W,
                                 int batch, double min_std, double min_mean_diff, void *out, int out_dtype,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && region && out, SVAMD_ERR_INVALID, "xcorr2d_fft: NULL pointer");
    SV_REQUIRE(batch > 0 && th > 0 && tw > 0, SVAMD_ERR_INVALID, "xcorr2d_fft: non-positive size");
    SV_REQUIRE(svamd_fft_supported(H, W), SVAMD_ERR_LIMIT,
               "xcorr2d_fft: region %dx%d is not a power-of-two shape within 4..%d", H, W, FFT_MAX_N);
    SV_REQUIRE(out_dtype == SVAMD_F32 || out_dtype == SVAMD_F64, SVAMD_ERR_INVALID,
               "xcorr2d_fft: output must be float32 or float64");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_xcorr2d_fft_workspace_bytes(batch, H, W),
               SVAMD_ERR_WORKSPACE, "xcorr2d_fft: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int hp = W / 2 + 8, ncols = W / 2 + 1;
    const int log2w = ilog2(W), log2h = ilog2(H);
    Carver ws(workspace);
    FftStats *stats = ws.take<FftStats>((size_t)batch);
    float2 *tw_w = ws.take<float2>(FFT_MAX_N);
    float2 *tw_h = ws.take<float2>(FFT_MAX_N);
    float2 *spec_t = ws.take<float2>((size_t)batch * H * hp);
    float2 *spec_r = ws.take<float2>((size_t)batch * H * hp);

    SV_HIP(hipMemsetAsync(stats, 0, (size_t)batch * sizeof(FftStats), st));
    hipLaunchKernelGGL(twiddle_kernel, dim3((W + 255) / 256), dim3(256), 0, st, tw_w, W);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(twiddle_kernel, dim3((H + 255) / 256), dim3(256), 0, st, tw_h, H);
    SV_LAUNCH_CHECK();
    const int64_t t_elems = (int64_t)th * tw, r_elems = (int64_t)H * W;
    const int sblocks = (int)std::min<int64_t>(64, (r_elems + 4095) / 4096);
    hipLaunchKernelGGL(fft_stats_kernel, dim3(sblocks, batch), dim3(256), 0, st, tmpl, t_elems, region, r_elems,
                       stats);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_rows_forward_kernel, dim3(H, batch), dim3(FFT_THREADS), 0, st, tmpl, th, tw, region,
                       H, W, log2w, stats, tw_w, spec_t, spec_r, hp);
    SV_LAUNCH_CHECK();
    const size_t lds = ((size_t)H * 2 * FFT_COLS + H) * sizeof(float2);
    static bool raised = false;
    if (!raised) {
        SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fft_cols_fused_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    hipLaunchKernelGGL(fft_cols_fused_kernel, dim3((ncols + FFT_COLS - 1) / FFT_COLS, batch),
                       dim3(FFT_COL_THREADS), lds, st, spec_t, spec_r, H, log2h, ncols, hp, tw_h);
    SV_LAUNCH_CHECK();
    dim3 g3((H + 1) / 2, batch);
    if (out_dtype == SVAMD_F32)
        hipLaunchKernelGGL((fft_rows_inverse_kernel<float>), g3, dim3(FFT_THREADS), 0, st, spec_t, th, tw, H, W,
                           log2w, hp, stats, min_std, min_mean_diff, tw_w, static_cast<float *>(out));
    else
        hipLaunchKernelGGL((fft_rows_inverse_kernel<double>), g3, dim3(FFT_THREADS), 0, st, spec_t, th, tw, H,
                           W, log2w, hp, stats, min_std, min_mean_diff, tw_w, static_cast<double *>(out));
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
