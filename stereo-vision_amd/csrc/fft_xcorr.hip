// K3-large: batched circular 2D cross-correlation through FFTs held in LDS, for
// power-of-two region shapes (rows and columns 4..4096).  Per pair:
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
// The kernels of this file give every transform a whole workgroup, with a barrier per
// stage; they serve the shapes with a side below 256.  When both sides are 256 ... 4096
// the same three passes run with N / 16 threads per transform (fft_wide.hip).
//
// Transforms are radix-4 (one radix-2 stage when log2 N is odd), in place:
// decimation in frequency forward (natural order in, digit-reversed out) and
// decimation in time backward (digit-reversed in, natural out), so the pointwise
// product in pass 2 needs no reordering at all.  Twiddles come from a table built
// in double precision.
#include "fft_common.h"
#include "kernels.h"

namespace svamd {

constexpr int FFT_MAX_N = 4096;
constexpr int FFT_COLS = 8;          // spectrum columns per tile in pass 2 (fewer for more than 1024 rows)
constexpr int FFT_THREADS = 256;

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
    const int64_t step = (int64_t)gridDim.x * blockDim.x, first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // eight loads in flight per thread
    auto accumulate = [&](const float *__restrict__ p, int64_t n, double &sum, double &sq) {
        for (int64_t base = first; base < n; base += 8 * step) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = base + k * step < n ? p[base + k * step] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double d = v[k];
                sum += d;
                sq += d * d;
            }
        }
    };
    accumulate(tp, t_elems, ts, tq);
    accumulate(rp, r_elems, rs, rq);
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
// One workgroup per (row, pair).  spec_t / spec_r: [pair][H][hp] complex, hp = W/2 + 16.
// raw != 0: the arrays are transformed as they are (no mean removed).
__global__ void __launch_bounds__(1024)
fft_rows_forward_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H,
                        int W, int log2w, const FftStats *__restrict__ stats,
                        const float2 *__restrict__ twiddle, float2 *__restrict__ spec_t,
                        float2 *__restrict__ spec_r, int hp, int raw)
{
    extern __shared__ __align__(16) float2 rows_lds[];
    float2 *buf = rows_lds, *twl = rows_lds + W;
    const int y = blockIdx.x, pair = blockIdx.y;
    float tmean = 0.f, rmean = 0.f;
    if (!raw) {
        const FftStats st = stats[pair];
        tmean = (float)(st.t_sum / ((double)th * tw_));
        rmean = (float)(st.r_sum / ((double)H * W));
    }
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
// One workgroup per (tile of `cols` spectrum columns, pair).  The T and R tiles are
// interleaved in LDS as [H][2 * cols] so that one call transforms all 2 * cols
// sequences; P = conj(T) R overwrites the T half, whose inverse transform is then
// stored over spec_t.  cols = 8 up to 1024 rows, 4 for 2048, 2 for 4096 (the tile
// and the twiddle table then fill the 160 KB of LDS).
constexpr int FFT_COL_THREADS = 1024;

__global__ void __launch_bounds__(FFT_COL_THREADS)
fft_cols_fused_kernel(float2 *__restrict__ spec_t, const float2 *__restrict__ spec_r, int H, int log2h,
                      int ncols, int hp, int cols, const float2 *__restrict__ twiddle)
{
    extern __shared__ __align__(16) float2 lds[];
    const int S = 2 * cols;
    float2 *tile = lds;                     // [H][S]
    float2 *twl = lds + H * S;              // [H]
    // Workgroups go to the 8 XCDs in turn, and a tile is a fraction of a 128-byte line in every
    // row: consecutive tiles are given to workgroups of ONE XCD (block b -> tile (b % 8) * n/8 +
    // b / 8, the grid is a multiple of 8 wide), so that the neighbours that share lines share an L2.
    const int per_xcd = gridDim.x / 8;
    const int tile_index = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int c0 = tile_index * cols, pair = blockIdx.y;
    if (c0 >= ncols)
        return;
    float2 *gt = spec_t + (int64_t)pair * H * hp;
    const float2 *gr = spec_r + (int64_t)pair * H * hp;
    for (int i = threadIdx.x; i < H * cols; i += blockDim.x) {
        const int y = i / cols, c = i % cols;
        const bool in = c0 + c < ncols;
        tile[y * S + c] = in ? gt[(int64_t)y * hp + c0 + c] : make_float2(0.f, 0.f);
        tile[y * S + cols + c] = in ? gr[(int64_t)y * hp + c0 + c] : make_float2(0.f, 0.f);
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x)
        twl[i] = twiddle[i];
    __syncthreads();
    fft_forward(tile, H, log2h, S, S, twl);
    for (int i = threadIdx.x; i < H * cols; i += blockDim.x) {
        const int y = i / cols, c = i % cols;
        const float2 tv = tile[y * S + c], rv = tile[y * S + cols + c];
        tile[y * S + c] = cmul_conj(rv, tv);            // conj(T) * R
    }
    __syncthreads();
    fft_inverse(tile, H, log2h, cols, S, twl);
    for (int i = threadIdx.x; i < H * cols; i += blockDim.x) {
        const int y = i / cols, c = i % cols;
        if (c0 + c < ncols)
            gt[(int64_t)y * hp + c0 + c] = tile[y * S + c];
    }
}

// ---- pass 3: inverse row transform of two rows at a time, scaling ---------------------
// raw != 0: the plain circular correlation sum (scaled by 1 / (H W) only).
template <typename TOut>
__global__ void __launch_bounds__(1024)
fft_rows_inverse_kernel(const float2 *__restrict__ spec, int th, int tw_, int H, int W, int log2w, int hp,
                        const FftStats *__restrict__ stats, double min_std, double min_mean_diff,
                        const float2 *__restrict__ twiddle, TOut *__restrict__ out, int raw)
{
    extern __shared__ __align__(16) float2 rows_lds[];
    float2 *buf = rows_lds, *twl = rows_lds + W;
    const int y0 = 2 * blockIdx.x, pair = blockIdx.y;
    TOut *o0 = out + ((int64_t)pair * H + y0) * W;
    TOut *o1 = o0 + W;
    const bool second = y0 + 1 < H;
    double scale = 1.0 / ((double)H * (double)W);
    if (!raw) {
        const FftStats st = stats[pair];
        const double tn = (double)th * tw_, rn = (double)H * W;
        const double tmean = st.t_sum / tn, rmean = st.r_sum / rn;
        const double tvar = st.t_sq / tn - tmean * tmean, rvar = st.r_sq / rn - rmean * rmean;
        const double tstd = sqrt(fmax(tvar, 0.0));
        // a one-pass variance cannot resolve a relative spread below ~1e-14: constant region
        const double rstd = rvar > 1e-14 * (st.r_sq / rn) ? sqrt(rvar) : 0.0;
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
        scale = scale / (tstd * tn) / rstd;
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
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        o0[x] = (TOut)((double)buf[x].x * scale);
        if (second)
            o1[x] = (TOut)((double)buf[x].y * scale);
    }
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
    const size_t hp = (size_t)W / 2 + 16;
    return align_up((size_t)batch * sizeof(FftStats), 256) + 2 * align_up(FFT_MAX_N * sizeof(float2), 256) +
           2 * align_up((size_t)batch * H * hp * sizeof(float2), 256) +
           align_up((size_t)batch * H * 4 * wide_row_parts(W) * sizeof(double), 256);
}

namespace {
// LDS limits are per function and per device; raising one is cheap, so it is done
// before every launch that needs more than the default 64 KB.
template <typename K>
hipError_t allow_lds(K kernel, size_t bytes)
{
    return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                             : hipSuccess;
}

int fft_correlate(const float *tmpl, int th, int tw, const float *region, int H, int W, int batch,
                  double min_std, double min_mean_diff, void *out, int out_dtype, void *workspace, int raw,
                  hipStream_t st)
{
    const int hp = W / 2 + 16, ncols = W / 2 + 1;   // row pitch: an odd number of 128-byte lines for W = 1024
    const int log2w = ilog2(W), log2h = ilog2(H);
    Carver ws(workspace);
    FftStats *stats = ws.take<FftStats>((size_t)batch);
    float2 *tw_w = ws.take<float2>(FFT_MAX_N);
    float2 *tw_h = ws.take<float2>(FFT_MAX_N);
    float2 *spec_t = ws.take<float2>((size_t)batch * H * hp);
    float2 *spec_r = ws.take<float2>((size_t)batch * H * hp);
    double *row_sums = ws.take<double>((size_t)batch * H * 4 * wide_row_parts(W));

    if (wide_size(H) && wide_size(W))           // N / 16 threads per transform (fft_wide.hip)
        return wide_correlate(tmpl, th, tw, region, H, W, batch, min_std, min_mean_diff, out, out_dtype, stats,
                              tw_w, tw_h, spec_t, spec_r, row_sums, hp, raw, st);
    SV_HIP(hipMemsetAsync(stats, 0, (size_t)batch * sizeof(FftStats), st));
    hipLaunchKernelGGL(twiddle_kernel, dim3((W + 255) / 256), dim3(256), 0, st, tw_w, W);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(twiddle_kernel, dim3((H + 255) / 256), dim3(256), 0, st, tw_h, H);
    SV_LAUNCH_CHECK();
    if (!raw) {
        const int64_t t_elems = (int64_t)th * tw, r_elems = (int64_t)H * W;
        const int sblocks = (int)std::min<int64_t>(batch >= 16 ? 64 : 1024, (r_elems + 4095) / 4096);
        hipLaunchKernelGGL(fft_stats_kernel, dim3(sblocks, batch), dim3(256), 0, st, tmpl, t_elems, region,
                           r_elems, stats);
        SV_LAUNCH_CHECK();
    }
    const size_t row_lds = (size_t)2 * W * sizeof(float2);
    // 4096-point rows: two workgroups fit a CU's LDS, so they had better be large
    const int row_threads = W >= 4096 ? 1024 : FFT_THREADS;
    SV_HIP(allow_lds(&fft_rows_forward_kernel, row_lds));
    hipLaunchKernelGGL(fft_rows_forward_kernel, dim3(H, batch), dim3(row_threads), row_lds, st, tmpl, th, tw,
                       region, H, W, log2w, stats, tw_w, spec_t, spec_r, hp, raw);
    SV_LAUNCH_CHECK();
    const int cols = H <= 1024 ? FFT_COLS : H == 2048 ? 4 : 2;
    const size_t lds = ((size_t)H * 2 * cols + H) * sizeof(float2);
    SV_HIP(allow_lds(&fft_cols_fused_kernel, lds));
    hipLaunchKernelGGL(fft_cols_fused_kernel, dim3(((ncols + cols - 1) / cols + 7) / 8 * 8, batch), dim3(FFT_COL_THREADS), lds,
                       st, spec_t, spec_r, H, log2h, ncols, hp, cols, tw_h);
    SV_LAUNCH_CHECK();
    dim3 g3((H + 1) / 2, batch);
    if (out_dtype == SVAMD_F32) {
        SV_HIP(allow_lds(&fft_rows_inverse_kernel<float>, row_lds));
        hipLaunchKernelGGL((fft_rows_inverse_kernel<float>), g3, dim3(row_threads), row_lds, st, spec_t, th, tw, H,
                           W, log2w, hp, stats, min_std, min_mean_diff, tw_w, static_cast<float *>(out), raw);
    } else {
        SV_HIP(allow_lds(&fft_rows_inverse_kernel<double>, row_lds));
        hipLaunchKernelGGL((fft_rows_inverse_kernel<double>), g3, dim3(row_threads), row_lds, st, spec_t, th, tw, H,
                           W, log2w, hp, stats, min_std, min_mean_diff, tw_w, static_cast<double *>(out), raw);
    }
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
}  // namespace

namespace svamd {
// Plain circular correlation of two P x Q float32 arrays through the FFT kernels:
// out[k][l] = sum_{m,n} a[m][n] * b[(m + k) mod P][(n + l) mod Q], float32.
size_t fft_raw_workspace_bytes(int P, int Q) { return svamd_xcorr2d_fft_workspace_bytes(1, P, Q); }

int fft_raw_correlate(const float *a, const float *b, int P, int Q, float *out, void *workspace,
                      hipStream_t stream)
{
    SV_REQUIRE(svamd_fft_supported(P, Q), SVAMD_ERR_LIMIT, "fft: %dx%d is not a supported power-of-two shape", P, Q);
    return fft_correlate(a, P, Q, b, P, Q, 1, 0.0, 0.0, out, SVAMD_F32, workspace, 1, stream);
}
}  // namespace svamd

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
    return fft_correlate(tmpl, th, tw, region, H, W, batch, min_std, min_mean_diff, out, out_dtype, workspace, 0,
                         static_cast<hipStream_t>(stream));
}
