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
// Transforms are radix-4 (one radix-2 stage when log2 N is odd), in place:
// decimation in frequency forward (natural order in, digit-reversed out) and
// decimation in time backward (digit-reversed in, natural out), so the pointwise
// product in pass 2 needs no reordering at all.  Twiddles come from a table built
// in double precision.
#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int FFT_MAX_N = 4096;
constexpr int FFT_COLS = 8;          // spectrum columns per tile in pass 2 (fewer for more than 1024 rows)
constexpr int FFT_THREADS = 256;

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

// =====================================================================================
// 1024-point fast path: one wavefront per transform, 16 elements per lane.
//
// The in-place radix-4 schedule above is kept (same positions, same digit-reversed
// output), but the five stages are grouped so that consecutive stages find their
// four operands inside one lane: lanes first hold positions lane + 64 m (stages
// of length 1024 and 256), then 64 B + s + 4 m with B = lane / 4, s = lane % 4
// (lengths 64 and 16), then 4 (lane + 64 u) + e (length 4).  Only two exchanges
// through LDS per transform, no workgroup barrier: a wavefront's LDS operations
// execute in order.  LDS positions are padded by 4 elements per 64 so that the
// strided accesses of the exchanges fall on distinct banks.
// =====================================================================================
constexpr int F1K = 1024;
constexpr int F1K_LDS = F1K + 4 * (F1K / 64);     // padded elements per sequence

// Position p of a sequence in LDS.  Besides the padding, every other block of 64
// positions has the two 16-byte halves of each run of four positions exchanged:
// the 32-byte reads and writes of positions 4 (lane + 64 u) + e, e = 0..3, start 8
// banks apart from lane to lane, so lanes 16 apart would meet on the same banks
// (the 16-lane groups of a 16-byte access mix lanes of both kinds: {0-3, 12-15,
// 20-27}, ...); with the halves exchanged they land on the other half of each 32
// bytes.  The other two access patterns (positions lane + 64 m and
// 64 (lane / 4) + lane % 4 + 4 m) stay free of conflicts under the exchange.
__device__ __forceinline__ int f1k_idx(int p)
{
    p ^= (p >> 5) & 2;
    return p + 4 * (p >> 6);
}

// The per-stage twiddle factors of the wavefront transforms, each stage's in the order
// its lanes read them (neighbouring lanes read neighbouring entries, or a few lanes'
// worth is broadcast): reads of the plain table exp(-2 pi i k / 1024) at strides 2, 4,
// 8, 12, ... elements would pile onto a few banks.
//   [0, 768)     length 1024:  w^j, w^2j, w^3j for j = 0..255
//   [768, 960)   length 256:   w^4l, w^8l, w^12l for l = 0..63
//   [960, 1008)  length 64:    w^16j, w^32j, w^48j for j = 0..15
//   [1008, 1020) length 16:    w^64s, w^128s, w^192s for s = 0..3
constexpr int F1K_TW = 1024;     // entries (1020 used)
__device__ __forceinline__ void f1k_load_twiddles(float2 *tw, const float2 *__restrict__ table, int tid, int nthreads)
{
    for (int i = tid; i < 1020; i += nthreads) {
        int k;
        if (i < 768)
            k = (i / 256 + 1) * (i % 256);
        else if (i < 960)
            k = ((i - 768) / 64 + 1) * 4 * ((i - 768) % 64);
        else if (i < 1008)
            k = ((i - 960) / 16 + 1) * 16 * ((i - 960) % 16);
        else
            k = ((i - 1008) / 4 + 1) * 64 * ((i - 1008) % 4);
        tw[i] = table[k];
    }
}

// v[4 u + e] <-> position 4 (lane + 64 u) + e, as two explicit 16-byte accesses per u
// (the exchanged halves of f1k_idx spelled out, so that they stay 16-byte accesses).
__device__ __forceinline__ void f1k_read_quads(float2 (&v)[16], const float2 *x, int lane)
{
    const int f = (lane >> 3) & 2;          // bit 6 of the position -> which half comes first
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = 4 * (lane + 64 * u);
        const float2 *base = x + p + 4 * (p >> 6);
        const float4 lo = *reinterpret_cast<const float4 *>(base + f);
        const float4 hi = *reinterpret_cast<const float4 *>(base + (2 ^ f));
        v[4 * u] = make_float2(lo.x, lo.y);
        v[4 * u + 1] = make_float2(lo.z, lo.w);
        v[4 * u + 2] = make_float2(hi.x, hi.y);
        v[4 * u + 3] = make_float2(hi.z, hi.w);
    }
}
__device__ __forceinline__ void f1k_write_quads(const float2 (&v)[16], float2 *x, int lane)
{
    const int f = (lane >> 3) & 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = 4 * (lane + 64 * u);
        float2 *base = x + p + 4 * (p >> 6);
        *reinterpret_cast<float4 *>(base + f) = make_float4(v[4 * u].x, v[4 * u].y, v[4 * u + 1].x, v[4 * u + 1].y);
        *reinterpret_cast<float4 *>(base + (2 ^ f)) =
            make_float4(v[4 * u + 2].x, v[4 * u + 2].y, v[4 * u + 3].x, v[4 * u + 3].y);
    }
}

// Layout of the half spectra between the 1024 x 1024 passes: [pair][group of 4
// columns][row][4 columns].  The column pass works on one group at a time, which is
// then 32 KB of consecutive memory instead of 1024 row pieces of 32 bytes (a
// quarter of a 128-byte line each); the row passes touch 32 consecutive bytes per
// row and group either way.  A pair keeps the footprint of the row-major layout
// (F1K x hp elements, hp >= 516).
__device__ __forceinline__ int64_t f1k_spec_index(int y, int k)
{
    return (int64_t)(k >> 2) * (F1K * 4) + y * 4 + (k & 3);
}

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

// Forward transform.  In: v[m] = position lane + 64 m (natural order).  Out:
// v[4 u + e] = position 4 (lane + 64 u) + e of the digit-reversed spectrum.
// x: this wavefront's LDS region (F1K_LDS elements); tw: the staged twiddle tables
// (f1k_load_twiddles).
__device__ __forceinline__ void f1k_forward(float2 (&v)[16], float2 *x, const float2 *__restrict__ tw, int lane)
{
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) {               // length 1024
        const int j = lane + 64 * m0;
        dif4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12], tw[j], tw[256 + j], tw[512 + j]);
    }
    {
        const float2 w1 = tw[768 + lane], w2 = tw[832 + lane], w3 = tw[896 + lane];   // length 256
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dif4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w1, w2, w3);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[f1k_idx(lane + 64 * m)] = v[m];
    __builtin_amdgcn_wave_barrier();
    const int blk = lane >> 2, s = lane & 3;
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[f1k_idx(64 * blk + s + 4 * m)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) {               // length 64
        const int j = s + 4 * m0;
        dif4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12], tw[960 + j], tw[976 + j], tw[992 + j]);
    }
    {
        const float2 w1 = tw[1008 + s], w2 = tw[1012 + s], w3 = tw[1016 + s];     // length 16
#pragma unroll
        for (int h = 0; h < 4; ++h)
            dif4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3], w1, w2, w3);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[f1k_idx(64 * blk + s + 4 * m)] = v[m];
    __builtin_amdgcn_wave_barrier();
    f1k_read_quads(v, x, lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u)                    // length 4
        dif4_plain(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
}

// Inverse transform (unscaled).  In: v[4 u + e] = digit-reversed position
// 4 (lane + 64 u) + e.  Out: v[m] = natural position lane + 64 m.
__device__ __forceinline__ void f1k_inverse(float2 (&v)[16], float2 *x, const float2 *__restrict__ tw, int lane)
{
#pragma unroll
    for (int u = 0; u < 4; ++u)                    // length 4
        dit4_plain(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
    f1k_write_quads(v, x, lane);
    __builtin_amdgcn_wave_barrier();
    const int blk = lane >> 2, s = lane & 3;
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[f1k_idx(64 * blk + s + 4 * m)];
    __builtin_amdgcn_wave_barrier();
    {
        const float2 w1 = tw[1008 + s], w2 = tw[1012 + s], w3 = tw[1016 + s];     // length 16
#pragma unroll
        for (int h = 0; h < 4; ++h)
            dit4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3], w1, w2, w3);
    }
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) {               // length 64
        const int j = s + 4 * m0;
        dit4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12], tw[960 + j], tw[976 + j], tw[992 + j]);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[f1k_idx(64 * blk + s + 4 * m)] = v[m];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[f1k_idx(lane + 64 * m)];
    __builtin_amdgcn_wave_barrier();
    {
        const float2 w1 = tw[768 + lane], w2 = tw[832 + lane], w3 = tw[896 + lane];   // length 256
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dit4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w1, w2, w3);
    }
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) {               // length 1024
        const int j = lane + 64 * m0;
        dit4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12], tw[j], tw[256 + j], tw[512 + j]);
    }
}

// Constants subtracted from the arrays before the float32 transforms, which keeps
// a large DC term out of the butterflies.  A full-size array only needs an
// approximate centre (the mean of its first row): the exact mean differs from it
// by a constant, i.e. only in the DC bin, which pass 2 clears.  A template that is
// zero-padded or cropped needs its exact mean (a constant inside a window is not a
// pure DC term).  One workgroup per pair.
__global__ void __launch_bounds__(1024)
fft_centres_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H, int W,
                   FftStats *__restrict__ stats)
{
    __shared__ double red[17];
    const int pair = blockIdx.x;
    const float *tp = t + (int64_t)pair * th * tw_;
    const float *rp = r + (int64_t)pair * H * W;
    const bool exact_t = th != H || tw_ != W;
    const int64_t tcount = exact_t ? (int64_t)th * tw_ : tw_;
    double ts = 0.0, rs = 0.0;
    for (int64_t i = threadIdx.x; i < tcount; i += blockDim.x)
        ts += tp[i];
    for (int i = threadIdx.x; i < W; i += blockDim.x)
        rs += rp[i];
    ts = block_sum(ts, red);
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) {
        FftStats st;
        st.t_sum = st.t_sq = st.r_sum = st.r_sq = 0.0;
        st.t_centre = (float)(ts / (double)tcount);
        st.r_centre = (float)(rs / (double)W);
        st.pad[0] = st.pad[1] = 0.f;
        stats[pair] = st;
    }
}

// Adds the per-row partial sums of pass 1 into the pair's statistics (fixed order:
// results do not depend on scheduling).  One workgroup per pair.
__global__ void __launch_bounds__(1024)
fft_row_sums_kernel(const double *__restrict__ row_sums, int H, FftStats *__restrict__ stats)
{
    __shared__ double red[17];
    const int pair = blockIdx.x;
    double a = 0, b = 0, c = 0, d = 0;
    for (int y = threadIdx.x; y < H; y += blockDim.x) {
        const double4 v = reinterpret_cast<const double4 *>(row_sums)[(int64_t)pair * H + y];
        a += v.x;
        b += v.y;
        c += v.z;
        d += v.w;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    c = block_sum(c, red);
    d = block_sum(d, red);
    if (threadIdx.x == 0) {
        stats[pair].t_sum = a;
        stats[pair].t_sq = b;
        stats[pair].r_sum = c;
        stats[pair].r_sq = d;
    }
}

// pass 1, W = 1024: a workgroup of four wavefronts takes 16 consecutive image rows,
// four at a time (one per wavefront), with every wavefront's next row already on its
// way while the current one is transformed (a wavefront that loads, transforms and
// stores a single row spends most of its life waiting for memory).  After the four
// transforms the whole workgroup splits the packed spectra and stores them: four rows
// x four columns of a column group are one 128-byte line (f1k_spec_index), written by
// eight neighbouring threads.  Also produces each row's exact sums of (x - centre) and
// (x - centre)^2 of both arrays.
constexpr int F1K_ROWS_PER_WAVE = 4;

__global__ void __launch_bounds__(256)
f1k_rows_forward_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H,
                        const FftStats *__restrict__ stats, double *__restrict__ row_sums,
                        const float2 *__restrict__ twiddle_table, float2 *__restrict__ spec_t,
                        float2 *__restrict__ spec_r, int hp)
{
    __shared__ float2 lds[4 * F1K_LDS];
    __shared__ float2 twiddle[F1K_TW];              // LDS copy: gathered reads stay off the memory pipeline
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y_block = blockIdx.x * 4 * F1K_ROWS_PER_WAVE, pair = blockIdx.y;   // H is a multiple of 16
    f1k_load_twiddles(twiddle, twiddle_table, threadIdx.x, 256);
    __syncthreads();
    float2 *x = lds + wave * F1K_LDS;
    const float *tbase = t + (int64_t)pair * th * tw_;
    const float *rbase = r + (int64_t)pair * H * F1K;
    const float tc = stats[pair].t_centre, rc = stats[pair].r_centre;
    const int mw = min(tw_, F1K), mh = min(th, H);
    float2 *ot = spec_t + (int64_t)pair * H * hp;
    float2 *orr = spec_r + (int64_t)pair * H * hp;

    float tn[16], rn[16];           // the next row, raw
    auto fetch = [&](int y) {
        const float *trow = tbase + (int64_t)y * tw_;
        const float *rrow = rbase + (int64_t)y * F1K;
        const bool trow_in = y < mh;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int xx = lane + 64 * m;
            tn[m] = (trow_in && xx < mw) ? trow[xx] : tc;
            rn[m] = rrow[xx];
        }
    };
    fetch(y_block + wave);
    for (int j = 0; j < F1K_ROWS_PER_WAVE; ++j) {
        const int y_group = y_block + 4 * j, y = y_group + wave;
        float2 v[16];
        // per-lane partial sums of 16 centred values fit float32 comfortably; lanes and
        // rows are combined in float64
        float fts = 0.f, ftq = 0.f, frs = 0.f, frq = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float tv = tn[m] - tc, rv = rn[m] - rc;
            v[m] = make_float2(tv, rv);
            fts += tv;
            ftq = fmaf(tv, tv, ftq);
            frs += rv;
            frq = fmaf(rv, rv, frq);
        }
        if (j + 1 < F1K_ROWS_PER_WAVE)
            fetch(y + 4);
        double ts = fts, tq = ftq, rs = frs, rq = frq;
        // template statistics cover the whole template: add what the crop to the
        // region's shape left out (columns beyond W, rows beyond H)
        for (int yy = y; yy < th; yy += H)
            for (int xx = (yy == y && y < mh ? mw : 0) + lane; xx < tw_; xx += 64) {
                const double d = (double)(tbase[(int64_t)yy * tw_ + xx] - tc);
                ts += d;
                tq += d * d;
            }
        ts = wave_sum(ts);
        tq = wave_sum(tq);
        rs = wave_sum(rs);
        rq = wave_sum(rq);
        if (lane == 0)                  // per-row partial sums; fft_row_sums_kernel adds them up
            reinterpret_cast<double4 *>(row_sums)[(int64_t)pair * H + y] = make_double4(ts, tq, rs, rq);
        f1k_forward(v, x, twiddle, lane);
        f1k_write_quads(v, x, lane);
        __syncthreads();
        // split z = fft(t + i r) into the two half spectra; item -> (column group, row, half)
        constexpr int GROUPS = F1K / 2 / 4 + 1;             // 129: columns 0 .. 512
        for (int item = threadIdx.x; item < GROUPS * 8; item += 256) {
            const int half = item & 1, w = (item >> 1) & 3, k0 = (item >> 3) * 4 + 2 * half;
            if (k0 > F1K / 2)
                continue;
            const float2 *xs = lds + w * F1K_LDS;
            float2 ta[2], ra[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = min(k0 + c, F1K / 2);
                const float2 zk = xs[f1k_idx(digit_reverse(k, 10))];
                const float2 zm = xs[f1k_idx(digit_reverse((F1K - k) & (F1K - 1), 10))];
                ta[c] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                ra[c] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
            }
            const int64_t at = f1k_spec_index(y_group + w, k0);
            if (k0 + 1 <= F1K / 2) {
                *reinterpret_cast<float4 *>(ot + at) = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
                *reinterpret_cast<float4 *>(orr + at) = make_float4(ra[0].x, ra[0].y, ra[1].x, ra[1].y);
            } else {
                ot[at] = ta[0];
                orr[at] = ra[0];
            }
        }
        __syncthreads();    // the next rows reuse the transform buffers
    }
}

// pass 2, H = 1024: a tile of F1K_TILE spectrum columns; wavefront c transforms
// column c of T (kept in registers), then column c of R, multiplies and transforms
// back.  The R tile is fetched into registers before the T transform starts, so its
// memory latency is covered by arithmetic.  A tile is one column group of the
// spectrum layout (f1k_spec_index): 32 KB of consecutive memory.  Workgroups of 4
// wavefronts measured faster than tiles of 8 columns in workgroups of 8: the phases
// of independent small workgroups on a CU overlap where one large workgroup waits
// at its barriers.
#ifndef SV_F1K_TILE
#define SV_F1K_TILE 4
#endif
#ifndef SV_F1K_MINWAVES
#define SV_F1K_MINWAVES 4
#endif
constexpr int F1K_TILE = SV_F1K_TILE;
static_assert(F1K_TILE == 4, "the spectrum layout groups columns in fours");
constexpr int F1K_COL = F1K_LDS + 2;              // column stride in LDS (bank spread for the transpose)
constexpr int F1K_QUADS = F1K * (F1K_TILE / 2) / (64 * F1K_TILE);   // 16-byte pieces per lane and tile

__device__ __forceinline__ void f1k_tile_fetch(const float2 *__restrict__ g, int hp, int c0, int ncols,
                                               float4 (&q)[F1K_QUADS])
{
#pragma unroll
    for (int n = 0; n < F1K_QUADS; ++n) {
        const int i = threadIdx.x + n * 64 * F1K_TILE;
        const int y = i / (F1K_TILE / 2), cp = (i % (F1K_TILE / 2)) * 2;
        q[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + cp + 1 < ncols) {
            q[n] = *reinterpret_cast<const float4 *>(g + f1k_spec_index(y, c0 + cp));
        } else if (c0 + cp < ncols) {
            const float2 a = g[f1k_spec_index(y, c0 + cp)];
            q[n].x = a.x;
            q[n].y = a.y;
        }
    }
}

// [y][8 columns] pieces -> [column][y] in LDS
__device__ __forceinline__ void f1k_tile_scatter(const float4 (&q)[F1K_QUADS], float2 *tile)
{
#pragma unroll
    for (int n = 0; n < F1K_QUADS; ++n) {
        const int i = threadIdx.x + n * 64 * F1K_TILE;
        const int y = i / (F1K_TILE / 2), cp = (i % (F1K_TILE / 2)) * 2;
        tile[cp * F1K_COL + f1k_idx(y)] = make_float2(q[n].x, q[n].y);
        tile[(cp + 1) * F1K_COL + f1k_idx(y)] = make_float2(q[n].z, q[n].w);
    }
}

// Workgroups loop over (pair, tile) work items and always have the next tile's
// memory reads in flight while they transform the current one (the transform of T
// stays in registers while R is transformed, and a tile prefetch is in flight:
// about 210 registers per lane, two wavefronts per SIMD).
__global__ void __launch_bounds__(64 * F1K_TILE, SV_F1K_MINWAVES)
f1k_cols_fused_kernel(float2 *__restrict__ spec_t, const float2 *__restrict__ spec_r, int ncols, int hp,
                      int tiles_per_pair, int total_tiles, const float2 *__restrict__ twiddle_table)
{
    __shared__ float2 tile[F1K_TILE * F1K_COL];
    __shared__ float2 twiddle[F1K_TW];              // LDS copy: gathered reads stay off the memory pipeline
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *x = tile + wave * F1K_COL;
    f1k_load_twiddles(twiddle, twiddle_table, threadIdx.x, 64 * F1K_TILE);
    const int first = blockIdx.x;

    float4 q[F1K_QUADS];
    int item = first;
    if (item < total_tiles) {
        const int pair = item / tiles_per_pair, c0 = (item % tiles_per_pair) * F1K_TILE;
        f1k_tile_fetch(spec_t + (int64_t)pair * F1K * hp, hp, c0, ncols, q);
    }
    for (; item < total_tiles; item += gridDim.x) {
        const int pair = item / tiles_per_pair, c0 = (item % tiles_per_pair) * F1K_TILE;
        float2 *gt = spec_t + (int64_t)pair * F1K * hp;
        const float2 *gr = spec_r + (int64_t)pair * F1K * hp;
        f1k_tile_scatter(q, tile);                     // T tile
        f1k_tile_fetch(gr, hp, c0, ncols, q);          // R tile: in flight during the T transform
        __syncthreads();
        float2 tv[16], v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m)
            tv[m] = x[f1k_idx(lane + 64 * m)];
        __builtin_amdgcn_wave_barrier();
        f1k_forward(tv, x, twiddle, lane);
        __syncthreads();
        f1k_tile_scatter(q, tile);                     // R tile
        const int next = item + gridDim.x;
        if (next < total_tiles) {                      // next T tile: in flight during the rest
            const int npair = next / tiles_per_pair, nc0 = (next % tiles_per_pair) * F1K_TILE;
            f1k_tile_fetch(spec_t + (int64_t)npair * F1K * hp, hp, nc0, ncols, q);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = x[f1k_idx(lane + 64 * m)];
        __builtin_amdgcn_wave_barrier();
        f1k_forward(v, x, twiddle, lane);
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = cmul_conj(v[m], tv[m]);             // conj(T) * R, position by position
        // the arrays were centred only approximately: removing their exact means is the
        // same as clearing the DC bin (position 0 of spectrum column 0)
        if (c0 + wave == 0 && lane == 0)
            v[0] = make_float2(0.f, 0.f);
        f1k_inverse(v, x, twiddle, lane);
#pragma unroll
        for (int m = 0; m < 16; ++m)
            x[f1k_idx(lane + 64 * m)] = v[m];
        __syncthreads();
#pragma unroll
        for (int n = 0; n < F1K_QUADS; ++n) {
            const int i = threadIdx.x + n * 64 * F1K_TILE;
            const int y = i / (F1K_TILE / 2), cp = (i % (F1K_TILE / 2)) * 2;
            const float2 a = tile[cp * F1K_COL + f1k_idx(y)], b = tile[(cp + 1) * F1K_COL + f1k_idx(y)];
            if (c0 + cp + 1 < ncols)
                *reinterpret_cast<float4 *>(gt + f1k_spec_index(y, c0 + cp)) = make_float4(a.x, a.y, b.x, b.y);
            else if (c0 + cp < ncols)
                gt[f1k_spec_index(y, c0 + cp)] = a;
        }
        __syncthreads();                               // LDS is rewritten by the next item
    }
}

// pass 3, W = 1024: one wavefront per two output rows, four wavefronts per workgroup
template <typename TOut>
__global__ void __launch_bounds__(256)
f1k_rows_inverse_kernel(const float2 *__restrict__ spec, const float *__restrict__ t, int th, int tw_,
                        const float *__restrict__ r, int H, int hp, const FftStats *__restrict__ stats,
                        double min_std, double min_mean_diff, const float2 *__restrict__ twiddle_table,
                        TOut *__restrict__ out)
{
    __shared__ float2 lds[4 * F1K_LDS];
    __shared__ float2 twiddle[F1K_TW];              // LDS copy: gathered reads stay off the memory pipeline
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y0 = 2 * (blockIdx.x * 4 + wave), pair = blockIdx.y;
    f1k_load_twiddles(twiddle, twiddle_table, threadIdx.x, 256);
    __syncthreads();
    if (y0 >= H)
        return;
    (void)t;
    (void)r;
    float2 *x = lds + wave * F1K_LDS;
    const FftStats st = stats[pair];
    const double tn = (double)th * tw_, rn = (double)H * F1K;
    const double tc = st.t_centre, rc = st.r_centre;
    double tmean, rmean, tvar, rvar;
    moments(st.t_sum, st.t_sq, tn, tc, tmean, tvar);
    moments(st.r_sum, st.r_sq, rn, rc, rmean, rvar);
    const double tstd = sqrt(fmax(tvar, 0.0));
    const double rstd = rvar > 1e-14 * (st.r_sq / rn + 1e-300) ? sqrt(rvar) : 0.0;
    TOut *o0 = out + ((int64_t)pair * H + y0) * F1K;
    TOut *o1 = o0 + F1K;
    const bool second = y0 + 1 < H;
    if (tstd < min_std || rstd == 0.0) {
        const TOut fill = tstd < min_std ? (fabs(tmean - rmean) < min_mean_diff ? TOut(1) : TOut(0))
                                         : TOut(NAN);
        for (int xx = lane; xx < F1K; xx += 64) {
            o0[xx] = fill;
            if (second)
                o1[xx] = fill;
        }
        return;
    }
    const float2 *p0 = spec + (int64_t)pair * H * hp + f1k_spec_index(y0, 0);
    const float2 *p1 = second ? p0 + 4 : p0;        // the next row of every column group
#pragma unroll
    for (int i = 0; i <= 8; ++i) {
        const int k = lane + 64 * i;
        if (k <= F1K / 2) {
            const float2 a = p0[f1k_spec_index(0, k)], b = p1[f1k_spec_index(0, k)];
            x[f1k_idx(digit_reverse(k, 10))] = make_float2(a.x - b.y, a.y + b.x);          // a + i b
            if (k > 0 && k < F1K / 2)       // Hermitian completion: conj(a) + i conj(b)
                x[f1k_idx(digit_reverse(F1K - k, 10))] = make_float2(a.x + b.y, b.x - a.y);
        }
    }
    __builtin_amdgcn_wave_barrier();
    float2 v[16];
    f1k_read_quads(v, x, lane);
    __builtin_amdgcn_wave_barrier();
    f1k_inverse(v, x, twiddle, lane);
    const float scale = (float)(1.0 / ((double)H * (double)F1K) / (tstd * tn) / rstd);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        o0[lane + 64 * m] = (TOut)(v[m].x * scale);
        if (second)
            o1[lane + 64 * m] = (TOut)(v[m].y * scale);
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
    const size_t hp = (size_t)W / 2 + 16;
    return align_up((size_t)batch * sizeof(FftStats), 256) + 2 * align_up(FFT_MAX_N * sizeof(float2), 256) +
           2 * align_up((size_t)batch * H * hp * sizeof(float2), 256) +
           align_up((size_t)batch * H * 4 * sizeof(double), 256);
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
    double *row_sums = ws.take<double>((size_t)batch * H * 4);

    SV_HIP(hipMemsetAsync(stats, 0, (size_t)batch * sizeof(FftStats), st));
    hipLaunchKernelGGL(twiddle_kernel, dim3((W + 255) / 256), dim3(256), 0, st, tw_w, W);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(twiddle_kernel, dim3((H + 255) / 256), dim3(256), 0, st, tw_h, H);
    SV_LAUNCH_CHECK();
    const bool fast = H == F1K && W == F1K && !raw;   // wavefront-per-transform kernels
    if (fast) {
        hipLaunchKernelGGL(fft_centres_kernel, dim3(batch), dim3(1024), 0, st, tmpl, th, tw, region, H, W, stats);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL(f1k_rows_forward_kernel, dim3((H + 4 * F1K_ROWS_PER_WAVE - 1) / (4 * F1K_ROWS_PER_WAVE), batch), dim3(256), 0, st, tmpl, th, tw, region, H,
                           stats, row_sums, tw_w, spec_t, spec_r, hp);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL(fft_row_sums_kernel, dim3(batch), dim3(1024), 0, st, row_sums, H, stats);
        SV_LAUNCH_CHECK();
        const int tiles_per_pair = (ncols + F1K_TILE - 1) / F1K_TILE;
        const int total_tiles = tiles_per_pair * batch;
#ifdef SV_F1K_GRID
        const int resident = SV_F1K_GRID;
#else
        const int resident = 256 * (SV_F1K_MINWAVES * 4 / F1K_TILE);   // workgroups the chip holds at once
#endif
        hipLaunchKernelGGL(f1k_cols_fused_kernel, dim3(total_tiles < resident ? total_tiles : resident),
                           dim3(64 * F1K_TILE), 0, st, spec_t, spec_r, ncols, hp, tiles_per_pair, total_tiles,
                           tw_h);
        SV_LAUNCH_CHECK();
        dim3 gf((H / 2 + 3) / 4, batch);
        if (out_dtype == SVAMD_F32)
            hipLaunchKernelGGL((f1k_rows_inverse_kernel<float>), gf, dim3(256), 0, st, spec_t, tmpl, th, tw,
                               region, H, hp, stats, min_std, min_mean_diff, tw_w, static_cast<float *>(out));
        else
            hipLaunchKernelGGL((f1k_rows_inverse_kernel<double>), gf, dim3(256), 0, st, spec_t, tmpl, th, tw,
                               region, H, hp, stats, min_std, min_mean_diff, tw_w, static_cast<double *>(out));
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
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
