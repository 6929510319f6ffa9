// Sliding-dot-product cross-correlation: the spatial 2D NCC (a strided, NaN
// masked, edge-truncated template scan), the large-shape circular 2D NCC and the
// two 1D correlators all reduce to
//     raw[i][j] = sum_{m,n} t'[m][n] * ext[i*sy + m][j*sx + n]
// over a mean-removed float32 template t' and a float32 image `ext` that has been
// extended (zero padded or wrapped circularly) so that no index leaves it.  The
// normalisation is an O(1)-per-output epilogue in float64 fed by summed-area
// tables (spatial) or global statistics (circular / 1D).
#include <mutex>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svamd {

inline int grid_for(int64_t total, int block = 256, int cap = 65536)
{
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- statistics ---------------------------------------------------------------
// One block; two passes (mean, then squared deviations), all in float64, the
// same two-pass definition numpy's mean/std and nanmean/nanstd use.
// Element i of an h x w array with row pitch `pitch` (no division when rows are contiguous).
__device__ __forceinline__ int64_t stats_offset(int64_t i, int w, int64_t pitch)
{
    if (pitch == w)
        return i;
    const int64_t y = i / w;
    return y * pitch + (i - y * w);
}

// Sum and count of a thread's elements first, first + stride, ... < total (of their
// squared deviations from `mean` when SQUARES), eight loads in flight at a time; a
// thread adds its elements in index order.
template <bool SQUARES>
__device__ __forceinline__ void stats_accumulate(const void *__restrict__ src, int dtype, int w, int64_t pitch,
                                                 int ignore_nan, int64_t first, int64_t stride, int64_t total,
                                                 double mean, double &s, double &c)
{
    for (int64_t base = first; base < total; base += 8 * stride) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = base + k * stride;
            v[k] = i < total ? load_as_double(src, dtype, stats_offset(i, w, pitch)) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool use = base + k * stride < total && (!ignore_nan || !isnan(v[k]));
            if (use) {
                const double d = v[k] - mean;
                s += SQUARES ? d * d : v[k];
                c += 1.0;
            }
        }
    }
}

__global__ void __launch_bounds__(1024)
stats_kernel(const void *__restrict__ src, int dtype, int h, int w, int64_t pitch, int ignore_nan,
             Stats *__restrict__ out)
{
    __shared__ double red[17];
    const int64_t total = (int64_t)h * w;
    double s = 0.0, c = 0.0;
    stats_accumulate<false>(src, dtype, w, pitch, ignore_nan, threadIdx.x, blockDim.x, total, 0.0, s, c);
    s = block_sum(s, red);
    c = block_sum(c, red);
    const double mean = s / c;
    double q = 0.0, c2 = 0.0;
    stats_accumulate<true>(src, dtype, w, pitch, ignore_nan, threadIdx.x, blockDim.x, total, mean, q, c2);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        out->mean = mean;
        out->std = sqrt(q / c);
        out->count = c;
        out->pad = 0.0;
    }
}

// Large arrays: the same two passes spread over many blocks.  Pass 0 leaves
// per-block (sum, count) in `partial`, pass 1 per-block sums of squared deviations
// from the mean already in `out`; stats_final_kernel folds the partials.
constexpr int STATS_BLOCKS = 512;
constexpr int64_t STATS_SINGLE_BLOCK_MAX = 1 << 16;

__global__ void __launch_bounds__(256)
stats_partial_kernel(const void *__restrict__ src, int dtype, int h, int w, int64_t pitch, int ignore_nan,
                     int pass, const Stats *__restrict__ st, double *__restrict__ partial)
{
    __shared__ double red[17];
    const int64_t total = (int64_t)h * w;
    const double mean = pass ? st->mean : 0.0;
    double s = 0.0, c = 0.0;
    const int64_t first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (pass)
        stats_accumulate<true>(src, dtype, w, pitch, ignore_nan, first, stride, total, mean, s, c);
    else
        stats_accumulate<false>(src, dtype, w, pitch, ignore_nan, first, stride, total, 0.0, s, c);
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = c;
    }
}

__global__ void __launch_bounds__(256)
stats_final_kernel(const double *__restrict__ partial, int nblocks, int pass, Stats *__restrict__ out)
{
    __shared__ double red[17];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        s += partial[2 * i];
        c += partial[2 * i + 1];
    }
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        if (pass == 0) {
            out->mean = s / c;
            out->count = c;
            out->pad = 0.0;
        } else {
            out->std = sqrt(s / c);
        }
    }
}

size_t stats_scratch_doubles() { return 2 * STATS_BLOCKS; }

int stats_launch(const void *src, int dtype, int h, int w, int64_t pitch, bool ignore_nan, Stats *out,
                 double *scratch, hipStream_t stream)
{
    const int64_t total = (int64_t)h * w;
    if (total <= STATS_SINGLE_BLOCK_MAX) {
        hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, stream, src, dtype, h, w, pitch,
                           ignore_nan ? 1 : 0, out);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    int64_t nb = (total + 256 * 16 - 1) / (256 * 16);
    if (nb > STATS_BLOCKS)
        nb = STATS_BLOCKS;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(stats_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, src, dtype, h, w,
                           pitch, ignore_nan ? 1 : 0, pass, out, scratch);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, stream, scratch, (int)nb, pass, out);
        SV_LAUNCH_CHECK();
    }
    return SVAMD_OK;
}

// The constant subtracted from region pixels before they are rounded to
// float32: the region mean, rounded to an integer for uint8 data so that the
// centred pixels stay exactly representable.
__device__ __forceinline__ double centre_value(const Stats *st, int dtype)
{
    return dtype == SVAMD_U8 ? rint(st->mean) : st->mean;
}

// ---- operand preparation ----------------------------------------------------------
// Mean-removed template in float32 (masked pixels -> 0).
__global__ void __launch_bounds__(256)
prep_template_kernel(const void *__restrict__ t, int dtype, int th, int tw, int64_t pitch, int oh, int ow,
                     const Stats *__restrict__ st, float *__restrict__ tp)
{
    const double mean = st->mean;
    const int64_t total = (int64_t)oh * ow;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = i / ow, x = i - y * ow;
        double v = load_as_double(t, dtype, y * pitch + x);
        tp[i] = isnan(v) ? 0.0f : (float)(v - mean);
    }
}

// Extended float32 image.  mode 0: region - c inside, 0 outside (zero padding of
// the centred data).  mode 1: circular wrap of region - c.
__global__ void __launch_bounds__(256)
extend_kernel(const void *__restrict__ src, int dtype, int rh, int rw, int64_t pitch, int mode,
              int exact_mean, const Stats *__restrict__ st, int eh, int ew, int64_t epitch,
              float *__restrict__ ext, const int64_t *__restrict__ table_total)
{
    // table_total (uint8 regions): the last entry of the region's exact integer table of
    // sums; the rounded mean taken from it needs neither `st` nor a pass over the region
    const double c = table_total ? rint((double)*table_total / ((double)rh * (double)rw))
                                 : exact_mean ? st->mean : centre_value(st, dtype);
    const int64_t total = (int64_t)eh * epitch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = i / epitch, x = i - y * epitch;
        float v = 0.0f;
        if (x < ew) {
            if (mode == 0) {
                if (y < rh && x < rw)
                    v = (float)(load_as_double(src, dtype, y * pitch + x) - c);
            } else {
                v = (float)(load_as_double(src, dtype, (y % rh) * pitch + (x % rw)) - c);
            }
        }
        ext[i] = v;
    }
}

// 1D: ext[k] = (v2 zero-padded by n-1 on both sides)[k] - mean(v2)
__global__ void __launch_bounds__(256)
extend_1d_full_kernel(const double *__restrict__ v2, int64_t n, const Stats *__restrict__ st,
                      float *__restrict__ ext)
{
    const double m2 = st->mean;
    const int64_t total = 3 * n - 2;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = k - (n - 1);
        double v = (s >= 0 && s < n) ? v2[s] : 0.0;
        ext[k] = (float)(v - m2);
    }
}

// 1D circular: ext[k] = padded[k mod N], padded = zeros(nt-1) ++ signal (raw)
__global__ void __launch_bounds__(256)
extend_1d_circular_kernel(const double *__restrict__ sig, int64_t nt, int64_t ns, float *__restrict__ ext)
{
    const int64_t N = ns + nt - 1, total = N + nt - 1;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        int64_t idx = k >= N ? k - N : k;
        ext[k] = idx >= nt - 1 ? (float)sig[idx - (nt - 1)] : 0.0f;
    }
}

// 1D circular, roles swapped (signal shorter than template): the mean-removed
// template, zero-padded to N = ns + nt - 1 and wrapped: ext[i] = tN[i mod N].
__global__ void __launch_bounds__(256)
extend_1d_template_kernel(const double *__restrict__ t, int64_t nt, int64_t N, int64_t total,
                          const Stats *__restrict__ st, float *__restrict__ ext)
{
    const double mean = st->mean;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = i >= N ? i - N : i;
        ext[i] = idx < nt ? (float)(t[idx] - mean) : 0.0f;
    }
}

__global__ void __launch_bounds__(256)
to_float_kernel(const double *__restrict__ src, int64_t n, float *__restrict__ dst)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

// out[k] = raw[(nt - 1 - k) mod N] / (std * nt)
__global__ void __launch_bounds__(256)
scale_1d_swapped_kernel(const double *__restrict__ raw, int64_t N, int64_t nt, const Stats *__restrict__ s1,
                        double *__restrict__ out)
{
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= N)
        return;
    int64_t u = nt - 1 - k;
    if (u < 0)
        u += N;
    out[k] = raw[u] / (double)nt / s1->std;
}

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// ---- the sliding dot product --------------------------------------------------------
// Strided columns (sx > 1).  Outputs sit sx image columns apart, so neighbouring
// outputs share no aligned image data the way the unit-stride kernel exploits;
// what they do share is image ROWS: image row r serves output row i as template
// row r - i * sy.  A lane owns one output column and K consecutive output rows
// (K = 8, 4, 2 or 1: the largest for which the rows of a group mostly overlap),
// and every image value it reads feeds K FMAs against K wave-uniform template
// values (scalar loads from a template padded with (K - 1) * sy zero rows above
// and below, so that no row test is needed).
//
// The template's columns are cut into pieces of P = 4 * N4 columns (N4 = 1..3).
// Per piece the image rows a group touches are walked in steps of 4; the (piece,
// step) pairs of a tile of 64 output columns x K output rows form one list that
// is dealt out in equal runs to the 4 wavefronts of a workgroup and to the
// workgroups along the grid's z dimension (a 15 000-output scan would otherwise
// occupy a fraction of the chip).  Staging is private to the wavefront (no
// barriers): the part of row r that lane l needs,
// img[r][(j0 + l) * sx + n0 ... + P), is loaded with neighbouring lanes on
// neighbouring columns and stored as lane l's own segment; a lane then reads its
// segment 16 bytes at a time.  Segment pitches of 4 * odd floats keep those reads
// free of bank conflicts.  The loads of the next step are in flight while the
// current one is computed, and so are the template values and the segment reads
// of the next staged row (K * N4 <= 8 keeps two rows' template values in SGPRs).  Float32 sums are folded into float64
// every SS_FLUSH products.  The four wavefronts add their sums through LDS; the
// z slices go to partial planes that slide_reduce_kernel adds.
constexpr int SS_WAVES = 4;       // wavefronts per workgroup
constexpr int SS_ROWS = 4;        // image rows staged per step
constexpr int SS_N4_MAX = 3;      // 16-byte groups per piece at most
constexpr int SS_FLUSH = 1024;
constexpr int SS_MAX_SLICES = 32;

struct StridedGeom {
    int th, tw, oh, ow, sx, sy;
    int eh, ew;                   // valid image rows / columns
    int64_t pitch;
    int lp, padrows;              // padded template: row length (pieces * P), zero rows above (below: SS_ROWS more)
    int npieces, iters_per_wave;
};

template <int K, int N4>
__global__ void __launch_bounds__(SS_WAVES *WAVE)
slide_strided_kernel(StridedGeom g, const float *__restrict__ tpad, const float *__restrict__ img,
                     double *__restrict__ out)
{
    constexpr int P = 4 * N4;                 // template columns per piece
    constexpr int SEG = 4 * (N4 | 1);         // LDS floats per lane and staged row
    constexpr int DQ = WAVE / P, DR = WAVE % P;
    extern __shared__ __align__(16) float ss_lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int j0 = blockIdx.x * WAVE, i0 = blockIdx.y * K;
    const int r_first = i0 * g.sy;
    const int r_end = min((i0 + K - 1) * g.sy + g.th, g.eh);      // exclusive
    const int nsteps = (r_end - r_first + SS_ROWS - 1) / SS_ROWS;
    const int total = g.npieces * nsteps;
    const int it_begin = min((int)(blockIdx.z * SS_WAVES + wave) * g.iters_per_wave, total);
    const int it_end = min(it_begin + g.iters_per_wave, total);
    float *wl = ss_lds + wave * (SS_ROWS * WAVE * SEG);
    const int64_t krow = (int64_t)g.sy * g.lp;        // template floats between two output rows of a lane

    float acc[K];
    double dacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        acc[k] = 0.f;
        dacc[k] = 0.0;
    }
    // element e of a staged row = (lane segment e / P, column e % P); a lane takes
    // e = lane, lane + 64, ...
    const int l_first = lane / P, n_first = lane % P;
    // image column (relative to the piece's first template column) and LDS slot of the
    // lane's u-th element: the same for every row, piece and step
    int col_of[P], slot_of[P];
    {
        int l = l_first, n = n_first;
#pragma unroll
        for (int u = 0; u < P; ++u) {
            col_of[u] = (j0 + l) * g.sx + n;
            slot_of[u] = l * SEG + n;
            l += DQ;
            n += DR;
            if (n >= P) {
                n -= P;
                ++l;
            }
        }
    }
    float stage[SS_ROWS][P];
    auto fetch = [&](int piece, int step) {
        const int n0 = piece * P, r0 = r_first + step * SS_ROWS;
        const int room = g.ew - n0;               // columns left of the image's right edge
#pragma unroll
        for (int rr = 0; rr < SS_ROWS; ++rr) {
            const int r = r0 + rr;
            const bool row_ok = r < r_end;
            const float *src = img + (int64_t)r * g.pitch + n0;
#pragma unroll
            for (int u = 0; u < P; ++u)
                stage[rr][u] = (row_ok && col_of[u] < room) ? src[col_of[u]] : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int rr = 0; rr < SS_ROWS; ++rr) {
            float *dst = wl + rr * (WAVE * SEG);
#pragma unroll
            for (int u = 0; u < P; ++u)
                dst[slot_of[u]] = stage[rr][u];
        }
    };

    int pending = 0;
    int piece = it_begin / nsteps, step = it_begin - piece * nsteps;
    if (it_begin < it_end)
        fetch(piece, step);
    const float *lseg = wl + lane * SEG;
    for (int it = it_begin; it < it_end; ++it) {
        store();
        int npiece = piece, nstep = step + 1;
        if (nstep == nsteps) {
            nstep = 0;
            ++npiece;
        }
        if (it + 1 < it_end)
            fetch(npiece, nstep);
        // ---- K output rows against each staged row ----
        // template row of output row i0 (k = 0) for the step's first image row; output
        // row i0 + k uses the row k * sy above.  Rows staged beyond r_end are zeros and
        // meet the SS_ROWS zero rows below the padded template.
        const float *t0 = tpad + (int64_t)(step * SS_ROWS + g.padrows) * g.lp + piece * P;
        // one unit = one staged row: its N4 segment reads and K * N4 template quads,
        // requested one row ahead of their use (two register sets take turns)
        f4v tv[2][K][N4], iv[2][N4];
        auto load_row = [&](int rr, f4v (&t)[K][N4], f4v (&v)[N4]) {
            const float *tp = t0 + (int64_t)rr * g.lp;
#pragma unroll
            for (int q = 0; q < N4; ++q)
                v[q] = *reinterpret_cast<const f4v *>(lseg + rr * (WAVE * SEG) + 4 * q);
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int q = 0; q < N4; ++q)
                    t[k][q] = *reinterpret_cast<const f4v *>(tp - k * krow + 4 * q);
        };
        load_row(0, tv[0], iv[0]);
#pragma unroll
        for (int rr = 0; rr < SS_ROWS; ++rr) {
            // this row's values have had the previous row's arithmetic to arrive; waiting
            // for them here, before the next row's requests go out, keeps that wait from
            // also covering requests made a moment ago (scalar loads return out of
            // order, so any later wait would be for everything outstanding)
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0), the other counters left alone
            __builtin_amdgcn_sched_barrier(0);
            if (rr + 1 < SS_ROWS)
                load_row(rr + 1, tv[(rr + 1) & 1], iv[(rr + 1) & 1]);
#pragma unroll
            for (int q = 0; q < N4; ++q) {
                const f4v v = iv[rr & 1][q];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const f4v t = tv[rr & 1][k][q];
                    acc[k] = fmaf(t.x, v.x, acc[k]);
                    acc[k] = fmaf(t.y, v.y, acc[k]);
                    acc[k] = fmaf(t.z, v.z, acc[k]);
                    acc[k] = fmaf(t.w, v.w, acc[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pending += SS_ROWS * P;
        if (pending >= SS_FLUSH) {
            pending = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                dacc[k] += (double)acc[k];
                acc[k] = 0.f;
            }
        }
        piece = npiece;
        step = nstep;
    }
    // ---- the four wavefronts' sums, added in wavefront order ----
    __syncthreads();
    double *red = reinterpret_cast<double *>(ss_lds);
#pragma unroll
    for (int k = 0; k < K; ++k)
        red[(wave * K + k) * WAVE + lane] = dacc[k] + (double)acc[k];
    __syncthreads();
    if (wave != 0)
        return;
    const int j = j0 + lane;
    if (j >= g.ow)
        return;
    double *dst = out + (int64_t)blockIdx.z * g.oh * g.ow;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (i0 + k >= g.oh)
            break;
        double sum = red[k * WAVE + lane];
#pragma unroll
        for (int w = 1; w < SS_WAVES; ++w)
            sum += red[(w * K + k) * WAVE + lane];
        dst[(int64_t)(i0 + k) * g.ow + j] = sum;
    }
}

// Template with `padrows` zero rows above, `padrows` + SS_ROWS below, rows of lp floats.
__global__ void __launch_bounds__(256)
pad_template_kernel(const float *__restrict__ tp, int th, int tw, int lp, int padrows, float *__restrict__ tpad)
{
    const int64_t total = (int64_t)(th + 2 * padrows + SS_ROWS) * lp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % lp);
        const int m = (int)(i / lp) - padrows;
        tpad[i] = (m >= 0 && m < th && n < tw) ? tp[(int64_t)m * tw + n] : 0.0f;
    }
}

// Unit column stride (sx == 1): register-tiled, LDS-staged kernel.
//
// A wavefront owns one output row and 512 output columns, 8 consecutive columns
// per lane.  The products of one template row are issued as packed FMAs over
// PAIRS of template columns: for an even output column j
//     out[j] += (T0[2k], T0[2k+1]) . (img[j+2k], img[j+2k+1]),        T0 = t
// and for the odd column j+1, with the template shifted right by one,
//     out[j+1] += (T1[2k], T1[2k+1]) . (img[j+2k], img[j+2k+1]),      T1 = (0, t)
// so both use the same aligned image pairs: a lane keeps a sliding window of 7
// pairs in registers and adds 4 new pairs (two 16-byte LDS reads) per 32 packed
// FMAs.  The template pairs are wave-uniform and arrive through the scalar data
// path.  The two halves of each accumulator pair are added at the end.
//
// A workgroup is up to 4 wavefronts = 4 consecutive output rows of the same
// column tile and slice of template-column chunks (512 template columns each; a
// workgroup sums the chunks of its slice itself, slide_reduce_kernel adds the
// slices).  It walks down
// the image rows its outputs touch, staging up to 4 rows at a time in LDS
// (double-buffered: the next rows are fetched into registers while the current
// ones are used); image row r serves output row i as template row r - i * sy.
// LDS rows are stored as they are: the 32-byte lane stride of the window reads costs
// two-way bank conflicts, but then a block's address is the previous one plus a
// constant (immediate offsets, one address register) - a conflict-free layout with
// swapped 16-byte slots cost 16 address instructions per 64 packed FMAs and was slower
// (the LDS is busy a quarter of the time at most).
constexpr int ST_TILE = 512;     // output columns per wavefront
constexpr int ST_CHUNK = 512;    // template columns per workgroup and chunk
// ... and for one-wavefront workgroups (a single output row: the 1D calls), so that a
// staged row of tile + chunk + 8 floats is 4 pieces for each of the 64 lanes
constexpr int ST_CHUNK_ONE_ROW = 504;
constexpr int ST_WAVES = 4;      // output rows per workgroup
constexpr int ST_ROWS = 4;       // image rows staged per step
// 16-byte pieces a thread stages per step: 4 keep the kernel at 120 registers (four
// wavefronts per SIMD); 6 (149 registers, three wavefronts) only where 4 pieces per
// thread cannot hold a single staged row (one-wavefront workgroups of the 1D calls)
constexpr int ST_STAGE_Q = 4, ST_STAGE_Q_WIDE = 6;
constexpr int ST_FLUSH = 1024;   // products accumulated in float32 before they are folded into float64

struct SlideGeom {
    int th, tw, lp;        // template shape; lp = packed template row length (multiple of 8, > tw)
    int oh, ow, sy;
    int eh, ew;            // valid image rows / columns (anything beyond reads as 0)
    int64_t pitch;
    int waves, grows, rowlen;
    int chunk_cols;                  // template columns per chunk
    int nchunks, chunks_per_slice;   // template-column chunks, and how many of them a workgroup sums itself
    int row_slices, rows_per_slice;  // template rows are cut into this many runs, one per grid slice
};

// Packed template: for every row m, T0 = (t[m][0..tw-1], 0...) and T1 = (0, t[m][0..tw-1], 0...),
// lp floats each.
__global__ void __launch_bounds__(256)
pack_template_kernel(const float *__restrict__ tp, int th, int tw, int lp, float *__restrict__ tpk)
{
    const int64_t total = (int64_t)th * 2 * lp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % lp);
        const int which = (int)((i / lp) & 1);
        const int64_t m = i / (2 * (int64_t)lp);
        const int src = n - which;
        tpk[i] = (src >= 0 && src < tw) ? tp[m * tw + src] : 0.0f;
    }
}

// The 8 floats at logical columns 8b .. 8b+7 of a staged row, as 4 pairs.
__device__ __forceinline__ void slide_load_block(const float *row, int b, f2v (&w)[4])
{
    const int c = 8 * b;
    const f4v lo = *reinterpret_cast<const f4v *>(row + c);
    const f4v hi = *reinterpret_cast<const f4v *>(row + c + 4);
    w[0] = lo.xy;
    w[1] = lo.zw;
    w[2] = hi.xy;
    w[3] = hi.zw;
}

__device__ __forceinline__ void slide_group(f2v (&ae)[4], f2v (&ao)[4], const f2v (&cur)[4],
                                            const f2v (&nxt)[4], const float *__restrict__ t0,
                                            const float *__restrict__ t1)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f2v a = f2v{t0[2 * s], t0[2 * s + 1]};
        const f2v b = f2v{t1[2 * s], t1[2 * s + 1]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f2v w = e + s < 4 ? cur[e + s] : nxt[e + s - 4];
            ae[e] = __builtin_elementwise_fma(a, w, ae[e]);
            ao[e] = __builtin_elementwise_fma(b, w, ao[e]);
        }
    }
}

template <int STAGE_Q>
__global__ void __launch_bounds__(ST_WAVES *WAVE)
slide_tiled_kernel(SlideGeom g, const float *__restrict__ tpk, const float *__restrict__ img,
                   double *__restrict__ out)
{
    extern __shared__ __align__(16) float st_lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int nthreads = g.waves * WAVE;
    const int c0 = blockIdx.x * ST_TILE;
    const int i0 = blockIdx.y * g.waves;
    const int i = i0 + wave;
    const bool has_row = i < g.oh;
    // grid slice -> (run of template-column chunks, run of template rows)
    const int zr = blockIdx.z % g.row_slices, zc = blockIdx.z / g.row_slices;
    const int mb = zr * g.rows_per_slice, me = min(mb + g.rows_per_slice, g.th);
    const int r_first = i0 * g.sy + mb;
    const int r_end = (min(i0 + g.waves, g.oh) - 1) * g.sy + me;   // exclusive
    const int nsteps = (r_end - r_first + g.grows - 1) / g.grows;
    const int qpr = g.rowlen / 4, total_q = g.grows * qpr;

    int n0 = 0;                      // first template column of the chunk in hand
    f4v stage[STAGE_Q];
    auto fetch = [&](int step) {
#pragma unroll
        for (int u = 0; u < STAGE_Q; ++u) {
            const int q = threadIdx.x + u * nthreads;
            f4v v = {0.f, 0.f, 0.f, 0.f};
            if (q < total_q) {
                const int rr = q / qpr, cq = q - rr * qpr;
                const int r = r_first + step * g.grows + rr;
                const int col = c0 + n0 + 4 * cq;
                if (r < r_end && r < g.eh && col < g.ew) {
                    v = *reinterpret_cast<const f4v *>(img + (int64_t)r * g.pitch + col);
                    if (col + 1 >= g.ew) v.y = 0.f;
                    if (col + 2 >= g.ew) v.z = 0.f;
                    if (col + 3 >= g.ew) v.w = 0.f;
                }
            }
            stage[u] = v;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < STAGE_Q; ++u) {
            const int q = threadIdx.x + u * nthreads;
            if (q < total_q) {
                const int rr = q / qpr, cq = q - rr * qpr;
                *reinterpret_cast<f4v *>(st_lds + (buf * g.grows + rr) * g.rowlen + 4 * cq) = stage[u];
            }
        }
    };

    f2v ae[4] = {}, ao[4] = {};
    double de[4] = {}, dodd[4] = {};
    int pending = 0;

    const int ch_begin = zc * g.chunks_per_slice;
    const int ch_end = min(ch_begin + g.chunks_per_slice, g.nchunks);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
    n0 = ch * g.chunk_cols;
    const int chunk = min(g.chunk_cols, g.lp - n0);
    const int groups = chunk / 8;
    fetch(0);
    store(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps)
            fetch(step + 1);
        const float *buf = st_lds + (step & 1) * g.grows * g.rowlen;
        for (int rr = 0; rr < g.grows; ++rr) {
            const int r = r_first + step * g.grows + rr;
            const int m = r - i * g.sy;
            if (!has_row || r >= r_end || m < mb || m >= me)
                continue;                                     // wave-uniform
            const float *row = buf + rr * g.rowlen;
            const float *t0 = tpk + ((int64_t)m * 2) * g.lp + n0;
            const float *t1 = t0 + g.lp;
            f2v cur[4], nxt[4];
            slide_load_block(row, lane, cur);
            int gi = 0;
            for (; gi + 2 <= groups; gi += 2) {
                slide_load_block(row, lane + gi + 1, nxt);
                slide_group(ae, ao, cur, nxt, t0 + 8 * gi, t1 + 8 * gi);
                __builtin_amdgcn_sched_barrier(0);   // the reload of `cur` stays behind its last use: no register copies
                slide_load_block(row, lane + gi + 2, cur);
                slide_group(ae, ao, nxt, cur, t0 + 8 * gi + 8, t1 + 8 * gi + 8);
            }
            if (gi < groups) {
                slide_load_block(row, lane + gi + 1, nxt);
                slide_group(ae, ao, cur, nxt, t0 + 8 * gi, t1 + 8 * gi);
            }
            pending += chunk;
            if (pending >= ST_FLUSH) {
                pending = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    de[e] += (double)(ae[e].x + ae[e].y);
                    dodd[e] += (double)(ao[e].x + ao[e].y);
                    ae[e] = f2v{0.f, 0.f};
                    ao[e] = f2v{0.f, 0.f};
                }
            }
        }
        if (step + 1 < nsteps)
            store((step + 1) & 1);
        __syncthreads();
    }
    }
    if (!has_row)
        return;
    double *dst = out + ((int64_t)blockIdx.z * g.oh + i) * g.ow;
    const int j0 = c0 + 8 * lane;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (j0 + 2 * e < g.ow)
            dst[j0 + 2 * e] = de[e] + (double)(ae[e].x + ae[e].y);
        if (j0 + 2 * e + 1 < g.ow)
            dst[j0 + 2 * e + 1] = dodd[e] + (double)(ao[e].x + ao[e].y);
    }
}

__global__ void __launch_bounds__(256)
slide_reduce_kernel(const double *__restrict__ partial, int nz, int64_t total, double *__restrict__ out)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    double s = 0.0;
    for (int z = 0; z < nz; ++z)
        s += partial[(int64_t)z * total + idx];
    out[idx] = s;
}

static int slide_lp(int tw) { return (int)align_up((size_t)tw + 1, 8); }
static int slide_chunk_cols(int64_t oh) { return oh == 1 ? ST_CHUNK_ONE_ROW : ST_CHUNK; }
static int slide_chunks(int tw, int64_t oh)
{
    const int cols = slide_chunk_cols(oh);
    return (slide_lp(tw) + cols - 1) / cols;
}
constexpr int ST_MAX_SLICES = 16;   // partial-sum planes at most (workgroups sum the chunks of a slice themselves)

// padded template row of the strided kernel: whole pieces (upper bound)
static int strided_lp(int tw) { return (int)align_up((size_t)tw, 4) + 4 * SS_N4_MAX; }

// Output rows per lane of the strided kernel: the largest K whose rows mostly
// share their image rows ((K - 1) * sy <= th / 2, which also bounds the padded
// template by twice the template).
static int strided_rows_per_lane(int th, int sy)
{
    for (int k = 8; k > 1; k >>= 1)
        if ((int64_t)(k - 1) * sy <= th / 2)
            return k;
    return 1;
}

// Operand buffer shared by both kernels: the unit-stride kernel's packed template
// (two shifted copies of every row) or the strided kernel's padded template.
size_t slide_pack_floats(int th, int tw)
{
    const size_t lp = (size_t)(slide_lp(tw) > strided_lp(tw) ? slide_lp(tw) : strided_lp(tw));
    return ((size_t)th * 2 + SS_ROWS) * lp;
}

// Grid slices of the unit-stride kernel: runs of template-column chunks (wide
// templates) times runs of template rows (outputs too few to fill the chip on their
// own), ST_MAX_SLICES partial-sum planes at most.
static int slide_row_slices(int64_t oh, int64_t ow, int column_slices)
{
    const int64_t waves = oh < ST_WAVES ? oh : ST_WAVES;
    const int64_t tiles = ((ow + ST_TILE - 1) / ST_TILE) * ((oh + waves - 1) / waves) * column_slices;
    int64_t want = (1024 + tiles - 1) / tiles;
    if (want > ST_MAX_SLICES / column_slices) want = ST_MAX_SLICES / column_slices;
    return (int)(want < 1 ? 1 : want);
}

size_t slide_partial_doubles(int tw, int64_t oh, int64_t ow)
{
    const int nchunks = slide_chunks(tw, oh);
    const int column_planes = nchunks < ST_MAX_SLICES ? nchunks : ST_MAX_SLICES;
    const int planes = column_planes * slide_row_slices(oh, ow, column_planes);
    return planes > 1 ? (size_t)planes * (size_t)oh * (size_t)ow : 0;
}

// Partial planes of the strided kernel: a double per region pixel bounds the number
// of z slices a strided scan may use (at least step_x * step_y of them fit).
size_t slide_strided_partial_doubles(int64_t rh, int64_t rw) { return (size_t)(rh * rw + 256); }

template <int K>
static void strided_dispatch(int n4, dim3 grid, size_t lds, hipStream_t stream, const StridedGeom &g,
                             const float *pack, const float *img, double *dst)
{
    // K * N4 <= 8 (strided_piece_groups keeps to it)
    if constexpr (K <= 2) {
        if (n4 == 3) {
            hipLaunchKernelGGL((slide_strided_kernel<K, 3>), grid, dim3(SS_WAVES * WAVE), lds, stream, g, pack, img, dst);
            return;
        }
    }
    if constexpr (K <= 4) {
        if (n4 == 2) {
            hipLaunchKernelGGL((slide_strided_kernel<K, 2>), grid, dim3(SS_WAVES * WAVE), lds, stream, g, pack, img, dst);
            return;
        }
    }
    hipLaunchKernelGGL((slide_strided_kernel<K, 1>), grid, dim3(SS_WAVES * WAVE), lds, stream, g, pack, img, dst);
}

// 16-byte groups per piece: among the widths whose template values for two rows fit
// the scalar registers (K * N4 <= 8), the one that pads the template's columns
// least, the wider on ties.
static int strided_piece_groups(int K, int tw)
{
    const int most = K >= 8 ? 1 : K == 4 ? 2 : SS_N4_MAX;
    int n4 = 1;
    for (int c = 2; c <= most; ++c)
        if ((tw + 4 * c - 1) / (4 * c) * c <= (tw + 4 * n4 - 1) / (4 * n4) * n4)
            n4 = c;
    return n4;
}

static int slide_strided_launch(const float *t, int th, int tw, const float *img, int64_t img_pitch, int eh,
                                int ew, int oh, int ow, int sx, int sy, double *out, float *pack,
                                double *partial, size_t partial_cap, hipStream_t stream, int *planes_left)
{
    StridedGeom g;
    g.th = th; g.tw = tw; g.oh = oh; g.ow = ow; g.sx = sx; g.sy = sy; g.eh = eh; g.ew = ew;
    g.pitch = img_pitch;
    const int K = strided_rows_per_lane(th, sy);
    g.padrows = (K - 1) * sy;
    const int64_t gx = ((int64_t)ow + WAVE - 1) / WAVE, gy = ((int64_t)oh + K - 1) / K;
    SV_REQUIRE(gy <= 65535, SVAMD_ERR_LIMIT, "slide: too many output rows (%d)", oh);
    const int n4 = strided_piece_groups(K, tw);
    const int P = 4 * n4;
    g.npieces = (tw + P - 1) / P;
    g.lp = g.npieces * P;
    // (piece, step) pairs of a full tile, dealt out to enough wavefronts for the chip;
    // no more partial planes than the workspace holds
    const int64_t steps = (((int64_t)(K - 1) * sy + th) + SS_ROWS - 1) / SS_ROWS;
    const int64_t total = g.npieces * steps;
    int64_t want = (4096 + gx * gy * SS_WAVES - 1) / (gx * gy * SS_WAVES);
    const int64_t plane = (int64_t)oh * ow;
    if (want > (int64_t)(partial_cap / (size_t)plane)) want = (int64_t)(partial_cap / (size_t)plane);
    if (want > SS_MAX_SLICES) want = SS_MAX_SLICES;
    if (want < 1) want = 1;
    int64_t ipw = (total + want * SS_WAVES - 1) / (want * SS_WAVES);
    if (ipw < 4) ipw = total < 4 ? total : 4;     // staging start-up is not worth fewer
    g.iters_per_wave = (int)ipw;
    const int nz = (int)((total + ipw * SS_WAVES - 1) / (ipw * SS_WAVES));
    hipLaunchKernelGGL(pad_template_kernel, dim3(grid_for((int64_t)(th + 2 * g.padrows + SS_ROWS) * g.lp)), dim3(256), 0,
                       stream, t, th, tw, g.lp, g.padrows, pack);
    SV_LAUNCH_CHECK();
    size_t lds = (size_t)SS_WAVES * SS_ROWS * WAVE * (4 * (n4 | 1)) * sizeof(float);
    const size_t red = (size_t)SS_WAVES * K * WAVE * sizeof(double);
    if (lds < red) lds = red;
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)nz);
    double *dst = nz > 1 ? partial : out;
    switch (K) {
    case 8: strided_dispatch<8>(n4, grid, lds, stream, g, pack, img, dst); break;
    case 4: strided_dispatch<4>(n4, grid, lds, stream, g, pack, img, dst); break;
    case 2: strided_dispatch<2>(n4, grid, lds, stream, g, pack, img, dst); break;
    default: strided_dispatch<1>(n4, grid, lds, stream, g, pack, img, dst); break;
    }
    SV_LAUNCH_CHECK();
    if (planes_left) {
        *planes_left = (int)nz;                    // the caller adds the planes up itself
    } else if (nz > 1) {
        hipLaunchKernelGGL(slide_reduce_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, stream,
                           partial, nz, plane, out);
        SV_LAUNCH_CHECK();
    }
    return SVAMD_OK;
}

// `partial_cap`: doubles available behind `partial` (only the strided kernel asks).
int slide_launch(const float *t, int th, int tw, const float *img, int64_t img_pitch, int eh, int ew,
                 int oh, int ow, int sx, int sy, double *out, float *pack, double *partial,
                 hipStream_t stream, size_t partial_cap, int *planes_left)
{
    if (sx != 1)
        return slide_strided_launch(t, th, tw, img, img_pitch, eh, ew, oh, ow, sx, sy, out, pack, partial,
                                    partial_cap, stream, planes_left);
    SlideGeom g;
    g.th = th; g.tw = tw; g.lp = slide_lp(tw);
    g.oh = oh; g.ow = ow; g.sy = sy; g.eh = eh; g.ew = ew; g.pitch = img_pitch;
    g.chunk_cols = slide_chunk_cols(oh);
    g.nchunks = slide_chunks(tw, oh);
    g.waves = oh < ST_WAVES ? oh : ST_WAVES;
    g.rowlen = ST_TILE + (g.lp < g.chunk_cols ? g.lp : g.chunk_cols) + 8;
    const int qpr = g.rowlen / 4;
    const int64_t rows_touched = (int64_t)(g.waves - 1) * sy + th;
    const bool wide = ST_STAGE_Q * g.waves * WAVE < qpr;
    g.grows = (wide ? ST_STAGE_Q_WIDE : ST_STAGE_Q) * g.waves * WAVE / qpr;
    if (g.grows > ST_ROWS) g.grows = ST_ROWS;
    if (g.grows > rows_touched) g.grows = (int)rows_touched;
    SV_REQUIRE(g.grows >= 1, SVAMD_ERR_LIMIT, "slide: staging does not fit");
    const int64_t gy = ((int64_t)oh + g.waves - 1) / g.waves;
    SV_REQUIRE(gy <= 65535, SVAMD_ERR_LIMIT, "slide: too many output rows (%d)", oh);
    hipLaunchKernelGGL(pack_template_kernel, dim3(grid_for((int64_t)th * 2 * g.lp)), dim3(256), 0, stream, t,
                       th, tw, g.lp, pack);
    SV_LAUNCH_CHECK();
    // Template chunks are spread over up to ST_MAX_SLICES grid slices when the output
    // alone does not give the chip enough workgroups; a slice sums its chunks itself.
    const int64_t gx = ((int64_t)ow + ST_TILE - 1) / ST_TILE;
    int64_t want = (4096 + gx * gy - 1) / (gx * gy);
    if (want > ST_MAX_SLICES) want = ST_MAX_SLICES;
    if (want > g.nchunks) want = g.nchunks;
    if (want < 1) want = 1;
    g.chunks_per_slice = (int)((g.nchunks + want - 1) / want);
    const int nzc = (g.nchunks + g.chunks_per_slice - 1) / g.chunks_per_slice;
    // few outputs and a tall template: runs of template rows as further grid slices (at
    // least 8 rows each; the workspace holds slide_partial_doubles() planes)
    const int column_planes = g.nchunks < ST_MAX_SLICES ? g.nchunks : ST_MAX_SLICES;
    int row_slices = slide_row_slices(oh, ow, column_planes);
    if (row_slices > ST_MAX_SLICES / nzc) row_slices = ST_MAX_SLICES / nzc;
    const size_t cap = partial_cap ? partial_cap : slide_partial_doubles(tw, oh, ow);
    const int64_t planes_held = (int64_t)(cap / ((size_t)oh * (size_t)ow));
    if (row_slices > planes_held / nzc) row_slices = (int)(planes_held / nzc);
    if (row_slices > th / 8) row_slices = th / 8;
    if (row_slices < 1) row_slices = 1;
    g.rows_per_slice = (th + row_slices - 1) / row_slices;
    g.row_slices = (th + g.rows_per_slice - 1) / g.rows_per_slice;
    const int nz = nzc * g.row_slices;
    dim3 grid((unsigned)gx, (unsigned)gy, nz);
    const size_t lds = (size_t)2 * g.grows * g.rowlen * sizeof(float);
    if (wide)
        hipLaunchKernelGGL(slide_tiled_kernel<ST_STAGE_Q_WIDE>, grid, dim3(g.waves * WAVE), lds, stream, g, pack,
                           img, nz > 1 ? partial : out);
    else
        hipLaunchKernelGGL(slide_tiled_kernel<ST_STAGE_Q>, grid, dim3(g.waves * WAVE), lds, stream, g, pack, img,
                           nz > 1 ? partial : out);
    SV_LAUNCH_CHECK();
    if (planes_left) {
        *planes_left = nz;
    } else if (nz > 1) {
        const int64_t total = (int64_t)oh * ow;
        hipLaunchKernelGGL(slide_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           partial, nz, total, out);
        SV_LAUNCH_CHECK();
    }
    return SVAMD_OK;
}

// ---- epilogues ------------------------------------------------------------------
template <typename TSat>
__device__ __forceinline__ TSat rect_sum(const TSat *__restrict__ s, int64_t tp, int y0, int x0, int y1,
                                         int x1)
{
    return s[(int64_t)y1 * tp + x1] - s[(int64_t)y0 * tp + x1] - s[(int64_t)y1 * tp + x0] +
           s[(int64_t)y0 * tp + x0];
}

// Spatial NCC normalisation.  raw = sum t' * (R - c) over the (cropped) overlap;
// the covariance term of the reference is sum t' * (R - mean(snapshot)) =
// raw + (c - mean(snapshot)) * sum_crop(t').
template <typename TSat>
__global__ void __launch_bounds__(256)
spatial_epilogue_kernel(double *__restrict__ out, int orows, int ocols, int sx, int sy, int th, int tw,
                        int rh, int rw, int region_dtype, const Stats *__restrict__ tst,
                        const Stats *__restrict__ rst, const double *__restrict__ tsat,
                        const TSat *__restrict__ s1, const TSat *__restrict__ s2,
                        const double *__restrict__ planes, int nplanes, const void *__restrict__ region)
{
    const int64_t total = (int64_t)orows * ocols;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    // the sliding kernel's sums: in `out`, or still spread over `nplanes` partial planes
    double raw;
    if (nplanes > 1) {
        raw = 0.0;
        for (int z = 0; z < nplanes; ++z)
            raw += planes[(int64_t)z * total + idx];
    } else {
        raw = out[idx];
    }
    const int i = (int)(idx / ocols), j = (int)(idx - (int64_t)i * ocols);
    const int y0 = i * sy, x0 = j * sx;
    const int h = min(th, rh - y0), w = min(tw, rw - x0);
    const double n = (double)h * (double)w;
    const int64_t tp = (int64_t)rw + 1;
    const TSat a1 = rect_sum(s1, tp, y0, x0, y0 + h, x0 + w);
    const TSat a2 = rect_sum(s2, tp, y0, x0, y0 + h, x0 + w);
    // integer tables: sums of the pixels; float64 tables: sums of (pixel - c)
    double rmean = (double)a1 / n;
    double rstd;
    bool flat;
    if constexpr (std::is_integral<TSat>::value) {
        // exact integer test for a constant snapshot
        const int64_t var_n2 = (int64_t)(h * w) * (int64_t)a2 - (int64_t)a1 * (int64_t)a1;
        flat = var_n2 == 0;
        rstd = sqrt((double)var_n2) / n;
    } else {
        const double m2 = (double)a2 / n;
        double var = m2 - rmean * rmean;
        // The variance is a difference of rounded table entries whose magnitude follows
        // the image, not the window: its absolute error is a few units in the last place
        // of the four corners involved.  A window whose variance is not far above that
        // (a low-contrast patch of a large image) takes its mean and deviation from its
        // own pixels instead, two passes, and counts as constant (NaN, the reference's
        // 0 / 0) only when every pixel has the same value.
        const double c1 = fmax(fmax(fabs((double)s1[(int64_t)y0 * tp + x0]), fabs((double)s1[(int64_t)y0 * tp + x0 + w])),
                               fmax(fabs((double)s1[(int64_t)(y0 + h) * tp + x0]),
                                    fabs((double)s1[(int64_t)(y0 + h) * tp + x0 + w])));
        const double c2 = fmax(fmax(fabs((double)s2[(int64_t)y0 * tp + x0]), fabs((double)s2[(int64_t)y0 * tp + x0 + w])),
                               fmax(fabs((double)s2[(int64_t)(y0 + h) * tp + x0]),
                                    fabs((double)s2[(int64_t)(y0 + h) * tp + x0 + w])));
        const double ulp = 2.220446049250313e-16 * (c2 + 2.0 * fabs(rmean) * c1) / n;
        flat = false;
        if (!(var > 1e7 * ulp)) {
            const double centre = rst->mean;            // what the tables' pixels were centred by
            double sum = 0.0, lo = INFINITY, hi = -INFINITY;
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const double p = load_as_double(region, region_dtype, (int64_t)(y0 + y) * rw + x0 + x);
                    sum += p - centre;
                    lo = fmin(lo, p);
                    hi = fmax(hi, p);
                }
            rmean = sum / n;
            double dev = 0.0;
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const double d = load_as_double(region, region_dtype, (int64_t)(y0 + y) * rw + x0 + x) - centre - rmean;
                    dev += d * d;
                }
            var = dev / n;
            flat = lo == hi;
        }
        rstd = sqrt(var);
    }
    const double tstd = tst->std;
    double shift = -rmean;
    if constexpr (std::is_integral<TSat>::value)      // pixels were centred by the rounded region mean
        shift = rint((double)s1[(int64_t)rh * tp + rw] / ((double)rh * (double)rw)) - rmean;
    const double tcrop = tsat[(int64_t)h * (tw + 1) + w];
    double v = (raw + shift * tcrop) / ((double)th * (double)tw);
    if (tstd == 0.0 || flat)
        v = NAN;
    else
        v /= (tstd * rstd);
    out[idx] = v;
}

// Circular NCC normalisation, in place: either the low-deviation branch's
// constant surface or raw / (std T * T.size * std R).
__global__ void __launch_bounds__(256)
circular_epilogue_kernel(double *__restrict__ out, int64_t total, double tsize, double min_std,
                         double min_mean_diff, const Stats *__restrict__ tst, const Stats *__restrict__ rst)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    if (tst->std < min_std) {
        out[idx] = fabs(tst->mean - rst->mean) < min_mean_diff ? 1.0 : 0.0;
        return;
    }
    out[idx] = out[idx] / (tst->std * tsize) / rst->std;
}

__global__ void __launch_bounds__(256)
scale_1d_kernel(double *__restrict__ out, int64_t total, double divisor, int use_std1, int use_std2,
                const Stats *__restrict__ s1, const Stats *__restrict__ s2)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    double v = out[idx] / divisor;
    if (use_std1 && use_std2)
        v /= (s1->std * s2->std);
    else if (use_std1)
        v /= s1->std;
    out[idx] = v;
}

// Operands of the zero-padded FFT route of the circular correlation, P x Q each:
// tpad = the mean-removed template (cropped to mh x mw), zeros elsewhere; ext = the
// mean-removed region wrapped circularly over eh x ew, zeros elsewhere.  Along a
// side that is a power of two, eh = rh = P: the transform's own period is the
// region's.  Along any other side eh = rh + mh - 1 <= P: the circular correlation
// at size P then equals the region-sized one at every lag k < rh (no template
// element ever meets the padding or its wrap-around).
__global__ void __launch_bounds__(256)
pad_operands_kernel(const void *__restrict__ t, int tdtype, int tw, int mh, int mw, const void *__restrict__ r,
                    int rdtype, int rh, int rw, int eh, int ew, const Stats *__restrict__ tst,
                    const Stats *__restrict__ rst, int P, int Q, float *__restrict__ tpad, float *__restrict__ ext)
{
    const double tmean = tst->mean, rmean = rst->mean;
    const int64_t total = (int64_t)P * Q;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / Q), x = (int)(i - (int64_t)y * Q);
        float tv = 0.f, rv = 0.f;
        if (y < mh && x < mw)
            tv = (float)(load_as_double(t, tdtype, (int64_t)y * tw + x) - tmean);
        if (y < eh && x < ew)
            rv = (float)(load_as_double(r, rdtype, (int64_t)(y % rh) * rw + (x % rw)) - rmean);
        tpad[i] = tv;
        ext[i] = rv;
    }
}

// out[y][x] (rh x rw, float64) from the top-left corner of the P x Q float32 result:
// the low-deviation branch's constant surface or raw / (std T * T.size * std R).
__global__ void __launch_bounds__(256)
circular_epilogue_f32_kernel(const float *__restrict__ raw, int Q, double *__restrict__ out, int rh, int rw,
                             double tsize, double min_std, double min_mean_diff, const Stats *__restrict__ tst,
                             const Stats *__restrict__ rst)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rh * rw)
        return;
    if (tst->std < min_std) {
        out[idx] = fabs(tst->mean - rst->mean) < min_mean_diff ? 1.0 : 0.0;
        return;
    }
    // a constant region: the reference divides zeros by a zero deviation, every element
    // is NaN (the transforms of an all-zero operand are zero only up to rounding)
    if (rst->std == 0.0) {
        out[idx] = NAN;
        return;
    }
    const int y = (int)(idx / rw), x = (int)(idx - (int64_t)y * rw);
    out[idx] = (double)raw[(int64_t)y * Q + x] / (tst->std * tsize) / rst->std;
}

}  // namespace svamd

using namespace svamd;

// ================================ a1 ==========================================
namespace {
struct A1Layout {
    int eh, ew;
    int64_t epitch;
    int orows, ocols;
    size_t n_tsat, n_sat, n_ext, n_tp;
};
A1Layout a1_layout(int th, int tw, int rh, int rw)
{
    A1Layout L;
    L.eh = rh + th - 1;
    L.ew = rw + tw - 1;
    L.epitch = (int64_t)align_up((size_t)L.ew, 4);
    L.n_tp = (size_t)th * tw;
    L.n_tsat = (size_t)(th + 1) * (tw + 1);
    L.n_sat = (size_t)(rh + 1) * (rw + 1);
    L.n_ext = (size_t)L.eh * L.epitch;
    L.orows = L.ocols = 0;
    return L;
}
}  // namespace

extern "C" size_t svamd_cross_correlate_2d_workspace_bytes(int th, int tw, int rh, int rw)
{
    if (th <= 0 || tw <= 0 || rh <= 0 || rw <= 0)
        return 0;
    A1Layout L = a1_layout(th, tw, rh, rw);
    return 2 * 256 + align_up(stats_scratch_doubles() * 8, 256) + align_up(L.n_tp * 4, 256) + 2 * align_up(L.n_tsat * 8, 256) +
           2 * align_up(L.n_sat * 8, 256) + align_up(L.n_ext * 4, 256) +
           align_up(slide_pack_floats(th, tw) * 4, 256) +
           align_up((slide_partial_doubles(tw, rh, rw) + slide_strided_partial_doubles(rh, rw)) * 8, 256);
}

extern "C" int svamd_cross_correlate_2d(const double *tmpl, int th, int tw, const void *region,
                                        int region_dtype, int rh, int rw, int step_x, int step_y,
                                        double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && region && out, SVAMD_ERR_INVALID, "cross_correlate_2d: NULL pointer");
    SV_REQUIRE(th > 0 && tw > 0 && rh > 0 && rw > 0 && step_x > 0 && step_y > 0, SVAMD_ERR_INVALID,
               "cross_correlate_2d: non-positive size or step");
    SV_REQUIRE(region_dtype >= SVAMD_U8 && region_dtype <= SVAMD_F64, SVAMD_ERR_INVALID,
               "cross_correlate_2d: unsupported region dtype %d", region_dtype);
    SV_REQUIRE(th <= rh && tw <= rw, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_2d_workspace_bytes(th, tw, rh, rw),
               SVAMD_ERR_WORKSPACE, "cross_correlate_2d: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    A1Layout L = a1_layout(th, tw, rh, rw);
    const int orows = (rh + step_y - 1) / step_y, ocols = (rw + step_x - 1) / step_x;

    Carver ws(workspace);
    Stats *tst = ws.take<Stats>(1);
    Stats *rst = ws.take<Stats>(1);
    double *sscr = ws.take<double>(stats_scratch_doubles());
    float *tp = ws.take<float>(L.n_tp);
    double *tsat = ws.take<double>(L.n_tsat);
    double *tsat2 = ws.take<double>(L.n_tsat);
    void *s1 = ws.take<double>(L.n_sat);  // int64 or double: both 8 bytes
    void *s2 = ws.take<double>(L.n_sat);
    float *ext = ws.take<float>(L.n_ext);
    float *pack = ws.take<float>(slide_pack_floats(th, tw));
    const size_t partial_cap = slide_partial_doubles(tw, rh, rw) + slide_strided_partial_doubles(rh, rw);
    double *partial = ws.take<double>(partial_cap);

    int rc;
    if ((rc = stats_launch(tmpl, SVAMD_F64, th, tw, tw, true, tst, sscr, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for((int64_t)L.n_tp)), dim3(256), 0, st, tmpl,
                       SVAMD_F64, th, tw, (int64_t)tw, th, tw, tst, tp);
    SV_LAUNCH_CHECK();
    // table of the centred template (sums over top-left crops); the table of its squares is not used
    // (the extended image is written after the tables: until then its buffer is their scratch)
    const size_t ext_bytes = L.n_ext * sizeof(float);
    if ((rc = sat2d_launch<float, double>(tp, th, tw, tw, 0.0, tsat, tsat2, st, ext, ext_bytes)) != SVAMD_OK)
        return rc;
    if (region_dtype == SVAMD_U8) {
        if ((rc = sat2d_any(region, region_dtype, rh, rw, rw, 0.0, s1, s2, st, ext, ext_bytes)) != SVAMD_OK) return rc;
        // The region's mean only serves as the constant the pixels are centred by before
        // they are rounded to float32 (the epilogue puts the window's own mean back): the
        // kernels that need it read the exact integer table's last entry.
    } else {
        // Floating pixels: float64 tables of (pixel - region mean).  Tables of the plain
        // pixels lose a window's variance in the rounding of sums of squares as soon as
        // the image's mean is large against its deviation.
        if ((rc = stats_launch(region, region_dtype, rh, rw, rw, false, rst, sscr, st)) != SVAMD_OK) return rc;
        if ((rc = sat2d_any(region, region_dtype, rh, rw, rw, 0.0, s1, s2, st, ext, ext_bytes, &rst->mean)) !=
            SVAMD_OK)
            return rc;
    }
    hipLaunchKernelGGL(extend_kernel, dim3(grid_for((int64_t)L.n_ext)), dim3(256), 0, st, region,
                       region_dtype, rh, rw, (int64_t)rw, 0, 0, rst, L.eh, L.ew, L.epitch, ext,
                       region_dtype == SVAMD_U8 ? static_cast<const int64_t *>(s1) + (int64_t)rh * (rw + 1) + rw
                                                : nullptr);
    SV_LAUNCH_CHECK();
    int nplanes = 1;     // the epilogue adds the sliding kernel's partial planes up itself
    if ((rc = slide_launch(tp, th, tw, ext, L.epitch, L.eh, L.ew, orows, ocols, step_x, step_y, out, pack,
                           partial, st, partial_cap, &nplanes)) != SVAMD_OK)
        return rc;
    const int64_t total = (int64_t)orows * ocols;
    dim3 eg((unsigned)((total + 255) / 256));
    if (region_dtype == SVAMD_U8)
        hipLaunchKernelGGL((spatial_epilogue_kernel<int64_t>), eg, dim3(256), 0, st, out, orows, ocols,
                           step_x, step_y, th, tw, rh, rw, region_dtype, tst, rst, tsat,
                           static_cast<const int64_t *>(s1), static_cast<const int64_t *>(s2), partial, nplanes,
                           region);
    else
        hipLaunchKernelGGL((spatial_epilogue_kernel<double>), eg, dim3(256), 0, st, out, orows, ocols,
                           step_x, step_y, th, tw, rh, rw, region_dtype, tst, rst, tsat,
                           static_cast<const double *>(s1), static_cast<const double *>(s2), partial, nplanes,
                           region);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ================================ a2 ==========================================
// Two routes.  The sliding kernel costs mh * mw * rh * rw products and its float32
// error does not grow with the shapes.  The FFT route pads both operands to the next
// powers of two above rh + mh - 1 and rw + mw - 1 (any region shape, circular
// semantics kept exactly: see pad_operands_kernel) and costs O(P Q log(P Q)); its
// float32 error is relative to |t| |r|, i.e. it grows like sqrt(region size /
// template size) on the normalised surface, so it is taken only when the (cropped)
// template covers at least a quarter of the region - and only when it is the faster
// one by the two routes' measured rates: the sliding kernel evaluates about 4.5e10
// products per millisecond, the transforms of a P x Q pair with the padding and the
// epilogue take about 0.03 ms + 1.8e-8 ms per element (0.33 ms at 4096 x 4096).
namespace {
constexpr double SLIDE_MS_PER_PRODUCT = 2.2e-11, FFT_MS_FIXED = 0.03, FFT_MS_PER_ELEMENT = 1.8e-8;

int next_pow2(int v)
{
    int p = 4;
    while (p < v)
        p <<= 1;
    return p;
}

struct A2Route {
    bool fft;
    int mh, mw, P, Q;
    int eh, ew;       // rows / columns of the wrapped region inside the P x Q operand
};
A2Route a2_route(int th, int tw, int rh, int rw)
{
    A2Route R;
    R.mh = th < rh ? th : rh;
    R.mw = tw < rw ? tw : rw;
    // a power-of-two side is its own circular size; any other is padded beyond the
    // template's reach
    const bool p2h = (rh & (rh - 1)) == 0 && rh >= 4, p2w = (rw & (rw - 1)) == 0 && rw >= 4;
    R.eh = p2h ? rh : rh + R.mh - 1;
    R.ew = p2w ? rw : rw + R.mw - 1;
    R.P = p2h ? rh : next_pow2(R.eh);
    R.Q = p2w ? rw : next_pow2(R.ew);
    const double products = (double)R.mh * R.mw * (double)rh * rw;
    R.fft = products * SLIDE_MS_PER_PRODUCT > FFT_MS_FIXED + FFT_MS_PER_ELEMENT * (double)R.P * R.Q &&
            4.0 * R.mh * R.mw >= (double)rh * rw && svamd_fft_supported(R.P, R.Q);
    return R;
}
}  // namespace

extern "C" size_t svamd_cross_correlate_2d_spectral_workspace_bytes(int th, int tw, int rh, int rw)
{
    if (th <= 0 || tw <= 0 || rh <= 0 || rw <= 0)
        return 0;
    const A2Route R = a2_route(th, tw, rh, rw);
    const size_t common = 2 * 256 + align_up(stats_scratch_doubles() * 8, 256);
    if (R.fft)
        return common + 3 * align_up((size_t)R.P * R.Q * 4, 256) + align_up(fft_raw_workspace_bytes(R.P, R.Q), 256);
    const int mh = R.mh, mw = R.mw;
    const size_t epitch = align_up((size_t)(rw + mw - 1), 4);
    return common + align_up((size_t)mh * mw * 4, 256) + align_up((size_t)(rh + mh - 1) * epitch * 4, 256) +
           align_up(slide_pack_floats(mh, mw) * 4, 256) + align_up(slide_partial_doubles(mw, rh, rw) * 8, 256);
}

extern "C" int svamd_cross_correlate_2d_spectral(const void *tmpl, int tmpl_dtype, int th, int tw,
                                                 const void *region, int region_dtype, int rh, int rw,
                                                 double min_std, double min_mean_diff, double *out,
                                                 void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && region && out, SVAMD_ERR_INVALID, "cross_correlate_2d_spectral: NULL pointer");
    SV_REQUIRE(th > 0 && tw > 0 && rh > 0 && rw > 0, SVAMD_ERR_INVALID,
               "cross_correlate_2d_spectral: non-positive size");
    SV_REQUIRE(tmpl_dtype >= SVAMD_U8 && tmpl_dtype <= SVAMD_F64 && region_dtype >= SVAMD_U8 &&
                   region_dtype <= SVAMD_F64,
               SVAMD_ERR_INVALID, "cross_correlate_2d_spectral: unsupported dtype");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_2d_spectral_workspace_bytes(th, tw, rh, rw),
               SVAMD_ERR_WORKSPACE,
               "cross_correlate_2d_spectral: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const A2Route R = a2_route(th, tw, rh, rw);
    const int mh = R.mh, mw = R.mw;

    Carver ws(workspace);
    Stats *tst = ws.take<Stats>(1);
    Stats *rst = ws.take<Stats>(1);
    double *sscr = ws.take<double>(stats_scratch_doubles());
    int rc;
    if ((rc = stats_launch(tmpl, tmpl_dtype, th, tw, tw, false, tst, sscr, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(region, region_dtype, rh, rw, rw, false, rst, sscr, st)) != SVAMD_OK) return rc;
    const int64_t total = (int64_t)rh * rw;
    if (R.fft) {
        const size_t n = (size_t)R.P * R.Q;
        float *tpad = ws.take<float>(n);
        float *ext = ws.take<float>(n);
        float *raw = ws.take<float>(n);
        void *fws = ws.take<unsigned char>(fft_raw_workspace_bytes(R.P, R.Q));
        hipLaunchKernelGGL(pad_operands_kernel, dim3(grid_for((int64_t)n)), dim3(256), 0, st, tmpl, tmpl_dtype, tw,
                           mh, mw, region, region_dtype, rh, rw, R.eh, R.ew, tst, rst, R.P, R.Q, tpad, ext);
        SV_LAUNCH_CHECK();
        if ((rc = fft_raw_correlate(tpad, ext, R.P, R.Q, raw, fws, st)) != SVAMD_OK) return rc;
        hipLaunchKernelGGL(circular_epilogue_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           raw, R.Q, out, rh, rw, (double)th * (double)tw, min_std, min_mean_diff, tst, rst);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    const int eh = rh + mh - 1, ew = rw + mw - 1;
    const int64_t epitch = (int64_t)align_up((size_t)ew, 4);
    float *tp = ws.take<float>((size_t)mh * mw);
    float *ext = ws.take<float>((size_t)eh * epitch);
    float *pack = ws.take<float>(slide_pack_floats(mh, mw));
    double *partial = ws.take<double>(slide_partial_doubles(mw, rh, rw));
    // template minus its mean, cropped top-left to the region's shape
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for((int64_t)mh * mw)), dim3(256), 0, st, tmpl,
                       tmpl_dtype, th, tw, (int64_t)tw, mh, mw, tst, tp);
    SV_LAUNCH_CHECK();
    // region minus its exact mean, wrapped circularly
    hipLaunchKernelGGL(extend_kernel, dim3(grid_for((int64_t)eh * epitch)), dim3(256), 0, st, region,
                       region_dtype, rh, rw, (int64_t)rw, 1, 1, rst, eh, ew, epitch, ext, nullptr);
    SV_LAUNCH_CHECK();
    if ((rc = slide_launch(tp, mh, mw, ext, epitch, eh, ew, rh, rw, 1, 1, out, pack, partial, st)) != SVAMD_OK)
        return rc;
    hipLaunchKernelGGL(circular_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       out, total, (double)th * (double)tw, min_std, min_mean_diff, tst, rst);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ---- long 1D correlations through the 2D transforms ---------------------------------
// raw[k] = sum_{n < nt} t[n] * e[n + k], k < L, over arrays cut into rows of Q samples:
// with n = i Q + j and k = a Q + b the sample e[n + k] is element (i + a, j + b) of the
// row image of e when j + b < Q and element (i + a + 1, j + b - Q) otherwise, so
//     raw[a Q + b] = C[a][b] + C[a + 1][b - Q]
// where C is the 2D linear correlation of the two row images.  C comes from the circular
// correlation of P x 2Q arrays whose right halves are zero (column lags -Q < b' < Q never
// wrap onto data) and whose P >= rows(e) + 2 rows hold every row lag without wrapping.
// No 1D transform longer than 4096 points and no twiddle pass between the dimensions is
// needed.  Taken when the sliding kernel would need more than CORR1D_FFT_MIN_PRODUCTS
// products and the row images fit the transforms (nt + L - 1 <= 4094 x 2048 samples); the
// float32 error is relative to |t| |e| and stays below 1e-5 of the result's scale for the
// lengths this admits.
namespace {
constexpr double CORR1D_FFT_MIN_PRODUCTS = 2e9;
struct Corr1dFft {
    bool ok;
    int Q, P, W;       // samples per row, rows and columns of the padded arrays
};
Corr1dFft corr1d_fft_plan(int64_t nt, int64_t L)
{
    Corr1dFft best = {false, 0, 0, 0};
    if ((double)nt * (double)L < CORR1D_FFT_MIN_PRODUCTS)
        return best;
    const int64_t le = L + nt - 1;
    int64_t best_size = 0;
    for (int Q = 2; Q <= 2048; Q <<= 1) {
        const int64_t rows = (le + Q - 1) / Q + 2;
        if (rows > 4096)
            continue;
        int P = 4;
        while (P < rows)
            P <<= 1;
        const int W = 2 * Q;
        if (!svamd_fft_supported(P, W))
            continue;
        const int64_t size = (int64_t)P * W;
        // the smallest arrays; among equals the squarest
        if (!best.ok || size < best_size || (size == best_size && std::abs(P - W) < std::abs(best.P - best.W))) {
            best = {true, Q, P, W};
            best_size = size;
        }
    }
    return best;
}
size_t corr1d_fft_workspace_bytes(const Corr1dFft &f)
{
    return f.ok ? 3 * align_up((size_t)f.P * f.W * 4, 256) + align_up(fft_raw_workspace_bytes(f.P, f.W), 256) : 0;
}

// row images of t (nt samples) and e (le samples), zero elsewhere
__global__ void __launch_bounds__(256)
corr1d_rows_kernel(const float *__restrict__ t, int64_t nt, const float *__restrict__ e, int64_t le, int Q,
                   int W, int64_t total, float *__restrict__ a, float *__restrict__ b)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / W;
        const int col = (int)(i - row * W);
        const int64_t n = row * Q + col;
        a[i] = col < Q && n < nt ? t[n] : 0.0f;
        b[i] = col < Q && n < le ? e[n] : 0.0f;
    }
}

__global__ void __launch_bounds__(256)
corr1d_combine_kernel(const float *__restrict__ c, int Q, int W, int64_t L, double *__restrict__ out)
{
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= L)
        return;
    const int64_t a = k / Q;
    const int b = (int)(k - a * Q);
    out[k] = (double)c[a * W + b] + (double)c[(a + 1) * W + Q + b];
}

int corr1d_fft_launch(const Corr1dFft &f, const float *t, int64_t nt, const float *e, int64_t L, double *out,
                      void *workspace, hipStream_t st)
{
    Carver ws(workspace);
    const size_t elems = (size_t)f.P * f.W;
    float *a = ws.take<float>(elems);
    float *b = ws.take<float>(elems);
    float *c = ws.take<float>(elems);
    void *fws = ws.take<unsigned char>(fft_raw_workspace_bytes(f.P, f.W));
    hipLaunchKernelGGL(corr1d_rows_kernel, dim3(grid_for((int64_t)elems)), dim3(256), 0, st, t, nt, e, L + nt - 1,
                       f.Q, f.W, (int64_t)elems, a, b);
    SV_LAUNCH_CHECK();
    int rc;
    if ((rc = fft_raw_correlate(a, b, f.P, f.W, c, fws, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(corr1d_combine_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, c, f.Q, f.W, L,
                       out);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
}  // namespace

// ================================ a3 ==========================================
extern "C" size_t svamd_cross_correlate_1d_workspace_bytes(int64_t n)
{
    if (n <= 0)
        return 0;
    return 2 * 256 + align_up(stats_scratch_doubles() * 8, 256) + align_up((size_t)n * 4, 256) + align_up((size_t)(3 * n - 2) * 4, 256) +
           align_up(slide_pack_floats(1, (int)n) * 4, 256) +
           align_up(slide_partial_doubles((int)n, 1, 2 * n - 1) * 8, 256) +
           corr1d_fft_workspace_bytes(corr1d_fft_plan(n, 2 * n - 1));
}

// The launches of cross_correlate_1d over a workspace of at least
// svamd_cross_correlate_1d_workspace_bytes(n) bytes (arguments already checked).
static int xcorr1d_enqueue(const double *v1, const double *v2, int64_t n, int normalised, double *out,
                           void *workspace, hipStream_t st)
{
    Carver ws(workspace);
    Stats *s1 = ws.take<Stats>(1);
    Stats *s2 = ws.take<Stats>(1);
    double *sscr = ws.take<double>(stats_scratch_doubles());
    float *tp = ws.take<float>((size_t)n);
    float *ext = ws.take<float>((size_t)(3 * n - 2));
    float *pack = ws.take<float>(slide_pack_floats(1, (int)n));
    double *partial = ws.take<double>(slide_partial_doubles((int)n, 1, 2 * n - 1));
    int rc;
    if ((rc = stats_launch(v1, SVAMD_F64, 1, (int)n, n, false, s1, sscr, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(v2, SVAMD_F64, 1, (int)n, n, false, s2, sscr, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for(n)), dim3(256), 0, st, v1, SVAMD_F64, 1, (int)n,
                       n, 1, (int)n, s1, tp);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(extend_1d_full_kernel, dim3(grid_for(3 * n - 2)), dim3(256), 0, st, v2, n, s2, ext);
    SV_LAUNCH_CHECK();
    const int64_t total = 2 * n - 1;
    const Corr1dFft fft = corr1d_fft_plan(n, total);
    if (fft.ok)
        rc = corr1d_fft_launch(fft, tp, n, ext, total, out, ws.take<unsigned char>(corr1d_fft_workspace_bytes(fft)), st);
    else
        rc = slide_launch(tp, 1, (int)n, ext, 3 * n - 2, 1, (int)(3 * n - 2), 1, (int)total, 1, 1, out, pack,
                          partial, st);
    if (rc != SVAMD_OK)
        return rc;
    hipLaunchKernelGGL(scale_1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total,
                       (double)n, normalised ? 1 : 0, normalised ? 1 : 0, s1, s2);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" int svamd_cross_correlate_1d(const double *v1, const double *v2, int64_t n, int normalised,
                                        double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(v1 && v2 && out, SVAMD_ERR_INVALID, "cross_correlate_1d: NULL pointer");
    SV_REQUIRE(n > 0 && n < (1LL << 30), SVAMD_ERR_INVALID, "cross_correlate_1d: bad length %lld",
               (long long)n);
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_1d_workspace_bytes(n),
               SVAMD_ERR_WORKSPACE, "cross_correlate_1d: workspace missing, misaligned or too small");
    return xcorr1d_enqueue(v1, v2, n, normalised, out, workspace, static_cast<hipStream_t>(stream));
}

// ---- cross_correlate_1d without a workspace argument and without allocation ------
// The workspace lives in the code object: every device that loads the library holds
// its own copy of this array, at its own address, until the process ends.  In host
// code the array's name is not a device address; the address on the current device
// comes from hipGetSymbolAddress, once per device.
constexpr int64_t XCORR1D_STATIC_MAX_N = SVAMD_XCORR1D_STATIC_MAX_N;
constexpr size_t XCORR1D_STATIC_BYTES = 2 * 256 + 2 * 512 * 8 /* statistics partials */ +
                                        (size_t)XCORR1D_STATIC_MAX_N * (4 + 12 + 24 + 4) /* operands, slack */ +
                                        (size_t)16 * 2 * XCORR1D_STATIC_MAX_N * 8 /* partial sums */ + 8 * 256 +
                                        ((size_t)11 << 20) /* padded arrays and spectra of the transform route */;
static __device__ __align__(256) unsigned char xcorr1d_scratch[XCORR1D_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_mutex;
void *scratch_ptr[MAX_DEVICES] = {};
}  // namespace

extern "C" int svamd_cross_correlate_1d_static(const double *v1, const double *v2, int64_t n,
                                               int normalised, double *out, void *stream)
{
    SV_REQUIRE(v1 && v2 && out, SVAMD_ERR_INVALID, "cross_correlate_1d_static: NULL pointer");
    SV_REQUIRE(n > 0, SVAMD_ERR_INVALID, "cross_correlate_1d_static: bad length %lld", (long long)n);
    SV_REQUIRE(n <= XCORR1D_STATIC_MAX_N, SVAMD_ERR_LIMIT,
               "cross_correlate_1d_static: n = %lld exceeds the built-in workspace's limit of %lld samples; "
               "use svamd_cross_correlate_1d with a workspace",
               (long long)n, (long long)XCORR1D_STATIC_MAX_N);
    SV_REQUIRE(svamd_cross_correlate_1d_workspace_bytes(n) <= XCORR1D_STATIC_BYTES, SVAMD_ERR_LIMIT,
               "cross_correlate_1d_static: built-in workspace too small for n = %lld", (long long)n);
    int dev = -1;
    SV_HIP(hipGetDevice(&dev));
    SV_REQUIRE(dev >= 0 && dev < MAX_DEVICES, SVAMD_ERR_LIMIT, "cross_correlate_1d_static: device index %d", dev);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (st) {
        // The array is looked up on the current device; kernels of a stream that lives on
        // another device would be handed an address that means nothing there.
        int stream_dev = -1;
        SV_HIP(hipStreamGetDevice(st, &stream_dev));
        SV_REQUIRE(stream_dev == dev, SVAMD_ERR_INVALID,
                   "cross_correlate_1d_static: the stream belongs to device %d but device %d is current; make the "
                   "stream's device current before the call", stream_dev, dev);
    }
    void *ws;
    {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        if (!scratch_ptr[dev]) {
            // First call on this device.  Looking the address up is not a stream operation:
            // keep it out of a graph capture (one plain call per device comes first).
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (st)
                SV_HIP(hipStreamIsCapturing(st, &capturing));
            SV_REQUIRE(capturing == hipStreamCaptureStatusNone, SVAMD_ERR_INVALID,
                       "cross_correlate_1d_static: the first call on a device must be made outside stream "
                       "capture (it looks up the built-in workspace's address)");
            void *p = nullptr;
            SV_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(xcorr1d_scratch)));
            SV_REQUIRE(p && (reinterpret_cast<uintptr_t>(p) & 255) == 0, SVAMD_ERR_HIP,
                       "cross_correlate_1d_static: built-in workspace has no usable device address");
            scratch_ptr[dev] = p;
        }
        ws = scratch_ptr[dev];
    }
    return xcorr1d_enqueue(v1, v2, n, normalised, out, ws, static_cast<hipStream_t>(stream));
}

// ================================ a5 ==========================================
// The circular correlation sums nt * N products when the template slides over the
// padded signal and ns * N when the signal slides over the padded template: the
// shorter array slides.
extern "C" size_t svamd_cross_correlate_1d_spectral_workspace_bytes(int64_t nt, int64_t ns)
{
    if (nt <= 0 || ns <= 0)
        return 0;
    const int64_t N = ns + nt - 1;
    const size_t common = 256 + align_up(stats_scratch_doubles() * 8, 256);
    if (ns < nt)
        return common + align_up((size_t)ns * 4, 256) + align_up((size_t)(N + ns - 1) * 4, 256) +
               align_up((size_t)N * 8, 256) + align_up(slide_pack_floats(1, (int)ns) * 4, 256) +
               align_up(slide_partial_doubles((int)ns, 1, N) * 8, 256) +
               corr1d_fft_workspace_bytes(corr1d_fft_plan(ns, N));
    return common + align_up((size_t)nt * 4, 256) + align_up((size_t)(ns + 2 * nt - 2) * 4, 256) +
           align_up(slide_pack_floats(1, (int)nt) * 4, 256) +
           align_up(slide_partial_doubles((int)nt, 1, N) * 8, 256) +
           corr1d_fft_workspace_bytes(corr1d_fft_plan(nt, N));
}

extern "C" int svamd_cross_correlate_1d_spectral(const double *tmpl, int64_t nt, const double *signal,
                                                 int64_t ns, double *out, void *workspace,
                                                 size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && signal && out, SVAMD_ERR_INVALID, "cross_correlate_1d_spectral: NULL pointer");
    SV_REQUIRE(nt > 0 && ns > 0 && nt + ns < (1LL << 30), SVAMD_ERR_INVALID,
               "cross_correlate_1d_spectral: bad lengths");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_1d_spectral_workspace_bytes(nt, ns),
               SVAMD_ERR_WORKSPACE,
               "cross_correlate_1d_spectral: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver ws(workspace);
    Stats *s1 = ws.take<Stats>(1);
    double *sscr = ws.take<double>(stats_scratch_doubles());
    const int64_t N = ns + nt - 1;
    int rc;
    if ((rc = stats_launch(tmpl, SVAMD_F64, 1, (int)nt, nt, false, s1, sscr, st)) != SVAMD_OK) return rc;
    if (ns < nt) {
        // c[k] = sum_j signal[j] * tN[(nt - 1 + j - k) mod N]: the raw signal slides over the
        // wrapped, zero-padded template; output position u = (nt - 1 - k) mod N
        float *sp = ws.take<float>((size_t)ns);
        float *ext = ws.take<float>((size_t)(N + ns - 1));
        double *raw = ws.take<double>((size_t)N);
        float *pack = ws.take<float>(slide_pack_floats(1, (int)ns));
        double *partial = ws.take<double>(slide_partial_doubles((int)ns, 1, N));
        hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(ns)), dim3(256), 0, st, signal, ns, sp);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL(extend_1d_template_kernel, dim3(grid_for(N + ns - 1)), dim3(256), 0, st, tmpl, nt, N,
                           N + ns - 1, s1, ext);
        SV_LAUNCH_CHECK();
        const Corr1dFft fft = corr1d_fft_plan(ns, N);
        if (fft.ok)
            rc = corr1d_fft_launch(fft, sp, ns, ext, N, raw, ws.take<unsigned char>(corr1d_fft_workspace_bytes(fft)),
                                   st);
        else
            rc = slide_launch(sp, 1, (int)ns, ext, N + ns - 1, 1, (int)(N + ns - 1), 1, (int)N, 1, 1, raw, pack,
                              partial, st);
        if (rc != SVAMD_OK)
            return rc;
        hipLaunchKernelGGL(scale_1d_swapped_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, raw, N,
                           nt, s1, out);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    float *tp = ws.take<float>((size_t)nt);
    float *ext = ws.take<float>((size_t)(ns + 2 * nt - 2));
    float *pack = ws.take<float>(slide_pack_floats(1, (int)nt));
    double *partial = ws.take<double>(slide_partial_doubles((int)nt, 1, N));
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for(nt)), dim3(256), 0, st, tmpl, SVAMD_F64, 1,
                       (int)nt, nt, 1, (int)nt, s1, tp);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(extend_1d_circular_kernel, dim3(grid_for(N + nt - 1)), dim3(256), 0, st, signal, nt,
                       ns, ext);
    SV_LAUNCH_CHECK();
    const Corr1dFft fft = corr1d_fft_plan(nt, N);
    if (fft.ok)
        rc = corr1d_fft_launch(fft, tp, nt, ext, N, out, ws.take<unsigned char>(corr1d_fft_workspace_bytes(fft)), st);
    else
        rc = slide_launch(tp, 1, (int)nt, ext, N + nt - 1, 1, (int)(N + nt - 1), 1, (int)N, 1, 1, out, pack,
                          partial, st);
    if (rc != SVAMD_OK)
        return rc;
    hipLaunchKernelGGL(scale_1d_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, out, N,
                       (double)nt, 1, 0, s1, s1);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
