// K3-large, sides of 256 ... 4096 points: the three passes of fft_xcorr.hip (rows forward
// with the two real arrays packed into one complex transform, columns forward x
// conjugate product x columns inverse, rows inverse two at a time) with N / 16 threads
// per transform.  Compiled without the SLP vectoriser: packed float32 arithmetic across
// the butterflies' operands costs this code about thirty registers per thread.
#include "fft_common.h"
#include "kernels.h"

namespace svamd {

// =====================================================================================
// Wide transforms: N = 256 ... 4096 points, N / 16 threads per transform, 16 elements
// per thread.  The same in-place schedule as above with the radix-2 stage (odd log2 N)
// last, cut into three phases whose butterflies find their operands in one thread:
//   A  lengths N and N/4       thread t holds positions t + (N/16) m
//   B  lengths N/16 and N/64   thread t holds (N/16) (t / S) + t % S + S m,  S = N/256
//   C  what is left (S points per block): thread t holds S (t + (N/16) u) + e
// i.e. two exchanges through LDS per transform (one for 256 points), each of them a
// barrier of the transform's threads - a wavefront barrier up to 1024 points, a
// workgroup barrier for the two or four wavefronts of 2048 and 4096 points - where the
// workgroup-per-transform kernels above have one per stage.  Frequency k ends at
// position wide_pos(k).  Twiddles: twelve per thread, read once from the table; the
// factors of a stage's other butterflies differ from those by powers of exp(-2 pi i / 16).
// LDS positions are padded (idx) so that the three access patterns are (nearly) free of
// bank conflicts while their addresses stay base + immediate.  profiles/fft_wide_model.py
// runs this schedule in numpy and counts the conflicts.
// =====================================================================================
template <int LOG2N>
struct Wide {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 16;
    static constexpr int SB = LOG2N - 8;
    static constexpr int S = 1 << SB;
    static constexpr int LC = S;
    static constexpr int SPAN = N + N / 16;  // LDS elements of one transform
    static constexpr int PITCH = SPAN;       // transforms side by side (256 points: 272 = 16 mod 32,
                                             // the two transforms of a half wavefront on different banks)
    static_assert(LOG2N >= 8 && LOG2N <= 12, "wide transforms: 256 to 4096 points");

    // LDS element of position p: one element of padding after every 16.  Every access
    // pattern is then a thread-dependent base plus a compile-time offset (an instruction's
    // immediate), 8-byte writes (16 lanes over 16 bank pairs) are free of conflicts and
    // 8-byte reads (32 lanes over 32 bank pairs) meet at most two-way on one bank pair.
    static constexpr __host__ __device__ int idx(int p) { return p + (p >> 4); }
    static __device__ __forceinline__ int base1(int t) { return idx(t); }
    static __device__ __forceinline__ int base2(int t) { return idx((N / 16) * (t >> SB) + (t & (S - 1))); }
    static __device__ __forceinline__ int base3(int t) { return idx(LC * t); }
    static constexpr __host__ __device__ int off1(int m) { return idx(T * m); }
    static constexpr __host__ __device__ int off2(int m) { return idx(S * m); }
    static constexpr __host__ __device__ int off3(int i) { return idx(LC * T * (i / LC)) + i % LC; }
    // where element i of thread t sits after the forward transform
    static __device__ __forceinline__ int out_index(int t, int i)
    {
        return SB ? (base3(t) + off3(i)) : (base2(t) + off2(i));
    }
    // position of frequency k in the transformed sequence
    static __device__ __forceinline__ int pos(int k)
    {
        int out = 0;
#pragma unroll
        for (int b = 0; b + 1 < LOG2N; b += 2) {
            out |= (k & 3) << (LOG2N - 2 - b);
            k >>= 2;
        }
        if (LOG2N & 1)
            out |= k & 1;
        return out;
    }
    static __device__ __forceinline__ void sync()
    {
        if (T <= 64)
            __builtin_amdgcn_wave_barrier();
        else
            __syncthreads();
    }
};

struct WideTw {
    float2 a1[3], a2[3], b1[3], b2[3];
};

template <int LOG2N>
__device__ __forceinline__ void wide_load_twiddles(WideTw &w, const float2 *__restrict__ table, int t)
{
    const int s = t & (Wide<LOG2N>::S - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        w.a1[c] = table[t * (c + 1)];
        w.a2[c] = table[4 * t * (c + 1)];
        w.b1[c] = table[16 * s * (c + 1)];
        w.b2[c] = table[64 * s * (c + 1)];
    }
}

// Makes the twiddles opaque to the optimiser at this point, so that what a transform
// derives from them (eighteen more factors) is computed where it is used: hoisted out of
// a kernel's row loop or shared between a kernel's three transforms, those factors would
// occupy some eighty registers for the whole kernel.
__device__ __forceinline__ void wide_touch(WideTw &w)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        asm volatile("" : "+v"(w.a1[c].x), "+v"(w.a1[c].y), "+v"(w.a2[c].x), "+v"(w.a2[c].y));
        asm volatile("" : "+v"(w.b1[c].x), "+v"(w.b1[c].y), "+v"(w.b2[c].x), "+v"(w.b2[c].y));
    }
}

// a * exp(-2 pi i K / 16)
template <int K>
__device__ __forceinline__ float2 mul_w16(float2 a)
{
    constexpr int k = ((K % 16) + 16) % 16;
    constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f, R = 0.70710678118654752f;
    constexpr float cosv[16] = {1.f, C, R, S, 0.f, -S, -R, -C, -1.f, -C, -R, -S, 0.f, S, R, C};
    constexpr float sinv[16] = {0.f, S, R, C, 1.f, C, R, S, 0.f, -S, -R, -C, -1.f, -C, -R, -S};
    if constexpr (k == 0)
        return a;
    else if constexpr (k == 4)
        return mul_mi(a);
    else if constexpr (k == 8)
        return make_float2(-a.x, -a.y);
    else if constexpr (k == 12)
        return mul_pi(a);
    else
        return cmul(a, make_float2(cosv[k], -sinv[k]));
}

// butterflies M0 of a phase's first stage: twiddles w[c] * exp(-2 pi i M0 (c + 1) / 16)
template <int M0>
__device__ __forceinline__ void wide_dif_step(float2 (&v)[16], const float2 (&w)[3])
{
    dif4(v[M0], v[M0 + 4], v[M0 + 8], v[M0 + 12], mul_w16<M0>(w[0]), mul_w16<2 * M0>(w[1]), mul_w16<3 * M0>(w[2]));
}
template <int M0>
__device__ __forceinline__ void wide_dit_step(float2 (&v)[16], const float2 (&w)[3])
{
    dit4(v[M0], v[M0 + 4], v[M0 + 8], v[M0 + 12], mul_w16<M0>(w[0]), mul_w16<2 * M0>(w[1]), mul_w16<3 * M0>(w[2]));
}
// butterflies whose twiddles are the constants exp(-2 pi i K / 16)
template <int K1, int K2, int K3>
__device__ __forceinline__ void dif4_w16(float2 &a, float2 &b, float2 &c, float2 &d)
{
    dif4_plain(a, b, c, d);
    b = mul_w16<K1>(b);
    c = mul_w16<K2>(c);
    d = mul_w16<K3>(d);
}
template <int K1, int K2, int K3>
__device__ __forceinline__ void dit4_w16(float2 &a, float2 &b, float2 &c, float2 &d)
{
    b = mul_w16<16 - K1>(b);
    c = mul_w16<16 - K2>(c);
    d = mul_w16<16 - K3>(d);
    dit4_plain(a, b, c, d);
}
__device__ __forceinline__ void radix2(float2 &a, float2 &b)
{
    const float2 s = cadd(a, b);
    b = csub(a, b);
    a = s;
}

template <int LC>
__device__ __forceinline__ void wide_last_forward(float2 (&v)[16])
{
    if constexpr (LC == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            radix2(v[2 * q], v[2 * q + 1]);
    } else if constexpr (LC == 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            dif4_plain(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
    } else if constexpr (LC == 8) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float2 *b = v + 8 * u;
            dif4_plain(b[0], b[2], b[4], b[6]);
            dif4_w16<2, 4, 6>(b[1], b[3], b[5], b[7]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                radix2(b[2 * q], b[2 * q + 1]);
        }
    } else if constexpr (LC == 16) {
        dif4_plain(v[0], v[4], v[8], v[12]);
        dif4_w16<1, 2, 3>(v[1], v[5], v[9], v[13]);
        dif4_w16<2, 4, 6>(v[2], v[6], v[10], v[14]);
        dif4_w16<3, 6, 9>(v[3], v[7], v[11], v[15]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dif4_plain(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
}
template <int LC>
__device__ __forceinline__ void wide_last_inverse(float2 (&v)[16])
{
    if constexpr (LC == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            radix2(v[2 * q], v[2 * q + 1]);
    } else if constexpr (LC == 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            dit4_plain(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
    } else if constexpr (LC == 8) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float2 *b = v + 8 * u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                radix2(b[2 * q], b[2 * q + 1]);
            dit4_plain(b[0], b[2], b[4], b[6]);
            dit4_w16<2, 4, 6>(b[1], b[3], b[5], b[7]);
        }
    } else if constexpr (LC == 16) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dit4_plain(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        dit4_plain(v[0], v[4], v[8], v[12]);
        dit4_w16<1, 2, 3>(v[1], v[5], v[9], v[13]);
        dit4_w16<2, 4, 6>(v[2], v[6], v[10], v[14]);
        dit4_w16<3, 6, 9>(v[3], v[7], v[11], v[15]);
    }
}

// Forward transform.  In: v[m] = position t + (N/16) m of the sequence in natural order.
// Out: element i at position Wide::out_index(t, i) (swizzled) of the transformed sequence.
// x: the transform's LDS region (N elements); every thread of the transform's barrier
// domain (its wavefront, or the workgroup from 2048 points) must be here.  The last LDS
// operations are this thread's reads of its out_index positions.
template <int LOG2N>
__device__ __forceinline__ void wide_forward(float2 (&v)[16], float2 *x, WideTw &w, int t)
{
    using P = Wide<LOG2N>;
    wide_touch(w);
    wide_dif_step<0>(v, w.a1);
    wide_dif_step<1>(v, w.a1);
    wide_dif_step<2>(v, w.a1);
    wide_dif_step<3>(v, w.a1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        dif4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w.a2[0], w.a2[1], w.a2[2]);
    const int i1 = P::base1(t), i2 = P::base2(t);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[i1 + P::off1(m)] = v[m];
    P::sync();
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[i2 + P::off2(m)];
    wide_dif_step<0>(v, w.b1);
    wide_dif_step<1>(v, w.b1);
    wide_dif_step<2>(v, w.b1);
    wide_dif_step<3>(v, w.b1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        dif4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w.b2[0], w.b2[1], w.b2[2]);
    if constexpr (P::SB > 0) {
        // written where this thread read: nobody else touches these positions in between
#pragma unroll
        for (int m = 0; m < 16; ++m)
            x[i2 + P::off2(m)] = v[m];
        P::sync();
        const int i3 = P::base3(t);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = x[i3 + P::off3(i)];
        wide_last_forward<P::LC>(v);
    }
}

// Inverse transform (unscaled).  In: element i = position out_index(t, i) of the
// transformed sequence; the first LDS operations are writes to exactly those positions.
// Out: v[m] = position t + (N/16) m in natural order; the last LDS operations are this
// thread's reads of positions idx(t + (N/16) m).
template <int LOG2N>
__device__ __forceinline__ void wide_inverse(float2 (&v)[16], float2 *x, WideTw &w, int t)
{
    using P = Wide<LOG2N>;
    wide_touch(w);
    const int i1 = P::base1(t), i2 = P::base2(t);
    if constexpr (P::SB > 0) {
        wide_last_inverse<P::LC>(v);
        const int i3 = P::base3(t);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[i3 + P::off3(i)] = v[i];
        P::sync();
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = x[i2 + P::off2(m)];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        dit4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w.b2[0], w.b2[1], w.b2[2]);
    wide_dit_step<0>(v, w.b1);
    wide_dit_step<1>(v, w.b1);
    wide_dit_step<2>(v, w.b1);
    wide_dit_step<3>(v, w.b1);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[i2 + P::off2(m)] = v[m];
    P::sync();
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[i1 + P::off1(m)];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        dit4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], w.a2[0], w.a2[1], w.a2[2]);
    wide_dit_step<0>(v, w.a1);
    wide_dit_step<1>(v, w.a1);
    wide_dit_step<2>(v, w.a1);
    wide_dit_step<3>(v, w.a1);
}

// Sum over the T threads of a transform that share a wavefront (T = 16, 32) or over the
// wavefront (T >= 64).
template <int T>
__device__ __forceinline__ double wide_group_sum(double v)
{
    if constexpr (T >= 64) {
        return wave_sum(v);
    } else {
#pragma unroll
        for (int o = T / 2; o > 0; o >>= 1)
            v += __shfl_xor(v, o, WAVE);
        return v;
    }
}

// Layout of the half spectra between the wide passes: [pair][group of 4 columns][row][4
// columns] (as the 1024 x 1024 path: a column group of the column pass is consecutive
// memory, the row passes write / read 4 rows x 4 columns as one 128-byte line).
__device__ __forceinline__ int64_t wide_spec_index(int H, int y, int k)
{
    return (int64_t)(k >> 2) * ((int64_t)H * 4) + y * 4 + (k & 3);
}

// Twiddle tables of both axes and, per pair, the constants subtracted from the arrays
// before the float32 transforms (which keeps a large DC term out of the butterflies).  A
// full-size array only needs an approximate centre (the mean of its first row): the exact
// mean differs from it by a constant, i.e. only in the DC bin, which pass 2 clears; that
// also holds for a template cropped to the region's shape.  A template that is zero-padded
// along an axis needs its exact mean (a constant inside a window is not a pure DC term).
// Blocks [0, pairs) take one pair each, the blocks after them the tables.
__global__ void __launch_bounds__(1024)
wide_setup_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H, int W,
                  FftStats *__restrict__ stats, int pairs, float2 *__restrict__ table_w,
                  float2 *__restrict__ table_h)
{
    __shared__ double red[17];
    if ((int)blockIdx.x >= pairs) {
        const int j = ((int)blockIdx.x - pairs) * 1024 + threadIdx.x;
        if (j < W + H) {
            const int n = j < W ? W : H, k = j < W ? j : j - W;
            double sn, cs;
            sincospi(-2.0 * (double)k / (double)n, &sn, &cs);
            (j < W ? table_w : table_h)[k] = make_float2((float)cs, (float)sn);
        }
        return;
    }
    const int pair = blockIdx.x;
    const float *tp = t + (int64_t)pair * th * tw_;
    const float *rp = r + (int64_t)pair * H * W;
    const bool exact_t = th < H || tw_ < W;
    const int64_t tcount = exact_t ? (int64_t)th * tw_ : tw_;
    double ts = 0.0, rs = 0.0;
    for (int64_t base = threadIdx.x; base < tcount; base += 8 * 1024) {     // eight loads in flight
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = base + k * 1024 < tcount ? tp[base + k * 1024] : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            ts += (double)v[k];
    }
    for (int i = threadIdx.x; i < W; i += 1024)
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

template <int LOG2W>
struct WideRows {
    using P = Wide<LOG2W>;
    static constexpr int THREADS = 4 * P::T > 256 ? 4 * P::T : 256;
    static constexpr int RG = THREADS / P::T;            // transforms side by side in a workgroup
    static constexpr int PARTS = P::T > 64 ? P::T / 64 : 1;   // partial sums per row
    static constexpr int GROUPS = P::N / 8 + 1;          // column groups of a half spectrum
    static constexpr size_t LDS_BYTES = (size_t)RG * P::PITCH * sizeof(float2);
};

// pass 1: a workgroup transforms RG image rows at a time (rows_per_wg in all), the next
// rows' pixels on their way meanwhile; then all its threads split the packed spectra
// and store them, RG rows x 4 columns of a column group as consecutive memory.  Also the
// exact sums of (x - centre) and (x - centre)^2 of both arrays, PARTS entries per row.
// raw != 0: the arrays are transformed as they are (no centre, no sums).
template <int LOG2W>
__global__ void __launch_bounds__(WideRows<LOG2W>::THREADS)
wide_rows_forward_kernel(const float *__restrict__ t, int th, int tw_, const float *__restrict__ r, int H,
                         const FftStats *__restrict__ stats, double *__restrict__ row_sums,
                         const float2 *__restrict__ table, float2 *__restrict__ spec_t,
                         float2 *__restrict__ spec_r, int hp, int rows_per_wg, int raw)
{
    using R = WideRows<LOG2W>;
    using P = Wide<LOG2W>;
    constexpr int W = P::N, T = P::T, RG = R::RG;
    extern __shared__ __align__(16) float2 wide_lds[];
    const int tr = threadIdx.x / T, tt = threadIdx.x % T;
    const int y_block = blockIdx.x * rows_per_wg, pair = blockIdx.y;
    float2 *x = wide_lds + tr * P::PITCH;
    WideTw w;
    wide_load_twiddles<LOG2W>(w, table, tt);
    const float *tbase = t + (int64_t)pair * th * tw_;
    const float *rbase = r + (int64_t)pair * H * W;
    const float tc = raw ? 0.f : stats[pair].t_centre, rc = raw ? 0.f : stats[pair].r_centre;
    const int mw = min(tw_, W), mh = min(th, H);
    float2 *ot = spec_t + (int64_t)pair * H * hp;
    float2 *orr = spec_r + (int64_t)pair * H * hp;

    float tn[16], rn[16];
    auto fetch = [&](int y) {
        const float *trow = tbase + (int64_t)y * tw_;
        const float *rrow = rbase + (int64_t)y * W;
        const bool trow_in = y < mh;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int xx = tt + T * m;
            tn[m] = (trow_in && xx < mw) ? trow[xx] : tc;
            rn[m] = rrow[xx];
        }
    };
    // 4096-point rows: 1024 threads leave 128 registers per thread, not enough to hold the
    // next row's pixels next to the transform
    constexpr bool PREFETCH = LOG2W < 12;
    if (PREFETCH)
        fetch(y_block + tr);
    for (int y_group = y_block; y_group < y_block + rows_per_wg; y_group += RG) {
        const int y = y_group + tr;
        if (!PREFETCH)
            fetch(y);
        float2 v[16];
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
        if (PREFETCH && y_group + RG < y_block + rows_per_wg)
            fetch(y + RG);
        if (!raw) {
            double ts = fts, tq = ftq, rs = frs, rq = frq;
            // the template's statistics cover the whole template: add what the crop to the
            // region's shape left out (columns beyond W, rows beyond H)
            for (int yy = y; yy < th; yy += H)
                for (int xx = (yy == y && y < mh ? mw : 0) + tt; xx < tw_; xx += T) {
                    const double d = (double)(tbase[(int64_t)yy * tw_ + xx] - tc);
                    ts += d;
                    tq += d * d;
                }
            ts = wide_group_sum<T>(ts);
            tq = wide_group_sum<T>(tq);
            rs = wide_group_sum<T>(rs);
            rq = wide_group_sum<T>(rq);
            if ((tt & 63) == 0)
                reinterpret_cast<double4 *>(row_sums)[((int64_t)pair * H + y) * R::PARTS + (tt >> 6)] =
                    make_double4(ts, tq, rs, rq);
        }
        wide_forward<LOG2W>(v, x, w, tt);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[P::out_index(tt, i)] = v[i];
        __syncthreads();
        // split z = fft(t + i r) into the two half spectra; item -> (column group, row, half)
        for (int item = threadIdx.x; item < R::GROUPS * 2 * RG; item += R::THREADS) {
            const int half = item & 1, row = (item >> 1) % RG, g = item / (2 * RG), k0 = 4 * g + 2 * half;
            if (k0 > W / 2)
                continue;
            const float2 *xs = wide_lds + row * P::PITCH;
            float2 ta[2], ra[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = min(k0 + c, W / 2);
                const float2 zk = xs[P::idx(P::pos(k))];
                const float2 zm = xs[P::idx(P::pos((W - k) & (W - 1)))];
                ta[c] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                ra[c] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
            }
            const int64_t at = wide_spec_index(H, y_group + row, k0);
            if (k0 + 1 <= W / 2) {
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

// pass 2: a tile of TC spectrum columns (TC = 4: one column group; 2: half of one, for
// 4096 rows, where a whole group would fill the LDS of a CU with one workgroup).  Column
// c's threads transform column c of T (kept in registers), then column c of R, multiply
// and transform back; the R tile is on its way from memory during the T transform.
template <int LOG2H>
struct WideCols {
    using P = Wide<LOG2H>;
    static constexpr int TC = LOG2H == 12 ? 2 : 4;
    static constexpr int THREADS = TC * P::T;
    static constexpr int COL = P::SPAN + 4;                         // column pitch in LDS (bank spread of the transposes)
    static constexpr int QUADS = P::N * TC / 2 / THREADS;           // 16-byte pieces per thread and tile (8)
    static constexpr size_t LDS_BYTES = (size_t)TC * COL * sizeof(float2);
};

template <int LOG2H>
__global__ void __launch_bounds__(WideCols<LOG2H>::THREADS)
wide_cols_fused_kernel(float2 *__restrict__ spec_t, const float2 *__restrict__ spec_r, int groups, int hp,
                       const float2 *__restrict__ table, int raw, const double *__restrict__ row_sums,
                       int sums_per_pair, FftStats *__restrict__ stats)
{
    using C = WideCols<LOG2H>;
    using P = Wide<LOG2H>;
    constexpr int H = P::N, T = P::T, TC = C::TC;
    extern __shared__ __align__(16) float2 wide_lds[];
    // blocks b and b + 8 (one XCD, dispatched close together) take the two halves of a group
    int g, half;
    if (TC == 4) {
        g = blockIdx.x;
        half = 0;
    } else {
        g = 8 * (blockIdx.x / 16) + (blockIdx.x % 8);
        half = (blockIdx.x / 8) & 1;
    }
    if (g >= groups || (TC == 2 && g == groups - 1 && half))
        return;                       // the last group holds one column only
    const int pair = blockIdx.y;
    const int col = threadIdx.x / T, tt = threadIdx.x % T;
    float2 *x = wide_lds + col * C::COL;
    WideTw w;
    wide_load_twiddles<LOG2H>(w, table, tt);
    const int64_t tile_at = (int64_t)pair * H * hp + (int64_t)g * (H * 4) + half * 2;
    float2 *gt = spec_t + tile_at;
    const float2 *gr = spec_r + tile_at;

    float4 q[C::QUADS];
    // piece n of thread i: TC = 4: row (i + n THREADS) / 2, columns 2 (i % 2), +1;  TC = 2: row i + n THREADS
    auto fetch = [&](const float2 *__restrict__ src) {
#pragma unroll
        for (int n = 0; n < C::QUADS; ++n) {
            const int i = threadIdx.x + n * C::THREADS;
            const int y = TC == 4 ? i >> 1 : i, cp = TC == 4 ? (i & 1) * 2 : 0;
            q[n] = *reinterpret_cast<const float4 *>(src + y * 4 + cp);
        }
    };
    auto scatter = [&]() {
#pragma unroll
        for (int n = 0; n < C::QUADS; ++n) {
            const int i = threadIdx.x + n * C::THREADS;
            const int y = TC == 4 ? i >> 1 : i, cp = TC == 4 ? (i & 1) * 2 : 0;
            wide_lds[cp * C::COL + P::idx(y)] = make_float2(q[n].x, q[n].y);
            wide_lds[(cp + 1) * C::COL + P::idx(y)] = make_float2(q[n].z, q[n].w);
        }
    };
    fetch(gt);
    scatter();
    fetch(gr);
    __syncthreads();
    float2 tv[16], v[16];
    const int i1 = P::base1(tt);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        tv[m] = x[i1 + P::off1(m)];
    wide_forward<LOG2H>(tv, x, w, tt);
    __syncthreads();
    scatter();
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = x[i1 + P::off1(m)];
    wide_forward<LOG2H>(v, x, w, tt);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        v[i] = cmul_conj(v[i], tv[i]);             // conj(T) * R, position by position
    // the arrays were centred only approximately: removing their exact means is the same
    // as clearing the DC bin (position 0 of spectrum column 0)
    if (!raw && g == 0 && half == 0 && threadIdx.x == 0)
        v[0] = make_float2(0.f, 0.f);
    wide_inverse<LOG2H>(v, x, w, tt);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        x[i1 + P::off1(m)] = v[m];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < C::QUADS; ++n) {
        const int i = threadIdx.x + n * C::THREADS;
        const int y = TC == 4 ? i >> 1 : i, cp = TC == 4 ? (i & 1) * 2 : 0;
        const float2 a = wide_lds[cp * C::COL + P::idx(y)], b = wide_lds[(cp + 1) * C::COL + P::idx(y)];
        *reinterpret_cast<float4 *>(gt + y * 4 + cp) = make_float4(a.x, a.y, b.x, b.y);
    }
    // One workgroup per pair also adds pass 1's partial sums into the pair's statistics, which
    // pass 3 needs (fixed order: the result does not depend on scheduling).
    if (!raw && g == 0 && half == 0) {
        double *red = reinterpret_cast<double *>(wide_lds);
        double sa = 0, sb = 0, sc = 0, sd = 0;
        for (int i = threadIdx.x; i < sums_per_pair; i += C::THREADS) {
            const double4 p = reinterpret_cast<const double4 *>(row_sums)[(int64_t)pair * sums_per_pair + i];
            sa += p.x;
            sb += p.y;
            sc += p.z;
            sd += p.w;
        }
        sa = block_sum(sa, red);        // starts with a barrier: the tile reads above are done
        sb = block_sum(sb, red);
        sc = block_sum(sc, red);
        sd = block_sum(sd, red);
        if (threadIdx.x == 0) {
            stats[pair].t_sum = sa;
            stats[pair].t_sq = sb;
            stats[pair].r_sum = sc;
            stats[pair].r_sq = sd;
        }
    }
}

// pass 3: a workgroup takes 2 RG rows of P' at a time: all threads load them (two rows
// = one complex sequence a + i b with its Hermitian completion) into the transform
// buffers, every transform's threads run one inverse transform, whose real / imaginary
// parts are two output rows.  raw != 0: scaled by 1 / (H W) only.
template <int LOG2W, typename TOut>
__global__ void __launch_bounds__(WideRows<LOG2W>::THREADS)
wide_rows_inverse_kernel(const float2 *__restrict__ spec, int th, int tw_, int H, int hp,
                         const FftStats *__restrict__ stats, double min_std, double min_mean_diff,
                         const float2 *__restrict__ table, TOut *__restrict__ out, int rows_per_wg, int raw)
{
    using R = WideRows<LOG2W>;
    using P = Wide<LOG2W>;
    constexpr int W = P::N, T = P::T, RG = R::RG;
    extern __shared__ __align__(16) float2 wide_lds[];
    const int tr = threadIdx.x / T, tt = threadIdx.x % T;
    const int y_block = blockIdx.x * rows_per_wg, pair = blockIdx.y;
    float2 *x = wide_lds + tr * P::PITCH;
    float scale = (float)(1.0 / ((double)H * (double)W));
    TOut *obase = out + (int64_t)pair * H * W;
    if (!raw) {
        const FftStats st = stats[pair];
        const double tn = (double)th * tw_, rn = (double)H * W;
        double tmean, rmean, tvar, rvar;
        moments(st.t_sum, st.t_sq, tn, (double)st.t_centre, tmean, tvar);
        moments(st.r_sum, st.r_sq, rn, (double)st.r_centre, rmean, rvar);
        const double tstd = sqrt(fmax(tvar, 0.0));
        const double rstd = rvar > 1e-14 * (st.r_sq / rn + 1e-300) ? sqrt(rvar) : 0.0;
        if (tstd < min_std || rstd == 0.0) {        // the whole workgroup: one pair
            const TOut fill = tstd < min_std ? (fabs(tmean - rmean) < min_mean_diff ? TOut(1) : TOut(0))
                                             : TOut(NAN);
            for (int64_t i = threadIdx.x; i < (int64_t)rows_per_wg * W; i += R::THREADS)
                obase[(int64_t)y_block * W + i] = fill;
            return;
        }
        scale = (float)(1.0 / ((double)H * (double)W) / (tstd * tn) / rstd);
    }
    WideTw w;
    wide_load_twiddles<LOG2W>(w, table, tt);
    const float2 *pbase = spec + (int64_t)pair * H * hp;
    for (int y_group = y_block; y_group < y_block + rows_per_wg; y_group += 2 * RG) {
        for (int item = threadIdx.x; item < R::GROUPS * 2 * RG; item += R::THREADS) {
            const int half = item & 1, row = (item >> 1) % RG, g = item / (2 * RG), k0 = 4 * g + 2 * half;
            if (k0 > W / 2)
                continue;
            float2 *xs = wide_lds + row * P::PITCH;
            const float2 *pa = pbase + wide_spec_index(H, y_group + 2 * row, k0);
            float2 a[2], b[2];
            if (k0 + 1 <= W / 2) {
                const float4 qa = *reinterpret_cast<const float4 *>(pa);
                const float4 qb = *reinterpret_cast<const float4 *>(pa + 4);
                a[0] = make_float2(qa.x, qa.y);
                a[1] = make_float2(qa.z, qa.w);
                b[0] = make_float2(qb.x, qb.y);
                b[1] = make_float2(qb.z, qb.w);
            } else {
                a[0] = a[1] = pa[0];
                b[0] = b[1] = pa[4];
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = k0 + c;
                if (k > W / 2)
                    break;
                xs[P::idx(P::pos(k))] = make_float2(a[c].x - b[c].y, a[c].y + b[c].x);           // a + i b
                if (k > 0 && k < W / 2)     // Hermitian completion: conj(a) + i conj(b)
                    xs[P::idx(P::pos(W - k))] = make_float2(a[c].x + b[c].y, b[c].x - a[c].y);
            }
        }
        __syncthreads();
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = x[P::out_index(tt, i)];
        wide_inverse<LOG2W>(v, x, w, tt);
        TOut *o0 = obase + (int64_t)(y_group + 2 * tr) * W;
        TOut *o1 = o0 + W;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            o0[tt + T * m] = (TOut)(v[m].x * scale);
            o1[tt + T * m] = (TOut)(v[m].y * scale);
        }
        __syncthreads();    // the next rows reuse the transform buffers
    }
}

bool wide_size(int n) { return n >= 256 && n <= 4096 && (n & (n - 1)) == 0; }
int wide_row_parts(int W) { return W > 1024 ? W / 1024 : 1; }

namespace {
template <typename K>
hipError_t allow_lds(K kernel, size_t bytes)
{
    return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                             : hipSuccess;
}

// Workgroups take two groups of rows each (the second group's loads are in flight while
// the first is transformed) when that leaves enough workgroups to fill the chip; measured
// on batches of 512 x 512 and 1024 x 1024 pairs and on single 2048 x 2048 pairs, where
// one, two and four groups per workgroup lie within 5 % of each other.
int wide_steps(int64_t total_steps, int64_t enough) { return total_steps >= enough ? 2 : 1; }

template <int LOG2W>
int wide_rows_forward(const float *tmpl, int th, int tw, const float *region, int H, int batch,
                      const FftStats *stats, double *row_sums, const float2 *table, float2 *spec_t,
                      float2 *spec_r, int hp, int raw, hipStream_t st)
{
    using R = WideRows<LOG2W>;
    const int steps = wide_steps((int64_t)batch * (H / R::RG), 512);
    SV_HIP(allow_lds(&wide_rows_forward_kernel<LOG2W>, R::LDS_BYTES));
    hipLaunchKernelGGL(wide_rows_forward_kernel<LOG2W>, dim3(H / (R::RG * steps), batch), dim3(R::THREADS),
                       R::LDS_BYTES, st, tmpl, th, tw, region, H, stats, row_sums, table, spec_t, spec_r, hp,
                       R::RG * steps, raw);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

template <int LOG2H>
int wide_cols(float2 *spec_t, const float2 *spec_r, int W, int hp, int batch, const float2 *table, int raw,
              const double *row_sums, int sums_per_pair, FftStats *stats, hipStream_t st)
{
    using C = WideCols<LOG2H>;
    const int groups = W / 8 + 1;
    const int blocks = C::TC == 4 ? groups : (groups + 7) / 8 * 16;
    SV_HIP(allow_lds(&wide_cols_fused_kernel<LOG2H>, C::LDS_BYTES));
    hipLaunchKernelGGL(wide_cols_fused_kernel<LOG2H>, dim3(blocks, batch), dim3(C::THREADS), C::LDS_BYTES, st,
                       spec_t, spec_r, groups, hp, table, raw, row_sums, sums_per_pair, stats);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

template <int LOG2W, typename TOut>
int wide_rows_inverse(const float2 *spec, int th, int tw, int H, int hp, int batch, const FftStats *stats,
                      double min_std, double min_mean_diff, const float2 *table, TOut *out, int raw,
                      hipStream_t st)
{
    using R = WideRows<LOG2W>;
    const int steps = wide_steps((int64_t)batch * (H / (2 * R::RG)), 1024);
    SV_HIP(allow_lds(&wide_rows_inverse_kernel<LOG2W, TOut>, R::LDS_BYTES));
    hipLaunchKernelGGL((wide_rows_inverse_kernel<LOG2W, TOut>), dim3(H / (2 * R::RG * steps), batch),
                       dim3(R::THREADS), R::LDS_BYTES, st, spec, th, tw, H, hp, stats, min_std, min_mean_diff,
                       table, out, 2 * R::RG * steps, raw);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

#define SV_WIDE_SWITCH(log2n, CALL)                      \
    switch (log2n) {                                     \
    case 8: return CALL(8);                              \
    case 9: return CALL(9);                              \
    case 10: return CALL(10);                            \
    case 11: return CALL(11);                            \
    default: return CALL(12);                            \
    }

}  // namespace

static int wide_rows_forward_any(int log2w, const float *tmpl, int th, int tw, const float *region, int H, int batch,
                          const FftStats *stats, double *row_sums, const float2 *table, float2 *spec_t,
                          float2 *spec_r, int hp, int raw, hipStream_t st)
{
#define SV_CALL(L) wide_rows_forward<L>(tmpl, th, tw, region, H, batch, stats, row_sums, table, spec_t, spec_r, hp, raw, st)
    SV_WIDE_SWITCH(log2w, SV_CALL)
#undef SV_CALL
}
static int wide_cols_any(int log2h, float2 *spec_t, const float2 *spec_r, int W, int hp, int batch, const float2 *table,
                  int raw, const double *row_sums, int sums_per_pair, FftStats *stats, hipStream_t st)
{
#define SV_CALL(L) wide_cols<L>(spec_t, spec_r, W, hp, batch, table, raw, row_sums, sums_per_pair, stats, st)
    SV_WIDE_SWITCH(log2h, SV_CALL)
#undef SV_CALL
}
namespace {
template <typename TOut>
int wide_rows_inverse_typed(int log2w, const float2 *spec, int th, int tw, int H, int hp, int batch,
                            const FftStats *stats, double min_std, double min_mean_diff, const float2 *table,
                            TOut *out, int raw, hipStream_t st)
{
#define SV_CALL(L) wide_rows_inverse<L, TOut>(spec, th, tw, H, hp, batch, stats, min_std, min_mean_diff, table, out, raw, st)
    SV_WIDE_SWITCH(log2w, SV_CALL)
#undef SV_CALL
}
}  // namespace

static int wide_rows_inverse_any(int log2w, const float2 *spec, int th, int tw, int H, int hp, int batch,
                                 const FftStats *stats, double min_std, double min_mean_diff, const float2 *table,
                                 void *out, int out_dtype, int raw, hipStream_t st)
{
    if (out_dtype == SVAMD_F32)
        return wide_rows_inverse_typed<float>(log2w, spec, th, tw, H, hp, batch, stats, min_std, min_mean_diff,
                                              table, static_cast<float *>(out), raw, st);
    return wide_rows_inverse_typed<double>(log2w, spec, th, tw, H, hp, batch, stats, min_std, min_mean_diff,
                                           table, static_cast<double *>(out), raw, st);
}

// The whole correlation: tables and centres, rows forward, columns (forward, product,
// inverse, and the pairs' statistics), rows inverse - four launches.
int wide_correlate(const float *tmpl, int th, int tw, const float *region, int H, int W, int batch, double min_std,
                   double min_mean_diff, void *out, int out_dtype, FftStats *stats, float2 *table_w,
                   float2 *table_h, float2 *spec_t, float2 *spec_r, double *row_sums, int hp, int raw,
                   hipStream_t st)
{
    const int log2w = ilog2(W), log2h = ilog2(H);
    const int pairs = raw ? 0 : batch;
    hipLaunchKernelGGL(wide_setup_kernel, dim3(pairs + (W + H + 1023) / 1024), dim3(1024), 0, st, tmpl, th, tw,
                       region, H, W, stats, pairs, table_w, table_h);
    SV_LAUNCH_CHECK();
    int rc = wide_rows_forward_any(log2w, tmpl, th, tw, region, H, batch, stats, row_sums, table_w, spec_t, spec_r,
                                   hp, raw, st);
    if (rc != SVAMD_OK)
        return rc;
    rc = wide_cols_any(log2h, spec_t, spec_r, W, hp, batch, table_h, raw, row_sums, H * wide_row_parts(W), stats,
                       st);
    if (rc != SVAMD_OK)
        return rc;
    return wide_rows_inverse_any(log2w, spec_t, th, tw, H, hp, batch, stats, min_std, min_mean_diff, table_w, out,
                                 out_dtype, raw, st);
}

}  // namespace svamd
