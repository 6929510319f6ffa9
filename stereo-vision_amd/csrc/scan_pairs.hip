// The inner loop of the stereo scan: for many (template window, search region)
// pairs at once, the globally normalised circular cross-correlation surface of
// the pair and its maximum, followed by the per-window selection rule.
//
// Per pair: the float64 sums of both boxes (two passes: mean, then squared
// deviations), the mean-removed template and a wrapped copy of the mean-removed
// region staged in LDS, and the surface evaluated directly,
//     c[k][l] = sum_{m<mh, n<mw} t'[m][n] * r'[(m + k) mod rh][(n + l) mod rw],
// which is what conj(FFT(t')) * FFT(r') transforms back to.  Statistics, scaling
// and the max-reduction are float64.  The multiply-accumulate runs in float64
// (scan_pairs_kernel: one workgroup per pair) or in float32 with short runs folded
// into float64: on the matrix cores, the sum read as a matrix product
// (scan_pairs_mfma16_kernel: four pairs of sides up to 16 per wavefront;
// scan_pairs_mfma_kernel: one pair of sides up to 32 per wavefront;
// scan_pairs_mfma_wide_kernel: one pair of sides up to 64 per workgroup), or with packed
// FMAs on the vector ALU (scan_pairs_f32_kernel) for the sides whose surfaces would
// leave most of a matrix tile empty.  launch_pairs() holds the choice per side.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace svamd {

struct PairDesc {
    int64_t t_off, t_pitch, t_h, t_w, r_off, r_pitch, r_h, r_w;
};

struct BestNan {
    double best;   // maximum over non-NaN values (-inf if none)
    int arg;       // its first index (INT_MAX if none)
    int has_nan;
};

__device__ __forceinline__ BestNan combine(BestNan a, BestNan b)
{
    BestNan r;
    r.has_nan = a.has_nan | b.has_nan;
    bool take_b = b.best > a.best || (b.best == a.best && b.arg < a.arg);
    r.best = take_b ? b.best : a.best;
    r.arg = take_b ? b.arg : a.arg;
    return r;
}

// A wavefront's best element through lane-shift operands (the steps of wave_sum<double>;
// a lane without a source combines with itself, which changes nothing); every lane
// receives the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_self(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ BestNan best_shifted(BestNan v)
{
    BestNan n;
    n.best = __hiloint2double(dpp_or_self<CTRL, ROW_MASK>(__double2hiint(v.best)),
                              dpp_or_self<CTRL, ROW_MASK>(__double2loint(v.best)));
    n.arg = dpp_or_self<CTRL, ROW_MASK>(v.arg);
    n.has_nan = dpp_or_self<CTRL, ROW_MASK>(v.has_nan);
    return n;
}
__device__ __forceinline__ BestNan wave_best_dpp(BestNan v)
{
    v = combine(v, best_shifted<0x111, 0xF>(v));
    v = combine(v, best_shifted<0x112, 0xF>(v));
    v = combine(v, best_shifted<0x114, 0xF>(v));
    v = combine(v, best_shifted<0x118, 0xF>(v));
    v = combine(v, best_shifted<0x142, 0xA>(v));
    v = combine(v, best_shifted<0x143, 0xC>(v));
    BestNan r;
    r.best = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.best), 63),
                              __builtin_amdgcn_readlane(__double2loint(v.best), 63));
    r.arg = __builtin_amdgcn_readlane(v.arg, 63);
    r.has_nan = __builtin_amdgcn_readlane(v.has_nan, 63);
    return r;
}

__device__ __forceinline__ BestNan block_best(BestNan v, double *red_d, int *red_i)
{
    v = wave_best_dpp(v);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int nw = blockDim.x / WAVE;
    __syncthreads();
    if (lane == 0) {
        red_d[wid] = v.best;
        red_i[2 * wid] = v.arg;
        red_i[2 * wid + 1] = v.has_nan;
    }
    __syncthreads();
    BestNan r;
    r.best = red_d[0];
    r.arg = red_i[0];
    r.has_nan = red_i[1];
    for (int w = 1; w < nw; ++w) {
        BestNan n;
        n.best = red_d[w];
        n.arg = red_i[2 * w];
        n.has_nan = red_i[2 * w + 1];
        r = combine(r, n);
    }
    return r;
}

constexpr int PAIR_KEEP = 8;        // rows per thread that stay in registers between the two staging passes

// Two sums over the workgroup (up to four wavefronts) at once, in place.
__device__ __forceinline__ void block_sum2(double &a, double &b, double *red)
{
    a = wave_sum(a);
    b = wave_sum(b);
    const int nw = (blockDim.x + WAVE - 1) / WAVE;
    if (nw > 1) {
        const int w = threadIdx.x / WAVE;
        __syncthreads();            // red may still be read from the previous exchange
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            red[2 * w] = a;
            red[2 * w + 1] = b;
        }
        __syncthreads();
        a = red[0];
        b = red[1];
        for (int i = 1; i < nw; ++i) {
            a += red[2 * i];
            b += red[2 * i + 1];
        }
    }
}

// Row pitch (in elements) of the float64 kernel's wrapped region image, for tasks of
// W surface columns.  A lane reads its window 16 bytes at a time and its neighbour's
// window starts 8 W bytes further on, so the lanes of one surface row touch every
// other (W = 4) or every fourth (W = 8) 16-byte slot; the 16-lane groups of a 16-byte
// LDS read ({0-3, 12-15, 20-27}, ...) span several consecutive rows.  W = 4: a pitch of
// 16 mod 32 elements puts the second half of row k + 1 on the banks the first half of
// row k leaves free, and odd rows are stored with the two slots of every 32 bytes
// exchanged.  W = 8: a pitch of 2 mod 4 elements moves each of a group's four rows
// onto its own quarter of the slots.  Either way a group covers all 64 banks.
template <int W>
__host__ __device__ __forceinline__ int pair64_row_pitch(int rw, int mwp)
{
    const int c = ((rw + W - 1) / W * W + mwp + 3) & ~3;       // columns the window reads reach
    return W == 4 ? ((c + 15) & ~31) + 16 : c + 2;
}

// One step of the float64 task loop: four template columns against the lane's window
// of W + 3 region values.  The window lives in a ring of W + 4 registers whose base
// B advances by 4 per step (whole rotations are unrolled by the caller).
template <int W, int B>
__device__ __forceinline__ void pair64_step(double (&part)[W], double (&win)[W + 4], const double *__restrict__ trow,
                                            const double *__restrict__ rrow, int n0, int sw)
{
    constexpr int R = W + 4;
    double tv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tv[j] = trow[n0 + j];
        win[(B + W + j) % R] = rrow[n0 + W + (j ^ sw)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
            part[j] = fma(tv[i], win[(B + i + j) % R], part[j]);
}

template <typename CT, int W>
__global__ void __launch_bounds__(256)
scan_pairs_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                  const PairDesc *__restrict__ desc, int64_t npairs, int max_side, double min_std,
                  double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                  double *__restrict__ surfaces, const int64_t *__restrict__ surf_off,
                  const int32_t *__restrict__ list, const int32_t *__restrict__ list_count)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double red[17];
    __shared__ double red_d[4];
    __shared__ int red_i[8];   // up to 4 wavefronts per workgroup
    CT *Tb = reinterpret_cast<CT *>(smem_raw);
    // With a list, the pairs list[0 .. *list_count) are evaluated (a count that an earlier
    // kernel of the stream left on the device) and results land at the pairs' own indices.
    if (list)
        npairs = *list_count;
    // staging walks the boxes in sweeps as wide as the largest side (a power of two)
    const int cw_shift = max_side <= 8 ? 3 : max_side <= 16 ? 4 : max_side <= 32 ? 5 : 6;
    const int lx = threadIdx.x & ((1 << cw_shift) - 1), ly = threadIdx.x >> cw_shift;
    const int ystep = blockDim.x >> cw_shift;

    for (int64_t item = blockIdx.x; item < npairs; item += gridDim.x) {
        const int64_t p = list ? (int64_t)list[item] : item;
        const PairDesc d = desc[p];
        const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
        __syncthreads();  // LDS reuse across loop iterations
        if (rh <= 0 || rw <= 0 || th < 0 || tw < 0 || th > max_side || tw > max_side || rh > max_side ||
            rw > max_side) {
            if (threadIdx.x == 0) {
                out_max[p] = NAN;
                if (out_argmax)
                    out_argmax[p] = -1;
            }
            continue;
        }
        const double *T = tbase + d.t_off;
        const double *R = rbase + d.r_off;
        const int tn = th * tw, rn = rh * rw;

        // LDS images: template rows padded with zeros to a multiple of 4 columns,
        // region rows wrapped horizontally and long enough for every 4-wide read
        const int mh = min(th, rh), mw = min(tw, rw);
        const int mwp = (mw + 3) & ~3;
        static_assert(sizeof(CT) == 8, "the bank layout below is for 8-byte elements");
        const int rxp = pair64_row_pitch<W>(rw, mwp);
        CT *Rx = Tb + mh * mwp;
        for (int i = threadIdx.x; i < rh * rxp; i += blockDim.x)
            Rx[i] = CT(0);
        // Both boxes are walked as sweeps of `ystep` rows: a thread keeps its column lx and
        // visits rows ly, ly + ystep, ...; its first PAIR_KEEP rows stay in registers
        // between the passes, later rows are read again.
        double tk[PAIR_KEEP], rk[PAIR_KEEP];
        const bool tcol = lx < tw, rcol = lx < rw;
        const double *Tl = T + lx, *Rl = R + lx;
        // pass 1: means
        double ts = 0.0, rs = 0.0;
#pragma unroll
        for (int j = 0; j < PAIR_KEEP; ++j) {
            const int y = ly + j * ystep;
            tk[j] = (tcol && y < th) ? Tl[y * d.t_pitch] : 0.0;
            rk[j] = (rcol && y < rh) ? Rl[y * d.r_pitch] : 0.0;
            ts += tk[j];
            rs += rk[j];
        }
        for (int y = ly + PAIR_KEEP * ystep; y < max(th, rh); y += ystep) {
            if (tcol && y < th)
                ts += Tl[y * d.t_pitch];
            if (rcol && y < rh)
                rs += Rl[y * d.r_pitch];
        }
        block_sum2(ts, rs, red);
        const double tmean = ts / (double)tn, rmean = rs / (double)rn;
        // pass 2: squared deviations; the centred values go to LDS on the way (the
        // reduction in between orders them after the zero fill)
        double tq = 0.0, rq = 0.0;
        auto put = [&](int y, double tval, double rval) {
            const double tv = tval - tmean;
            if (tcol && y < th)
                tq += tv * tv;
            if (y < mh && lx < mwp)                     // columns mw .. mwp - 1 are zero padding
                Tb[y * mwp + lx] = lx < mw ? (CT)tv : CT(0);
            if (rcol && y < rh) {
                const double v = rval - rmean;
                rq += v * v;
                const int sw = W == 4 ? (y & 1) << 1 : 0;
                Rx[y * rxp + (lx ^ sw)] = (CT)v;
                if (lx < mw - 1)                        // the horizontal wrap
                    Rx[y * rxp + ((lx + rw) ^ sw)] = (CT)v;
            }
        };
#pragma unroll
        for (int j = 0; j < PAIR_KEEP; ++j)
            put(ly + j * ystep, tk[j], rk[j]);
        for (int y = ly + PAIR_KEEP * ystep; y < max(max(th, rh), mh); y += ystep)
            put(y, (tcol && y < th) ? Tl[y * d.t_pitch] : 0.0, (rcol && y < rh) ? Rl[y * d.r_pitch] : 0.0);
        block_sum2(tq, rq, red);
        const double tstd = sqrt(tq / (double)tn);
        const double rstd = sqrt(rq / (double)rn);
        double *surf = surfaces ? surfaces + surf_off[p] : nullptr;

        if (tstd < min_std) {  // homogeneous template: constant surface
            const double fill = fabs(tmean - rmean) < min_mean_diff ? 1.0 : 0.0;
            if (surf)
                for (int i = threadIdx.x; i < rn; i += blockDim.x)
                    surf[i] = fill;
            if (threadIdx.x == 0) {
                out_max[p] = fill;
                if (out_argmax)
                    out_argmax[p] = 0;
            }
            continue;
        }
        __syncthreads();

        // An empty template transforms to an all-zero spectrum: the surface is 0
        // wherever the normalised region is finite.
        const double scale = tn == 0 ? ((rstd > 0.0) ? 0.0 : NAN) : 1.0 / (tstd * (double)tn) / rstd;

        // One task = W horizontally adjacent surface elements (k, l0 .. l0 + W - 1): the
        // region row is walked with a sliding register window, so every four 16-byte LDS
        // reads (two of template, two of region) feed 4 W FMAs.
        BestNan mine;
        mine.best = -INFINITY;
        mine.arg = 0x7fffffff;
        mine.has_nan = 0;
        const int lgroups = (rw + W - 1) / W;
        for (int task = threadIdx.x; task < rh * lgroups; task += blockDim.x) {
            const int k = task / lgroups, l0 = (task - k * lgroups) * W;
            double acc[W];
#pragma unroll
            for (int j = 0; j < W; ++j)
                acc[j] = 0.0;
            for (int m = 0; m < mh; ++m) {
                int rr = m + k;
                if (rr >= rh)
                    rr -= rh;
                const CT *trow = Tb + m * mwp;
                const CT *rrow = Rx + rr * rxp + l0;
                const int sw = W == 4 ? (rr & 1) << 1 : 0;   // W = 4: odd rows have their slots exchanged in pairs
                double part[W], win[W + 4];
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    part[j] = 0.0;
                    win[j] = rrow[j ^ sw];
                }
                // whole rotations of the window ring in the loop: every register is back in
                // its role at the loop's end
                int n0 = 0;
                if constexpr (W == 4) {
                    for (; n0 + 8 <= mwp; n0 += 8) {
                        pair64_step<W, 0>(part, win, trow, rrow, n0, sw);
                        pair64_step<W, 4>(part, win, trow, rrow, n0 + 4, sw);
                    }
                    if (n0 < mwp)
                        pair64_step<W, 0>(part, win, trow, rrow, n0, sw);
                } else {
                    for (; n0 + 12 <= mwp; n0 += 12) {
                        pair64_step<W, 0>(part, win, trow, rrow, n0, sw);
                        pair64_step<W, 4>(part, win, trow, rrow, n0 + 4, sw);
                        pair64_step<W, 8>(part, win, trow, rrow, n0 + 8, sw);
                    }
                    if (n0 < mwp) {
                        pair64_step<W, 0>(part, win, trow, rrow, n0, sw);
                        if (n0 + 4 < mwp)
                            pair64_step<W, 4>(part, win, trow, rrow, n0 + 4, sw);
                    }
                }
#pragma unroll
                for (int j = 0; j < W; ++j)
                    acc[j] += part[j];
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (l0 + j >= rw)
                    continue;
                const int idx = k * rw + l0 + j;
                const double v = tn == 0 ? scale : acc[j] * scale;
                if (surf)
                    surf[idx] = v;
                if (isnan(v)) {
                    mine.has_nan = 1;
                } else if (v > mine.best) {      // ascending idx within a thread: first maximum kept
                    mine.best = v;
                    mine.arg = idx;
                }
            }
        }
        const BestNan all = block_best(mine, red_d, red_i);
        if (threadIdx.x == 0) {
            out_max[p] = all.has_nan ? NAN : all.best;
            if (out_argmax)
                out_argmax[p] = all.arg == 0x7fffffff ? -1 : all.arg;
        }
    }
}

// ---- float32 multiply-accumulate: one wavefront per group of pairs ---------------------
// The same surface with packed FMAs.  A lane owns a task = two vertically adjacent
// surface rows (k, k + 1) x G adjacent columns; its G accumulators are register
// pairs (row k, row k + 1).  The region is staged as rows of such pairs,
//     S[a][x] = (r'[a][x mod rw], r'[(a + 1) mod rh][x mod rw]),
// so one 16-byte LDS read delivers two columns of both rows and every template
// value t'[m][n] (read four at a time, the same address for the whole pair) is
// issued as one v_pk_fma_f32 per column with t' broadcast to both halves: 4 G
// packed FMAs per three (G = 8) or two (G = 4) 16-byte window reads.  The window
// of G + 4 columns rotates through registers.  Partial sums stay in float32 for
// about 256 products and are then folded into float64.
// Small pairs leave most of a wavefront's lanes without a task, so a wavefront
// takes P = 64 / tasks pairs at a time (each with its own LDS slot).
typedef float pf2 __attribute__((ext_vector_type(2)));
typedef float pf4 __attribute__((ext_vector_type(4)));

template <int TH>
__device__ __forceinline__ void pk_fma_tb(pf2 &acc, const pf2 &tpair, const pf2 &w)
{
    if constexpr (TH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(tpair), "v"(w));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(tpair), "v"(w));
}

struct ColQuad {           // four consecutive columns of a row pair, as the two 16-byte reads left them
    pf4 lo, hi;
    template <int C> __device__ __forceinline__ pf2 col() const
    {
        if constexpr (C == 0)
            return lo.xy;
        else if constexpr (C == 1)
            return lo.zw;
        else if constexpr (C == 2)
            return hi.xy;
        else
            return hi.zw;
    }
};

// sw (0 or 4): the row's 16-byte slots are stored swapped in pairs (see pair_row_swap)
__device__ __forceinline__ ColQuad load_colquad(const float *p, int sw)
{
    ColQuad q;
    q.lo = *reinterpret_cast<const pf4 *>(p + sw);
    q.hi = *reinterpret_cast<const pf4 *>(p + (4 ^ sw));
    return q;
}

template <int G, int I = 0, int J = 0>
__device__ __forceinline__ void pair_group(pf2 (&acc)[G], const pf4 &t, const ColQuad &q0, const ColQuad &q1,
                                           const ColQuad &q2)
{
    if constexpr (I < 4) {
        if constexpr (J < G) {
            constexpr int C = I + J;
            const ColQuad &q = C < 4 ? q0 : (C < 8 ? q1 : q2);
            const pf2 tp = (I & 2) ? pf2(t.zw) : pf2(t.xy);
            pk_fma_tb<(I & 1)>(acc[J], tp, q.template col<(C & 3)>());
            pair_group<G, I, J + 1>(acc, t, q0, q1, q2);
        } else {
            pair_group<G, I + 1, 0>(acc, t, q0, q1, q2);
        }
    }
}

// One template row (groups * 4 columns, zero padded) against one row pair of S,
// starting at the task's first column.
template <int G>
__device__ __forceinline__ void pair_row(pf2 (&acc)[G], const float *__restrict__ trow,
                                         const float *__restrict__ srow, int sw, int groups)
{
    // the window's quads rotate through q0, q1, q2 (q0, q1 for G = 4); whole rotations
    // in the loop, so that every register is back in its role at the loop's end
    ColQuad q0 = load_colquad(srow, sw), q1, q2;
    const pf4 *t4 = reinterpret_cast<const pf4 *>(trow);
    int g = 0;
    if constexpr (G == 8) {
        // four quads in rotation: the quad and the template values of the NEXT group are
        // requested before the current group's FMAs, so one wavefront alone hides the
        // LDS latency
        ColQuad q3;
        q1 = load_colquad(srow + 8, sw);
        q2 = load_colquad(srow + 16, sw);
        pf4 ta = t4[0], tb;
        for (; g + 4 <= groups; g += 4) {
            q3 = load_colquad(srow + 8 * (g + 3), sw);
            tb = t4[g + 1];
            pair_group<G>(acc, ta, q0, q1, q2);
            __builtin_amdgcn_sched_barrier(0);
            q0 = load_colquad(srow + 8 * (g + 4), sw);
            ta = t4[g + 2];
            pair_group<G>(acc, tb, q1, q2, q3);
            __builtin_amdgcn_sched_barrier(0);
            q1 = load_colquad(srow + 8 * (g + 5), sw);
            tb = t4[g + 3];
            pair_group<G>(acc, ta, q2, q3, q0);
            __builtin_amdgcn_sched_barrier(0);
            q2 = load_colquad(srow + 8 * (g + 6), sw);
            ta = t4[g + 4];
            pair_group<G>(acc, tb, q3, q0, q1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < groups) {
            q3 = load_colquad(srow + 8 * (g + 3), sw);
            tb = t4[g + 1];
            pair_group<G>(acc, ta, q0, q1, q2);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < groups) {
                q0 = load_colquad(srow + 8 * (g + 4), sw);
                ta = t4[g + 2];
                pair_group<G>(acc, tb, q1, q2, q3);
            __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < groups)
                    pair_group<G>(acc, ta, q2, q3, q0);
            __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // G = 4: two quads per group, three in rotation (one requested ahead)
        q1 = load_colquad(srow + 8, sw);
        pf4 ta = t4[0], tb, tc;
        for (; g + 3 <= groups; g += 3) {
            q2 = load_colquad(srow + 8 * (g + 2), sw);
            tb = t4[g + 1];
            pair_group<G>(acc, ta, q0, q1, q1);
            __builtin_amdgcn_sched_barrier(0);
            q0 = load_colquad(srow + 8 * (g + 3), sw);
            tc = t4[g + 2];
            pair_group<G>(acc, tb, q1, q2, q2);
            __builtin_amdgcn_sched_barrier(0);
            q1 = load_colquad(srow + 8 * (g + 4), sw);
            ta = t4[g + 3];
            pair_group<G>(acc, tc, q2, q0, q0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < groups) {
            q2 = load_colquad(srow + 8 * (g + 2), sw);
            tb = t4[g + 1];
            pair_group<G>(acc, ta, q0, q1, q1);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < groups)
                pair_group<G>(acc, tb, q1, q2, q2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Columns of the row-pair image S that hold data: the last task starts at column
// gcols - G, walks mwp template columns and its window reads run G + 4 columns ahead
// (the quads requested in advance), i.e. up to column gcols + mwp + 3.
__host__ __device__ __forceinline__ int pair_row_cols(int gcols, int mwp)
{
    return (gcols + mwp + 4 + 3) & ~3;
}

// LDS banks.  A lane reads its window 16 bytes at a time; its neighbour's window
// starts 8 G bytes further on, so the lanes of one surface row pair touch only every
// other (G = 4) or every fourth (G = 8) slot of four banks, and the other lanes of
// a 16-lane read group (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32) sit on row pairs 2, 4, ... rows further down.  Two measures spread a
// group over all 64 banks for surfaces about 32 columns wide (8 or 4 lanes per row
// pair): the row pitch (in columns of 8 bytes) is 8 mod 16 for G = 4, which puts the
// second half of row pair k + 1 onto the banks that the first half of row pair k
// leaves free, and 2 mod 4 for G = 8; and rows whose index has bit 1 (G = 4) or
// bit 2 (G = 8) set are stored with the two 16-byte slots of every 32 bytes
// exchanged.  pair_row_swap returns the float offset to XOR into an in-row offset.
template <int G>
__host__ __device__ __forceinline__ int pair_row_pitch(int cols)
{
    return G == 4 ? ((cols + 7) & ~15) + 8 : cols + 2;
}

template <int G>
__device__ __forceinline__ int pair_row_swap(int a)
{
    return ((a >> (G == 4 ? 1 : 2)) & 1) << 2;
}

// The float64 sum over each row of 16 lanes, delivered to all 16 with the same bits:
// a butterfly of lane exchanges (neighbours, pairs, mirrored halves of 8, mirrored row)
// in which both partners of an exchange add the same two numbers.
template <int CTRL>
__device__ __forceinline__ double dpp_exchanged(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_exchanged<0xB1>(v);      // quad_perm [1, 0, 3, 2]
    v += dpp_exchanged<0x4E>(v);      // quad_perm [2, 3, 0, 1]
    v += dpp_exchanged<0x141>(v);     // row_half_mirror
    v += dpp_exchanged<0x140>(v);     // row_mirror
    return v;
}

struct SlotInfo {
    double tmean, rmean, tstd, rstd;
    int th, tw, rh, rw;
    int state;        // 0: evaluate; 1: constant surface (low-deviation template); 2: invalid -> NaN; 3: no pair
    int pad;
};

constexpr int PAIR_MAX_SLOTS = 16;
constexpr int PAIR_FLUSH = 256;     // float32 products per accumulator between float64 folds

__device__ __forceinline__ BestNan wave_best(BestNan v) { return wave_best_dpp(v); }

// Two sums over the workgroup at once (in place): one wavefront needs no LDS round
// trip, two or four wavefronts exchange their parts through red[0..7].
__device__ __forceinline__ void pair_sum2(double &a, double &b, double *red, int nthr)
{
    a = wave_sum(a);
    b = wave_sum(b);
    if (nthr > WAVE) {
        const int w = threadIdx.x / WAVE;
        __syncthreads();            // red may still be read from the previous exchange
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            red[2 * w] = a;
            red[2 * w + 1] = b;
        }
        __syncthreads();
        a = red[0] + red[2];
        b = red[1] + red[3];
        if (nthr > 2 * WAVE) {
            a += red[4];
            b += red[5];
        }
        if (nthr > 3 * WAVE) {
            a += red[6];
            b += red[7];
        }
    }
}

__device__ __forceinline__ bool pair_desc_ok(const PairDesc &d, int max_side)
{
    return !(d.r_h <= 0 || d.r_w <= 0 || d.t_h < 0 || d.t_w < 0 || d.t_h > max_side || d.t_w > max_side ||
             d.r_h > max_side || d.r_w > max_side);
}

// A lane's first PAIR_KEEP rows of both boxes (column lx, rows ly, ly + ystep, ...),
// requested but not yet used: the caller decides when to wait for them.
__device__ __forceinline__ void pair_request_rows(const double *__restrict__ tbase, const double *__restrict__ rbase,
                                                  const PairDesc &d, bool ok, int lx, int ly, int ystep,
                                                  double (&tk)[PAIR_KEEP], double (&rk)[PAIR_KEEP])
{
    const bool tcol = ok && lx < (int)d.t_w, rcol = ok && lx < (int)d.r_w;
    const double *Tl = tbase + d.t_off + lx, *Rl = rbase + d.r_off + lx;
#pragma unroll
    for (int j = 0; j < PAIR_KEEP; ++j) {
        const int y = ly + j * ystep;
        tk[j] = (tcol && y < (int)d.t_h) ? Tl[y * d.t_pitch] : 0.0;
        rk[j] = (rcol && y < (int)d.r_h) ? Rl[y * d.r_pitch] : 0.0;
    }
}

#ifdef SV_PAIR_STAMPS
__device__ unsigned long long pair_stamps[8];
#define PAIR_STAMP(slot) do { const unsigned long long now_ = clock64(); stamp_acc_[slot] += now_ - stamp_; stamp_ = now_; } while (0)
#else
#define PAIR_STAMP(slot) do { } while (0)
#endif

// SHARED: several pairs (sides up to 16) share one wavefront and are staged by rows of 16
// lanes; an instantiation of its own, so that its staging registers do not count against
// the occupancy of the kernel for larger pairs.
template <int G, bool SHARED>
__global__ void __launch_bounds__(4 * WAVE)
scan_pairs_f32_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                      const PairDesc *__restrict__ desc, int64_t npairs, int max_side, int slots,
                      int tasks_per_slot, int slot_floats, int t_floats, int s_floats, double min_std,
                      double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                      double *__restrict__ surfaces, const int64_t *__restrict__ surf_off)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ SlotInfo info[PAIR_MAX_SLOTS];
    __shared__ double red[17];
    __shared__ double red_d[4];
    __shared__ int red_i[8];
    float *lds = reinterpret_cast<float *>(smem_raw);
    // 64 threads when several pairs share the wavefront, 128 or 256 (two or four
    // wavefronts on one pair) otherwise; `lane` is the thread's index in the workgroup
    const int lane = threadIdx.x, nthr = blockDim.x;
    const int tstride = (max_side + 3) & ~3;     // template row stride in every slot
    // staging walks the arrays in sweeps as wide as the largest side (a power of two)
    const int cw_shift = max_side <= 8 ? 3 : max_side <= 16 ? 4 : max_side <= 32 ? 5 : 6;
    const int lx = lane & ((1 << cw_shift) - 1), ly = lane >> cw_shift, ystep = nthr >> cw_shift;
#ifdef SV_PAIR_STAMPS
    unsigned long long stamp_acc_[8] = {};
#endif

    for (int64_t base = (int64_t)blockIdx.x * slots; base < npairs; base += (int64_t)gridDim.x * slots) {
        __syncthreads();   // LDS reuse across iterations
        // The staging phases issue few instructions between long waits; given priority
        // over the wavefronts that are in their FMA phase they get through quickly
        // (oldest-first issue would otherwise starve a young wavefront's staging for as
        // long as an older one's FMA phase lasts), and more wavefronts compute at a time.
        __builtin_amdgcn_s_setprio(3);
#ifdef SV_PAIR_STAMPS
        unsigned long long stamp_ = clock64();
#endif
        if constexpr (SHARED) {
            // ---- statistics and staging of pairs that share the wavefront: four pairs at a
            // time, one per row of 16 lanes, a lane owning one column of its pair's template
            // and then of its region with the column's pixels in registers (sides are 16 at
            // most here).  Sums run down the column and then across the row of lanes.
            // Everything a lane can reach is zero before the staging lands on it: pairs walk
            // the wave-uniform row and column-group counts of the largest pair in the
            // wavefront, past their own columns and rows, against template zeros - and
            // 0 * NaN is NaN, so no bit patterns of earlier pairs may be left.  (One
            // wavefront: its LDS operations complete in order.)
            {
                float4 *z = reinterpret_cast<float4 *>(lds);
                const int n4 = slots * slot_floats / 4;
                for (int i = lane; i < n4; i += WAVE)
                    z[i] = float4{0.f, 0.f, 0.f, 0.f};
            }
            const int x = lane & 15, row16 = lane >> 4;
            for (int s0 = 0; s0 < slots; s0 += 4) {
                const int s = s0 + row16;
                const int64_t p = base + s;
                const bool present = s < slots && p < npairs;
                PairDesc d;
                d.t_off = d.r_off = 0;
                d.t_pitch = d.r_pitch = 0;
                d.t_h = d.t_w = d.r_h = d.r_w = 0;
                if (present)
                    d = desc[p];
                const bool ok = present && pair_desc_ok(d, max_side);
                const int th = ok ? (int)d.t_h : 0, tw = ok ? (int)d.t_w : 0;
                const int rh = ok ? (int)d.r_h : 0, rw = ok ? (int)d.r_w : 0;
                const double *Tl = tbase + d.t_off + x, *Rl = rbase + d.r_off + x;
                double tc[16], rc[16];
#pragma unroll
                for (int y = 0; y < 16; ++y) {
                    tc[y] = (x < tw && y < th) ? Tl[y * d.t_pitch] : 0.0;
                    rc[y] = (x < rw && y < rh) ? Rl[y * d.r_pitch] : 0.0;
                }
                double ts = 0.0, rs = 0.0;
#pragma unroll
                for (int y = 0; y < 16; ++y) {
                    ts += tc[y];
                    rs += rc[y];
                }
                const double tn = (double)(th * tw), rn = (double)(rh * rw);
                const double tmean = row16_sum(ts) / tn, rmean = row16_sum(rs) / rn;
                double tq = 0.0, rq = 0.0;
#pragma unroll
                for (int y = 0; y < 16; ++y) {
                    tc[y] = (x < tw && y < th) ? tc[y] - tmean : 0.0;
                    rc[y] = (x < rw && y < rh) ? rc[y] - rmean : 0.0;
                    tq += tc[y] * tc[y];
                    rq += rc[y] * rc[y];
                }
                const double tstd = sqrt(row16_sum(tq) / tn), rstd = sqrt(row16_sum(rq) / rn);
                const bool flat = tstd < min_std;
                if (x == 0 && s < slots) {
                    info[s].tmean = tmean;
                    info[s].rmean = rmean;
                    info[s].tstd = tstd;
                    info[s].rstd = rstd;
                    info[s].th = th;
                    info[s].tw = tw;
                    info[s].rh = rh;
                    info[s].rw = rw;
                    info[s].state = !present ? 3 : !ok ? 2 : flat ? 1 : 0;
                }
                if (!ok || flat)
                    continue;
                float *Ts = lds + (size_t)s * slot_floats, *Ss = Ts + t_floats;
                const int mh = min(th, rh), mw = min(tw, rw);
                const int mwp = (mw + 3) & ~3;
                const int gcols = (rw + G - 1) / G * G;
                const int rcols = pair_row_cols(gcols, mwp), rxp = pair_row_pitch<G>(rcols);
                if (x < mw) {
#pragma unroll
                    for (int y = 0; y < 16; ++y)
                        if (y < mh)
                            Ts[y * tstride + x] = (float)tc[y];
                }
                // S[a][x'] = (r'[a][x' mod rw], r'[(a + 1) mod rh][x' mod rw]) for x' = x, x + rw, ...
                if (x < rw) {
#pragma unroll
                    for (int a = 0; a < 16; ++a) {
                        if (a < rh) {
                            const float cur = (float)rc[a];
                            const float nxt = (float)((a + 1 < rh) ? rc[a + 1 < 16 ? a + 1 : 0] : rc[0]);
                            const int sw = pair_row_swap<G>(a);
                            for (int xx = x; xx < rcols; xx += rw)
                                *reinterpret_cast<pf2 *>(Ss + 2 * a * rxp + ((2 * xx) ^ sw)) = pf2{cur, nxt};
                        }
                    }
                }
            }
        } else {
        // ---- statistics and staging, one slot after the other, all lanes ----
        for (int s = 0; s < slots; ++s) {
            const int64_t p = base + s;
            if (p >= npairs) {
                if (lane == 0)
                    info[s].state = 3;
                continue;
            }
            const PairDesc d = desc[p];
            const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
            if (!pair_desc_ok(d, max_side)) {
                if (lane == 0)
                    info[s].state = 2;
                continue;
            }
            PAIR_STAMP(6);      // descriptor read
            const double *T = tbase + d.t_off;
            const double *R = rbase + d.r_off;
            const int tn = th * tw, rn = rh * rw;
            const int mh = min(th, rh), mw = min(tw, rw);
            const int mwp = (mw + 3) & ~3;
            const int gcols = (rw + G - 1) / G * G;
            const int rcols = pair_row_cols(gcols, mwp), rxp = pair_row_pitch<G>(rcols);
            float *Ts = lds + (size_t)s * slot_floats, *Ss = Ts + t_floats, *Ps = Ss + s_floats;
            // The template slot is zero beyond the pair's own rows and columns: every lane
            // of the wavefront walks the same number of rows and column groups.
            for (int i = lane; i < t_floats; i += nthr)
                Ts[i] = 0.0f;
            // Pairs that share a wavefront walk the wave-uniform row and column-group
            // counts of the largest of them: a narrower pair's lanes read past its own
            // row pitch and, on its last row, into rows of the slot it never stages.
            // Those products meet template zeros, but 0 * NaN is NaN, so nothing a lane
            // can reach may keep the bit patterns of whatever used the LDS before.  (One
            // wavefront: its LDS operations complete in order, the staging below lands on
            // top of this fill.)
            if (slots > 1) {
                float4 *Sz = reinterpret_cast<float4 *>(Ss);
                for (int i = lane; i < s_floats / 4; i += nthr)
                    Sz[i] = float4{0.f, 0.f, 0.f, 0.f};
            }
            PAIR_STAMP(7);      // template slot zeroed
            // The arrays are walked as sweeps of `ystep` rows: a lane keeps its column lx and
            // visits rows ly, ly + ystep, ...  The first PAIR_KEEP rows of a lane stay in
            // registers between the two passes (all of them for sides up to 32 with two
            // wavefronts); later rows are read again.
            double tk[PAIR_KEEP], rk[PAIR_KEEP];
            const bool tcol = lx < tw, rcol = lx < rw;
            const double *Tl = T + lx, *Rl = R + lx;
            pair_request_rows(tbase, rbase, d, true, lx, ly, ystep, tk, rk);
            // pass 1: means
            double ts = 0.0, rs = 0.0;
#pragma unroll
            for (int j = 0; j < PAIR_KEEP; ++j) {
                ts += tk[j];
                rs += rk[j];
            }
            for (int y = ly + PAIR_KEEP * ystep; y < max(th, rh); y += ystep) {
                if (tcol && y < th)
                    ts += Tl[y * d.t_pitch];
                if (rcol && y < rh)
                    rs += Rl[y * d.r_pitch];
            }
            PAIR_STAMP(0);      // first pass: rows arrived and summed
            pair_sum2(ts, rs, red, nthr);
            PAIR_STAMP(1);      // loads arrived, means reduced
            const double tmean = ts / (double)tn, rmean = rs / (double)rn;
            // pass 2: squared deviations; the centred values go to LDS on the way (the
            // reduction in between orders them after the zero fill)
            double tq = 0.0, rq = 0.0;
#pragma unroll
            for (int j = 0; j < PAIR_KEEP; ++j) {
                const int y = ly + j * ystep;
                if (tcol && y < th) {
                    const double v = tk[j] - tmean;
                    tq += v * v;
                    if (y < mh && lx < mw)
                        Ts[y * tstride + lx] = (float)v;
                }
                if (rcol && y < rh) {
                    const double v = rk[j] - rmean;
                    rq += v * v;
                    Ps[y * rw + lx] = (float)v;
                }
            }
            for (int y = ly + PAIR_KEEP * ystep; y < max(th, rh); y += ystep) {
                if (tcol && y < th) {
                    const double v = Tl[y * d.t_pitch] - tmean;
                    tq += v * v;
                    if (y < mh && lx < mw)
                        Ts[y * tstride + lx] = (float)v;
                }
                if (rcol && y < rh) {
                    const double v = Rl[y * d.r_pitch] - rmean;
                    rq += v * v;
                    Ps[y * rw + lx] = (float)v;
                }
            }
            pair_sum2(tq, rq, red, nthr);
            PAIR_STAMP(2);      // deviations, LDS copies
            const double tstd = sqrt(tq / (double)tn), rstd = sqrt(rq / (double)rn);
            const bool flat = tstd < min_std;
            if (lane == 0) {
                info[s].tmean = tmean;
                info[s].rmean = rmean;
                info[s].tstd = tstd;
                info[s].rstd = rstd;
                info[s].th = th;
                info[s].tw = tw;
                info[s].rh = rh;
                info[s].rw = rw;
                info[s].state = flat ? 1 : 0;
            }
            if (flat)
                continue;
            // S[a][x] = (r'[a][x mod rw], r'[(a + 1) mod rh][x mod rw]) from the plain copy
            for (int x = lane; x < rxp; x += nthr) {
                if (x >= rcols) {
                    // the pitch's padding columns: a lane that shares its wavefront with a wider
                    // pair walks past its own columns (against template zeros) and must not meet
                    // whatever bit patterns the LDS held before
                    for (int a = 0; a < rh; ++a)
                        *reinterpret_cast<pf2 *>(Ss + 2 * (a * rxp + x)) = pf2{0.f, 0.f};
                    continue;
                }
                const int xs = x % rw;
                float cur = Ps[xs];
                for (int a = 0; a < rh; ++a) {
                    const float nxt = Ps[(a + 1 < rh ? a + 1 : 0) * rw + xs];
                    *reinterpret_cast<pf2 *>(Ss + 2 * a * rxp + ((2 * x) ^ pair_row_swap<G>(a))) = pf2{cur, nxt};
                    cur = nxt;
                }
            }
        }
        }
        __syncthreads();
        PAIR_STAMP(3);          // row-pair image
        __builtin_amdgcn_s_setprio(0);

        // ---- the surfaces: lane -> (slot, task) ----
        // Loop bounds are made wave-uniform (the largest row / column-group / task
        // count among the wavefront's pairs): lanes of smaller pairs multiply the
        // zeros of their template slot, lanes without a task compute on their
        // slot's first task and drop the result.  No divergent control flow inside.
        const int slot = slots > 1 ? lane / tasks_per_slot : 0;
        const int first_task = slots > 1 ? lane - slot * tasks_per_slot : lane;
        const bool evaluate = slot < slots && info[slot].state == 0;
        SlotInfo si;
        si.th = si.tw = 0;
        si.rh = si.rw = 1;
        si.tstd = si.rstd = si.tmean = si.rmean = 0.0;
        if (evaluate)
            si = info[slot];
        const int mh = min(si.th, si.rh), mw = min(si.tw, si.rw);
        const int mwp = (mw + 3) & ~3;
        const int lgroups = (si.rw + G - 1) / G;
        const int rxp = evaluate ? pair_row_pitch<G>(pair_row_cols(lgroups * G, mwp)) : 0;
        const int ntasks = evaluate ? (si.rh + 1) / 2 * lgroups : 0;
        int u_rows = mh, u_groups = mwp / 4, u_tasks = ntasks;
#pragma unroll
        for (int o = WAVE / 2; o > 0; o >>= 1) {
            u_rows = max(u_rows, __shfl_xor(u_rows, o, WAVE));
            u_groups = max(u_groups, __shfl_xor(u_groups, o, WAVE));
            u_tasks = max(u_tasks, __shfl_xor(u_tasks, o, WAVE));
        }
        u_rows = __builtin_amdgcn_readfirstlane(u_rows);
        u_groups = __builtin_amdgcn_readfirstlane(u_groups);
        u_tasks = __builtin_amdgcn_readfirstlane(u_tasks);
        const int rounds = slots > 1 ? (u_tasks > 0 ? 1 : 0) : (u_tasks + nthr - 1) / nthr;
        const int flush_rows = u_groups * 4 >= PAIR_FLUSH ? 1 : PAIR_FLUSH / max(u_groups * 4, 1);

        BestNan mine;
        mine.best = -INFINITY;
        mine.arg = 0x7fffffff;
        mine.has_nan = 0;
        const int slot_c = slot < slots ? slot : 0;
        const float *Ts = lds + (size_t)slot_c * slot_floats, *Ss = Ts + t_floats;
        const int tn = si.th * si.tw;
        const double scale = tn == 0 ? ((si.rstd > 0.0) ? 0.0 : NAN) : 1.0 / (si.tstd * (double)tn) / si.rstd;
        double *surf = (surfaces && evaluate) ? surfaces + surf_off[base + slot] : nullptr;
        for (int round = 0; round < rounds; ++round) {
            const int task = first_task + round * nthr;
            const bool has_task = evaluate && task < ntasks;
            const int tk = has_task ? task : 0;
            const int kp = tk / lgroups, l0 = (tk - kp * lgroups) * G, k = 2 * kp;
            pf2 acc[G];
            double d0[G], d1[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                acc[j] = pf2{0.f, 0.f};
                d0[j] = d1[j] = 0.0;
            }
            int since = 0, a = k < si.rh ? k : 0;
            const float *srow = Ss + 2 * (a * rxp + l0), *trow = Ts;      // walked, not recomputed, row by row
            for (int m = 0; m < u_rows; ++m) {
                if (u_groups > 0)
                    pair_row<G>(acc, trow, srow, pair_row_swap<G>(a), u_groups);
                trow += tstride;
                const bool wrap = a + 1 == si.rh;
                a = wrap ? 0 : a + 1;
                srow = wrap ? Ss + 2 * l0 : srow + 2 * rxp;
                if (++since == flush_rows || m == u_rows - 1) {
                    since = 0;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        d0[j] += (double)acc[j].x;
                        d1[j] += (double)acc[j].y;
                        acc[j] = pf2{0.f, 0.f};
                    }
                }
            }
            if (!has_task)
                continue;
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                if (k + row >= si.rh)
                    continue;
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (l0 + j >= si.rw)
                        continue;
                    const int idx = (k + row) * si.rw + l0 + j;
                    const double v = tn == 0 ? scale : (row ? d1[j] : d0[j]) * scale;
                    if (surf)
                        surf[idx] = v;
                    if (isnan(v)) {
                        mine.has_nan = 1;
                    } else if (v > mine.best || (v == mine.best && idx < mine.arg)) {
                        mine.best = v;
                        mine.arg = idx;
                    }
                }
            }
        }
        PAIR_STAMP(4);          // surfaces
        __builtin_amdgcn_s_setprio(3);
        // ---- per-slot results ----
        for (int s = 0; s < slots; ++s) {
            const int state = info[s].state;
            if (state == 3)
                continue;
            const int64_t p = base + s;
            if (state == 0) {
                BestNan v = mine;
                if (slot != s) {
                    v.best = -INFINITY;
                    v.arg = 0x7fffffff;
                    v.has_nan = 0;
                }
                v = nthr > WAVE ? block_best(v, red_d, red_i) : wave_best(v);
                if (lane == 0) {
                    out_max[p] = v.has_nan ? NAN : v.best;
                    if (out_argmax)
                        out_argmax[p] = v.arg == 0x7fffffff ? -1 : v.arg;
                }
            } else if (state == 1) {
                const double fill = fabs(info[s].tmean - info[s].rmean) < min_mean_diff ? 1.0 : 0.0;
                if (surfaces) {
                    double *surf = surfaces + surf_off[p];
                    for (int i = lane; i < info[s].rh * info[s].rw; i += nthr)
                        surf[i] = fill;
                }
                if (lane == 0) {
                    out_max[p] = fill;
                    if (out_argmax)
                        out_argmax[p] = 0;
                }
            } else if (lane == 0) {
                out_max[p] = NAN;
                if (out_argmax)
                    out_argmax[p] = -1;
            }
        }
        PAIR_STAMP(5);          // maxima
    }
#ifdef SV_PAIR_STAMPS
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i)
            atomicAdd(&pair_stamps[i], stamp_acc_[i]);
#endif
}

// ---- float32 multiply-accumulate on the matrix cores: one wavefront per pair -------------
// The surface of a pair is a matrix product: with m' = (m + k) mod rh,
//     c[k][l] = sum_{m', n} A[k][(m', n)] * B[(m', n)][l],
//     A[k][(m', n)] = t'[(m' - k) mod rh][n]   (t' zero beyond its mh x mw values),
//     B[(m', n)][l] = r'[m'][(n + l) mod rw],
// i.e. rh * mw steps of rank 1, the same products as the direct sum and no others.  One
// v_mfma_f32_32x32x2_f32 covers two steps for a 32 x 32 tile of (k, l): it replaces 32
// vector FMAs per lane by one instruction and two 4-byte LDS reads, and float32 matrix and
// vector instructions have the same peak rate.  Operand addresses are a per-lane base plus
// the step's offset: lane k reads template row m' - k, or m' - k + rh for the rows before
// its own (one select per row of 16 instructions), and the region is staged wrapped
// horizontally to nloop + 32 columns (lane l reads column n + l).  Partial sums stay in
// float32 for at most 256 products (four accumulator tiles, each for a quarter of the rows)
// and are added in float64.  Sides up to 32.
typedef float pm16 __attribute__((ext_vector_type(16)));
constexpr int PM_SIDE = 32;
constexpr int PM_TP = 33;                 // template row pitch: the 32 rows of an A read on 32 banks
constexpr int PM_WAVES = 4;               // independent wavefronts (pairs) per workgroup

// ROWS: the largest side of the call, rounded up to 26 or 32 (rows a lane keeps, rows and
// columns of the LDS images); SETS: accumulator tiles, one per ROWS / SETS rows (at most 256
// products each).  Up to 26 the images are small enough for four wavefronts per SIMD.
template <int ROWS, int SETS>
__global__ void __launch_bounds__(PM_WAVES * WAVE, ROWS <= 26 ? 4 : 3)
scan_pairs_mfma_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                       const PairDesc *__restrict__ desc, int64_t npairs, int max_side, double min_std,
                       double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                       double *__restrict__ surfaces, const int64_t *__restrict__ surf_off)
{
    constexpr int HALF = ROWS / 2;                                   // rows per lane
    constexpr int PM_RP = ((ROWS + 3) & ~3) + PM_SIDE;               // region row pitch
    constexpr int PM_SLOT_FLOATS = ROWS * PM_TP + ROWS * PM_RP + 16; // + the reach of the reads requested ahead
    __shared__ float slots[PM_WAVES * PM_SLOT_FLOATS];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    float *Tt = slots + wave * PM_SLOT_FLOATS, *Rr = Tt + ROWS * PM_TP;
    const int lx = lane & 31, ly = lane >> 5;
    const int64_t first = (int64_t)blockIdx.x * PM_WAVES + wave, stride = (int64_t)gridDim.x * PM_WAVES;

    // A lane keeps column lx, rows ly, ly + 2, ... of both boxes.  Loads are not predicated: a
    // lane beyond the box reads the box's last row / column again and its value is masked
    // where it is used.  (Nothing is requested a pair ahead: the 64 registers that would
    // hold cost the third wavefront per SIMD, and with three wavefronts two compute while
    // one waits.)
    double tk[HALF], rk[HALF];
    auto request = [&](const PairDesc &d, bool ok) {
        const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
        if (ok && th > 0 && tw > 0) {
            const double *T = tbase + d.t_off;
            const unsigned pitch = (unsigned)d.t_pitch, cx = (unsigned)min(lx, tw - 1);
#pragma unroll
            for (int j = 0; j < HALF; ++j)
                tk[j] = T[(unsigned)min(ly + 2 * j, th - 1) * pitch + cx];
        } else {
#pragma unroll
            for (int j = 0; j < HALF; ++j)
                tk[j] = 0.0;
        }
        if (ok) {
            const double *R = rbase + d.r_off;
            const unsigned pitch = (unsigned)d.r_pitch, cx = (unsigned)min(lx, rw - 1);
#pragma unroll
            for (int j = 0; j < HALF; ++j)
                rk[j] = R[(unsigned)min(ly + 2 * j, rh - 1) * pitch + cx];
        } else {
#pragma unroll
            for (int j = 0; j < HALF; ++j)
                rk[j] = 0.0;
        }
    };
#ifdef SV_PAIR_STAMPS
    unsigned long long stamp_acc_[8] = {};
    unsigned long long stamp_ = clock64();
#endif
    // Descriptors travel two pairs ahead, one 8-byte word per lane (a scalar load would be
    // waited for at the next LDS operation: they share a counter).
    auto desc_word = [&](int64_t q) -> int64_t {
        return q < npairs ? reinterpret_cast<const int64_t *>(desc + q)[lane & 7] : (int64_t)0;
    };
    auto unpack = [&](int64_t w) {
        auto word = [&](int i) -> int64_t {
            const int lo = __builtin_amdgcn_readlane((int)(w & 0xffffffff), i);
            const int hi = __builtin_amdgcn_readlane((int)(w >> 32), i);
            return ((int64_t)hi << 32) | (uint32_t)lo;
        };
        PairDesc d;
        d.t_off = word(0);
        d.t_pitch = word(1);
        d.t_h = word(2);
        d.t_w = word(3);
        d.r_off = word(4);
        d.r_pitch = word(5);
        d.r_h = word(6);
        d.r_w = word(7);
        return d;
    };
    PairDesc d_next = unpack(desc_word(first));
    bool ok_next = first < npairs && pair_desc_ok(d_next, max_side) && max_side <= ROWS;
    int64_t w_after = desc_word(first + stride);
    for (int64_t p = first; p < npairs; p += stride) {
        const PairDesc d = d_next;
        const bool ok = ok_next;
        request(d, ok);
        d_next = unpack(w_after);
        ok_next = p + stride < npairs && pair_desc_ok(d_next, max_side) && max_side <= ROWS;
        w_after = desc_word(p + 2 * stride);
        if (!ok) {
            if (lane == 0) {
                out_max[p] = NAN;
                if (out_argmax)
                    out_argmax[p] = -1;
            }
            continue;
        }
        const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
        const int tn = th * tw, rn = rh * rw;
        const int mh = min(th, rh), mw = min(tw, rw);
        const int nloop = (mw + 3) & ~3;
        PAIR_STAMP(6);          // descriptor of the next pair
        double ts = 0.0, rs = 0.0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int y = ly + 2 * j;
            tk[j] = (lx < tw && y < th) ? tk[j] : 0.0;
            rk[j] = (lx < rw && y < rh) ? rk[j] : 0.0;
            ts += tk[j];
            rs += rk[j];
        }
        PAIR_STAMP(0);          // pixels arrived
        const double tmean = wave_sum(ts) / (double)tn, rmean = wave_sum(rs) / (double)rn;
        double tq = 0.0, rq = 0.0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int y = ly + 2 * j;
            tk[j] = (lx < tw && y < th) ? tk[j] - tmean : 0.0;
            rk[j] = (lx < rw && y < rh) ? rk[j] - rmean : 0.0;
            tq = fma(tk[j], tk[j], tq);
            rq = fma(rk[j], rk[j], rq);
        }
        const double tstd = sqrt(wave_sum(tq) / (double)tn), rstd = sqrt(wave_sum(rq) / (double)rn);
        PAIR_STAMP(1);          // statistics
        double *surf = surfaces ? surfaces + surf_off[p] : nullptr;
        if (tstd < min_std) {      // homogeneous template: constant surface
            const double fill = fabs(tmean - rmean) < min_mean_diff ? 1.0 : 0.0;
            if (surf)
                for (int i = lane; i < rn; i += WAVE)
                    surf[i] = fill;
            if (lane == 0) {
                out_max[p] = fill;
                if (out_argmax)
                    out_argmax[p] = 0;
            }
            continue;
        }
        // An empty template transforms to an all-zero spectrum: the surface is 0 wherever
        // the normalised region is finite.
        const double scale = tn == 0 ? ((rstd > 0.0) ? 0.0 : NAN) : 1.0 / (tstd * (double)tn) / rstd;

        // ---- staging (this wavefront's LDS operations complete in order) ----
        __builtin_amdgcn_wave_barrier();
        {
            // column x of the wrapped region image holds column x mod rw: fetched from the lane
            // that keeps it (same row parity), so that no write needs a predicate
            const int src0 = 4 * ((lx % rw) + 32 * ly), src1 = 4 * (((lx + 32) % rw) + 32 * ly);
            float *tcol = Tt + lx, *rcol = Rr + lx;
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int y = ly + 2 * j;
                // zeros beyond the template's own rows and columns (the 32 lanes of a row cover
                // every column a step reads, the 32 rows every row)
                tcol[y * PM_TP] = (y < mh && lx < mw) ? (float)tk[j] : 0.0f;
                const int rv = __float_as_int((float)rk[j]);
                rcol[y * PM_RP] = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, rv));
                const float w1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, rv));
                if (PM_RP >= 2 * PM_SIDE || lx + PM_SIDE < PM_RP)      // columns from PM_RP on are the next row's
                    rcol[y * PM_RP + PM_SIDE] = w1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        PAIR_STAMP(2);          // staging

        // ---- the surface ----
        pm16 acc[SETS];
#pragma unroll
        for (int q = 0; q < SETS; ++q)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                acc[q][v] = 0.0f;
        if (tn != 0) {
            // lanes beyond the surface's rows repeat its last row (their results are dropped)
            const int kk = min(lx, rh - 1);
            const float *pa_lo = Tt - kk * PM_TP + ly, *pa_hi = pa_lo + rh * PM_TP;   // + m' * PM_TP + n
            const float *pb = Rr + lx + ly;                                          // + m' * PM_RP + n
            const int rows_q = (rh + SETS - 1) / SETS;
            const int chunks = nloop / 4;
#pragma unroll
            for (int q = 0; q < SETS; ++q) {
                const int m_end = min((q + 1) * rows_q, rh);
                for (int m = q * rows_q; m < m_end; ++m) {
                    const float *ra = (m >= kk ? pa_lo : pa_hi) + m * PM_TP, *rb = pb + m * PM_RP;
                    // two chunks per round, each chunk's operands requested a chunk ahead (past the
                    // row's end they are read and dropped: still inside the slot); the scheduling
                    // barriers keep the requests ahead of the products they hide behind
                    // (measured: +3 % with three accumulator sets and four wavefronts per SIMD, -3 %
                    // with four sets and three wavefronts, which therefore keep the plain loop)
                    float a0 = ra[0], a1 = ra[2], b0 = rb[0], b1 = rb[2];
                    int c = 0;
                    for (; SETS < 4 && c + 2 <= chunks; c += 2) {
                        const float na0 = ra[4 * c + 4], na1 = ra[4 * c + 6];
                        const float nb0 = rb[4 * c + 4], nb1 = rb[4 * c + 6];
                        __builtin_amdgcn_sched_barrier(0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[q], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        a0 = ra[4 * c + 8];
                        a1 = ra[4 * c + 10];
                        b0 = rb[4 * c + 8];
                        b1 = rb[4 * c + 10];
                        __builtin_amdgcn_sched_barrier(0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(na0, nb0, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(na1, nb1, acc[q], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    for (; c < chunks; ++c) {
                        const float na0 = ra[4 * c + 4], na1 = ra[4 * c + 6];
                        const float nb0 = rb[4 * c + 4], nb1 = rb[4 * c + 6];
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[q], 0, 0, 0);
                        a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;
                    }
                }
            }
        }
        PAIR_STAMP(4);          // products
        // element v of a lane: row 8 (v / 4) + 4 (lane / 32) + v % 4, column lane % 32; a lane's
        // indices ascend with v, so its first maximum is the one a strict comparison keeps
        BestNan mine;
        mine.best = -INFINITY;
        mine.arg = 0x7fffffff;
        mine.has_nan = 0;
        const int idx0 = 4 * ly * rw + lx;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int k0 = 8 * (v / 4) + (v % 4);            // + 4 ly
            const bool valid = k0 + 4 * ly < rh && lx < rw;
            double sum = (double)acc[0][v];
#pragma unroll
            for (int q = 1; q < SETS; ++q) {
                sum += (double)acc[q][v];
            }
            const double val = tn == 0 ? scale : sum * scale;
            const int idx = idx0 + k0 * rw;
            mine.has_nan |= (valid && isnan(val)) ? 1 : 0;
            const bool better = valid && val > mine.best;
            mine.best = better ? val : mine.best;
            mine.arg = better ? idx : mine.arg;
        }
        if (surf) {             // the whole surface is wanted: rare, kept out of the loop above
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int k0 = 8 * (v / 4) + (v % 4);
                double sum = (double)acc[0][v];
#pragma unroll
                for (int q = 1; q < SETS; ++q) {
                    sum += (double)acc[q][v];
                }
                if (k0 + 4 * ly < rh && lx < rw)
                    surf[idx0 + k0 * rw] = tn == 0 ? scale : sum * scale;
            }
        }
        mine = wave_best(mine);
        if (lane == 0) {
            out_max[p] = mine.has_nan ? NAN : mine.best;
            if (out_argmax)
                out_argmax[p] = mine.arg == 0x7fffffff ? -1 : mine.arg;
        }
        PAIR_STAMP(5);          // maximum
    }
#ifdef SV_PAIR_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i)
            atomicAdd(&pair_stamps[i], stamp_acc_[i]);
#endif
}

// Sides up to 16: four pairs per wavefront, one per row of 16 lanes, on the four-block
// instruction v_mfma_f32_16x16x1_f32 (block b multiplies pair b's operands: one rank-1 step
// of a 16 x 16 tile per instruction).  Staging as above with a lane owning one column of
// its pair's template and region (16 rows in registers); sums run down the column and
// across the row of 16 lanes.  The four pairs walk the largest row and column counts among
// them; what a smaller pair reads beyond its own is a zero of its template or region
// image.  At most 256 products per surface element: one float32 accumulator.
constexpr int PS_SIDE = 16;
constexpr int PS_TP = 17;                 // template row pitch
constexpr int PS_RP = 2 * PS_SIDE;        // region row pitch (wrapped to nloop + 16 columns)
constexpr int PS_WAVES = 4;

// ROWS: the largest side of the call (8, 12 or 16): rows a lane keeps, rows of the LDS
// images - smaller pairs cost fewer instructions and leave room for more wavefronts.
template <int ROWS>
__global__ void __launch_bounds__(PS_WAVES * WAVE, ROWS <= 12 ? 4 : 3)
scan_pairs_mfma16_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                         const PairDesc *__restrict__ desc, int64_t npairs, int max_side, double min_std,
                         double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                         double *__restrict__ surfaces, const int64_t *__restrict__ surf_off)
{
    constexpr int PS_PAIR_FLOATS = ROWS * PS_TP + ROWS * PS_RP;
    constexpr int PS_SLOT_FLOATS = 4 * PS_PAIR_FLOATS + 8;               // a wavefront's four pairs
    __shared__ float slots[PS_WAVES * PS_SLOT_FLOATS];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int x = lane & 15, g = lane >> 4;
    float *Tt = slots + wave * PS_SLOT_FLOATS + g * PS_PAIR_FLOATS, *Rr = Tt + ROWS * PS_TP;
    const int64_t first = ((int64_t)blockIdx.x * PS_WAVES + wave) * 4, stride = (int64_t)gridDim.x * PS_WAVES * 4;

    // descriptors are read one round ahead (every lane of a row its pair's)
    auto read_desc = [&](int64_t q) {
        PairDesc d;
        d.t_off = d.r_off = d.t_pitch = d.r_pitch = 0;
        d.t_h = d.t_w = d.r_h = d.r_w = 0;
        if (q < npairs)
            d = desc[q];
        return d;
    };
    PairDesc d_next = read_desc(first + g);
    for (int64_t base = first; base < npairs; base += stride) {
        const int64_t p = base + g;
        const bool present = p < npairs;
        const PairDesc d = d_next;
        d_next = read_desc(p + stride);
        const bool ok = present && pair_desc_ok(d, max_side) && max_side <= ROWS;
        const int th = ok ? (int)d.t_h : 0, tw = ok ? (int)d.t_w : 0;
        const int rh = ok ? (int)d.r_h : 0, rw = ok ? (int)d.r_w : 0;
        // ---- statistics: lane x keeps column x of its pair's boxes; loads are not predicated
        // (a lane beyond the box reads its last row / column again; masked below) ----
        double tc[ROWS], rc[ROWS];
        {
            const bool tl = th > 0 && tw > 0, rl = rh > 0 && rw > 0;
            const double *T = tbase + (tl ? d.t_off : 0), *R = rbase + (rl ? d.r_off : 0);
            const unsigned tp = tl ? (unsigned)d.t_pitch : 0u, rp = rl ? (unsigned)d.r_pitch : 0u;
            const unsigned tx = tl ? (unsigned)min(x, tw - 1) : 0u, rx = rl ? (unsigned)min(x, rw - 1) : 0u;
#pragma unroll
            for (int y = 0; y < ROWS; ++y) {
                tc[y] = T[(unsigned)min(y, max(th - 1, 0)) * tp + tx];
                rc[y] = R[(unsigned)min(y, max(rh - 1, 0)) * rp + rx];
            }
        }
        double ts = 0.0, rs = 0.0;
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            tc[y] = (x < tw && y < th) ? tc[y] : 0.0;
            rc[y] = (x < rw && y < rh) ? rc[y] : 0.0;
            ts += tc[y];
            rs += rc[y];
        }
        const double tn = (double)(th * tw), rn = (double)(rh * rw);
        const double tmean = row16_sum(ts) / tn, rmean = row16_sum(rs) / rn;
        double tq = 0.0, rq = 0.0;
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            tc[y] = (x < tw && y < th) ? tc[y] - tmean : 0.0;
            rc[y] = (x < rw && y < rh) ? rc[y] - rmean : 0.0;
            tq = fma(tc[y], tc[y], tq);
            rq = fma(rc[y], rc[y], rq);
        }
        const double tstd = sqrt(row16_sum(tq) / tn), rstd = sqrt(row16_sum(rq) / rn);
        const bool flat = ok && tstd < min_std;
        const bool evaluate = ok && !flat;
        const int mh = min(th, rh), mw = min(tw, rw);
        const int nloop = evaluate ? (mw + 3) & ~3 : 0;
        // An empty template transforms to an all-zero spectrum: the surface is 0 wherever
        // the normalised region is finite.
        const double scale = th * tw == 0 ? ((rstd > 0.0) ? 0.0 : NAN) : 1.0 / (tstd * tn) / rstd;

        // ---- staging (this wavefront's LDS operations complete in order).  A pair that is
        // not evaluated stages zeros: its lanes walk the other pairs' steps. ----
        __builtin_amdgcn_wave_barrier();
        {
            const int rwc = max(rw, 1);
            const int src0 = 4 * (16 * g + x % rwc), src1 = 4 * (16 * g + (x + 16) % rwc);
#pragma unroll
            for (int y = 0; y < ROWS; ++y) {
                Tt[y * PS_TP + x] = (evaluate && y < mh && x < mw) ? (float)tc[y] : 0.0f;
                const int rv = __float_as_int(evaluate ? (float)rc[y] : 0.0f);
                Rr[y * PS_RP + x] = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, rv));
                Rr[y * PS_RP + x + 16] = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, rv));
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- the surfaces: the step counts of the largest pair in the wavefront ----
        int u_rows = evaluate ? rh : 0, u_cols = nloop;
        u_rows = max(max(__builtin_amdgcn_readlane(u_rows, 0), __builtin_amdgcn_readlane(u_rows, 16)),
                     max(__builtin_amdgcn_readlane(u_rows, 32), __builtin_amdgcn_readlane(u_rows, 48)));
        u_cols = max(max(__builtin_amdgcn_readlane(u_cols, 0), __builtin_amdgcn_readlane(u_cols, 16)),
                     max(__builtin_amdgcn_readlane(u_cols, 32), __builtin_amdgcn_readlane(u_cols, 48)));
        // two accumulators take turns, so that an instruction does not wait for the one before it
        pm16 acc, acc_b;
#pragma unroll
        for (int v = 0; v < 16; ++v)
            acc[v] = acc_b[v] = 0.0f;
        {
            // lane (g, k) reads template row m' - k, or m' - k + rh for the rows before its own;
            // lanes beyond the surface's rows repeat its last row (their results are dropped)
            const int rhc = max(rh, 1), kk = min(x, rhc - 1);
            const float *pa_lo = Tt - kk * PS_TP, *pa_hi = pa_lo + rhc * PS_TP;
            const float *pb = Rr + x;
            for (int m = 0; m < u_rows; ++m) {
                // rows from rh on (a smaller pair): the region image holds zeros there and any
                // template row will do - the select below stays inside the image
                const float *ra = (m >= kk ? pa_lo : pa_hi) + min(m, rhc - 1 + kk) * PS_TP, *rb = pb + m * PS_RP;
                for (int n = 0; n < u_cols; n += 4) {
                    const float a0 = ra[n], a1 = ra[n + 1], a2 = ra[n + 2], a3 = ra[n + 3];
                    const float b0 = rb[n], b1 = rb[n + 1], b2 = rb[n + 2], b3 = rb[n + 3];
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, b0, acc, 0, 0, 0);
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, b1, acc_b, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a2, b2, acc, 0, 0, 0);
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x1f32(a3, b3, acc_b, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v)
            acc[v] += acc_b[v];
        // ---- results: element v of a lane belongs to pair v / 4, row 4 (lane / 16) + v % 4,
        // column lane % 16.  The raw sums are compared (a pair's scale is positive and the
        // same for all its elements; distinct float32 sums stay distinct in float64), first
        // inside the lane, then across the lanes in a way that ends with pair b's result in
        // row b of the lanes - the row that holds the pair's shape, scale and state. ----
        float bv[4];
        int bi[4], bn[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int brh = __builtin_amdgcn_readlane(rh, 16 * b), brw = __builtin_amdgcn_readlane(rw, 16 * b);
            bv[b] = -INFINITY;
            bi[b] = 0x7fffffff;
            bn[b] = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * g + e;
                const bool valid = k < brh && x < brw;
                const float val = acc[4 * b + e];
                bn[b] |= (valid && isnan(val)) ? 1 : 0;
                const bool better = valid && val > bv[b];       // ascending index inside the lane
                bv[b] = better ? val : bv[b];
                bi[b] = better ? k * brw + x : bi[b];
            }
        }
        auto merge = [](float &v, int &i, int &n, float ov, int oi, int on) {
            const bool take = ov > v || (ov == v && oi < i);
            v = take ? ov : v;
            i = take ? oi : i;
            n |= on;
        };
        {
            // lanes 32 apart: the lower half keeps pairs 0 and 1, the upper half pairs 2 and 3
            const bool upper = lane >= 32;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float sv = upper ? bv[h] : bv[h + 2];
                const int si = upper ? bi[h] : bi[h + 2], sn = upper ? bn[h] : bn[h + 2];
                const float ov = __shfl_xor(sv, 32, WAVE);
                const int oi = __shfl_xor(si, 32, WAVE), on = __shfl_xor(sn, 32, WAVE);
                float &kv = bv[h];
                if (upper) {
                    bv[h] = bv[h + 2];
                    bi[h] = bi[h + 2];
                    bn[h] = bn[h + 2];
                }
                merge(kv, bi[h], bn[h], ov, oi, on);
            }
            // lanes 16 apart: even rows keep the first of their two pairs, odd rows the second
            const bool odd = (g & 1) != 0;
            const float sv = odd ? bv[0] : bv[1];
            const int si = odd ? bi[0] : bi[1], sn = odd ? bn[0] : bn[1];
            const float ov = __shfl_xor(sv, 16, WAVE);
            const int oi = __shfl_xor(si, 16, WAVE), on = __shfl_xor(sn, 16, WAVE);
            if (odd) {
                bv[0] = bv[1];
                bi[0] = bi[1];
                bn[0] = bn[1];
            }
            merge(bv[0], bi[0], bn[0], ov, oi, on);
        }
        // across the 16 lanes of the row (every lane ends with the row's best)
        {
            float v = bv[0];
            int i = bi[0], n = bn[0];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
                merge(v, i, n, __shfl_xor(v, o, WAVE), __shfl_xor(i, o, WAVE), __shfl_xor(n, o, WAVE));
            bv[0] = v;
            bi[0] = i;
            bn[0] = n;
        }
        if (present && x == 0) {
            double result;
            int64_t arg;
            if (!ok) {
                result = NAN;
                arg = -1;
            } else if (flat) {
                result = fabs(tmean - rmean) < min_mean_diff ? 1.0 : 0.0;
                arg = 0;
            } else {
                // a constant region (scale infinite or NaN): every element is 0 / 0
                const bool all_nan = !isfinite(scale);
                const double val = th * tw == 0 ? scale : (double)bv[0] * scale;
                result = (bn[0] || all_nan) ? NAN : val;
                arg = (all_nan || bi[0] == 0x7fffffff) ? -1 : bi[0];
            }
            out_max[p] = result;
            if (out_argmax)
                out_argmax[p] = arg;
        }
        if (surfaces) {          // the whole surfaces are wanted: rare
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (base + b >= npairs)
                    continue;
                const int state = __builtin_amdgcn_readlane(!ok ? 2 : flat ? 1 : 0, 16 * b);
                if (state == 2)
                    continue;
                const int brh = __builtin_amdgcn_readlane(rh, 16 * b), brw = __builtin_amdgcn_readlane(rw, 16 * b);
                double *surf = surfaces + surf_off[base + b];
                if (state == 1) {
                    const double tm = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tmean), 16 * b),
                                                       __builtin_amdgcn_readlane(__double2loint(tmean), 16 * b));
                    const double rm = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rmean), 16 * b),
                                                       __builtin_amdgcn_readlane(__double2loint(rmean), 16 * b));
                    const double fill = fabs(tm - rm) < min_mean_diff ? 1.0 : 0.0;
                    for (int i = lane; i < brh * brw; i += WAVE)
                        surf[i] = fill;
                    continue;
                }
                const double bscale = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(scale), 16 * b),
                                                       __builtin_amdgcn_readlane(__double2loint(scale), 16 * b));
                const bool empty = __builtin_amdgcn_readlane(th * tw, 16 * b) == 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * g + e;
                    if (k < brh && x < brw)
                        surf[k * brw + x] = empty ? bscale : (double)acc[4 * b + e] * bscale;
                }
            }
        }
    }
}

// Sides 33 ... 64: one pair per workgroup of T wavefronts (T = 3 up to 48, else 4) on
// v_mfma_f32_16x16x4_f32.  The surface is cut into 16 x 16 tiles; wavefront w owns the
// tiles of columns 16 w ... 16 w + 15, so that one step reads T template operands (one per
// row tile) and one region operand for T instructions.  Operand images as above (the
// template not doubled: a select per row and tile; the region wrapped to nloop + 16 T
// columns), sized by the call's largest side.  Float32 sums are folded into float64 after
// at most 256 products.
typedef float pm4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pw_round4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline int pw_tp(int max_side) { return pw_round4(max_side) + 1; }
__host__ __device__ inline int pw_rp(int max_side, int tiles) { return pw_round4(max_side) + 16 * tiles + 4; }

template <int T>
__global__ void __launch_bounds__(T * WAVE)
scan_pairs_mfma_wide_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                            const PairDesc *__restrict__ desc, int64_t npairs, int max_side, double min_std,
                            double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                            double *__restrict__ surfaces, const int64_t *__restrict__ surf_off)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double red[17];
    __shared__ double red_d[4];
    __shared__ int red_i[8];
    const int tp = pw_tp(max_side), rp = pw_rp(max_side, T);
    float *Tt = reinterpret_cast<float *>(smem_raw), *Rr = Tt + max_side * tp;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    constexpr int NTHR = T * WAVE;

    for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
        const PairDesc d = desc[p];
        __syncthreads();            // the images are rewritten
        if (!pair_desc_ok(d, max_side) || max_side > 16 * T) {
            if (threadIdx.x == 0) {
                out_max[p] = NAN;
                if (out_argmax)
                    out_argmax[p] = -1;
            }
            continue;
        }
        const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
        const int tn = th * tw, rn = rh * rw;
        const int mh = min(th, rh), mw = min(tw, rw);
        const int nloop = pw_round4(mw);
        const double *Tl = tbase + d.t_off + lane, *Rl = rbase + d.r_off + lane;
        const bool tcol = lane < tw, rcol = lane < rw;
        // pass 1: means (wavefront w walks rows w, w + T, ...)
        double ts = 0.0, rs = 0.0;
        for (int y = wave; y < max(th, rh); y += T) {
            if (tcol && y < th)
                ts += Tl[y * d.t_pitch];
            if (rcol && y < rh)
                rs += Rl[y * d.r_pitch];
        }
        pair_sum2(ts, rs, red, NTHR);
        const double tmean = ts / (double)tn, rmean = rs / (double)rn;
        // pass 2: squared deviations, and the images on the way.  Column x of the wrapped
        // region image holds column x mod rw: fetched from the lane that keeps it.
        double tq = 0.0, rq = 0.0;
        const int src0 = 4 * (lane % rw), src1 = 4 * ((lane + WAVE) % rw);
        for (int y = wave; y < max(th, rh); y += T) {
            double tv = 0.0, rv = 0.0;
            if (tcol && y < th) {
                tv = Tl[y * d.t_pitch] - tmean;
                tq = fma(tv, tv, tq);
            }
            if (rcol && y < rh) {
                rv = Rl[y * d.r_pitch] - rmean;
                rq = fma(rv, rv, rq);
            }
            if (y < rh) {
                // zeros beyond the template's own rows and columns
                if (lane < tp)
                    Tt[y * tp + lane] = (y < mh && lane < mw) ? (float)tv : 0.0f;
                const int bits = __float_as_int((float)rv);
                const float w0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, bits));
                const float w1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, bits));
                Rr[y * rp + lane] = w0;
                if (lane + WAVE < rp)
                    Rr[y * rp + lane + WAVE] = w1;
            }
        }
        pair_sum2(tq, rq, red, NTHR);
        const double tstd = sqrt(tq / (double)tn), rstd = sqrt(rq / (double)rn);
        double *surf = surfaces ? surfaces + surf_off[p] : nullptr;
        if (tstd < min_std) {      // homogeneous template: constant surface
            const double fill = fabs(tmean - rmean) < min_mean_diff ? 1.0 : 0.0;
            if (surf)
                for (int i = threadIdx.x; i < rn; i += NTHR)
                    surf[i] = fill;
            if (threadIdx.x == 0) {
                out_max[p] = fill;
                if (out_argmax)
                    out_argmax[p] = 0;
            }
            continue;
        }
        __syncthreads();
        const double scale = tn == 0 ? ((rstd > 0.0) ? 0.0 : NAN) : 1.0 / (tstd * (double)tn) / rstd;

        // ---- the surface: lane (kk, i) of an instruction supplies step n0 + kk of row /
        // column i of the tile ----
        const int li = lane & 15, kq = lane >> 4;
        pm4 acc[T];
        double dsum[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                acc[t][v] = 0.0f;
                dsum[t][v] = 0.0;
            }
        if (tn != 0) {
            int kc[T];
            const float *pa_lo[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                kc[t] = min(16 * t + li, rh - 1);       // rows beyond the surface repeat its last row
                pa_lo[t] = Tt - kc[t] * tp + kq;
            }
            const float *pb = Rr + 16 * wave + li + kq;
            const int flush_rows = max(256 / max(nloop, 1), 1);
            int since = 0;
            for (int m = 0; m < rh; ++m) {
                const float *ra[T];
#pragma unroll
                for (int t = 0; t < T; ++t)
                    ra[t] = pa_lo[t] + (m >= kc[t] ? m : m + rh) * tp;
                const float *rb = pb + m * rp;
                for (int n = 0; n < nloop; n += 4) {
                    const float b = rb[n];
                    float a[T];
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        a[t] = ra[t][n];
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b, acc[t], 0, 0, 0);
                }
                if (++since == flush_rows || m == rh - 1) {
                    since = 0;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            dsum[t][v] += (double)acc[t][v];
                            acc[t][v] = 0.0f;
                        }
                }
            }
        }
        // element v of tile t in a lane: row 16 t + 4 (lane / 16) + v, column 16 w + lane % 16
        BestNan mine;
        mine.best = -INFINITY;
        mine.arg = 0x7fffffff;
        mine.has_nan = 0;
        const int l = 16 * wave + li;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int k = 16 * t + 4 * kq + v;
                const bool valid = k < rh && l < rw;
                const double val = tn == 0 ? scale : dsum[t][v] * scale;
                const int idx = k * rw + l;
                if (surf && valid)
                    surf[idx] = val;
                mine.has_nan |= (valid && isnan(val)) ? 1 : 0;
                const bool better = valid && val > mine.best;      // ascending index inside the lane
                mine.best = better ? val : mine.best;
                mine.arg = better ? idx : mine.arg;
            }
        const BestNan all = block_best(mine, red_d, red_i);
        if (threadIdx.x == 0) {
            out_max[p] = all.has_nan ? NAN : all.best;
            if (out_argmax)
                out_argmax[p] = all.arg == 0x7fffffff ? -1 : all.arg;
        }
    }
}

// image_scan's sequential selection rule, one thread per window.
__global__ void __launch_bounds__(256)
scan_fold_kernel(const double *__restrict__ corr_max, const uint8_t *__restrict__ is_centre,
                 const uint8_t *__restrict__ skip, const int64_t *__restrict__ win_off, int64_t nwin,
                 double threshold, double centre_tol, int64_t *__restrict__ best_pair,
                 double *__restrict__ best_val, double *__restrict__ min_margin)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= nwin)
        return;
    double best = 0.0, margin = INFINITY;
    int64_t arg = -1;
    for (int64_t p = win_off[w]; p < win_off[w + 1]; ++p) {
        if (skip && skip[p])
            continue;
        const double c = corr_max[p];
        if (isnan(c))
            continue;
        const bool centre = is_centre[p] != 0;
        const double bar = centre ? best : best + centre_tol;
        margin = fmin(margin, fmin(fabs(c - threshold), fabs(c - bar)));
        const bool upd = c > threshold && (centre ? c >= bar : c > bar);
        if (upd) {
            best = c;
            arg = p;
        }
    }
    best_pair[w] = arg;
    best_val[w] = best;
    if (min_margin)
        min_margin[w] = margin;
}

}  // namespace svamd

using namespace svamd;

namespace svamd {

// How the float32 kernel cuts a call's pairs into wavefront work: columns per task
// (8 or 4), wavefronts per workgroup and pairs per workgroup.
struct PairShape {
    int G, waves, slots;
};

// SVAMD_PAIR_SHAPE="G,waves,slots" replaces the built-in choice (measurements only).
static void pair_shape_override(int max_side, PairShape &shape)
{
    (void)max_side;
    const char *env = getenv("SVAMD_PAIR_SHAPE");
    int g = 0, w = 0, sl = 0;
    if (env && sscanf(env, "%d,%d,%d", &g, &w, &sl) == 3 && (g == 4 || g == 8) && w >= 1 && w <= 4 && sl >= 1) {
        shape.G = g;
        shape.waves = w;
        shape.slots = sl;
    }
}

int launch_pairs(const double *tbase, const double *rbase, const int64_t *desc, int64_t npairs, int max_side,
                 double min_std, double min_mean_diff, int acc_precision, double *out_max, int64_t *out_argmax,
                 double *surfaces, const int64_t *surf_off, const int32_t *list, const int32_t *list_count,
                 unsigned list_grid, hipStream_t st)
{
    const PairDesc *pd = reinterpret_cast<const PairDesc *>(desc);
    const size_t side4 = ((size_t)max_side + 3) & ~(size_t)3;
    // float64 kernel: tasks of 8 surface columns from 16 columns on (half the LDS reads per FMA), else of 4
    const int w64 = max_side >= 16 ? 8 : 4;
    const size_t elems = (size_t)max_side * side4 +
                         (size_t)max_side * (w64 == 8 ? pair64_row_pitch<8>(max_side, (int)side4)
                                                      : pair64_row_pitch<4>(max_side, (int)side4));
    // small windows get small workgroups
    const int tasks = max_side * ((max_side + w64 - 1) / w64);
    const int threads = tasks <= 64 ? 64 : tasks <= 128 ? 128 : 256;
    const unsigned grid = list ? list_grid : (unsigned)(npairs < (1 << 20) ? npairs : (1 << 20));
    // SVAMD_PAIR_MFMA=0 keeps the vector-ALU kernel for every side (measurements only)
    static const bool mfma_off = getenv("SVAMD_PAIR_MFMA") && atoi(getenv("SVAMD_PAIR_MFMA")) == 0;
    if (acc_precision == SVAMD_ACC_F32 && max_side <= PS_SIDE && !mfma_off) {
        // four pairs per wavefront on the matrix cores
        const int64_t groups = (npairs + 4 * PS_WAVES - 1) / (4 * PS_WAVES);
        const unsigned mgrid = (unsigned)(groups < 256 * 4 ? groups : 256 * 4);
        if (max_side <= 8)
            hipLaunchKernelGGL(scan_pairs_mfma16_kernel<8>, dim3(mgrid), dim3(PS_WAVES * WAVE), 0, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        else if (max_side <= 12)
            hipLaunchKernelGGL(scan_pairs_mfma16_kernel<12>, dim3(mgrid), dim3(PS_WAVES * WAVE), 0, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        else
            hipLaunchKernelGGL(scan_pairs_mfma16_kernel<16>, dim3(mgrid), dim3(PS_WAVES * WAVE), 0, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    // Sides 17 ... 20 fill two fifths of a 32 x 32 tile or less: the vector kernel is 10 - 15 % faster there
    // (20: 37.5 against 34 TFLOP/s), from 21 on the matrix cores win (21: 36 against 26).
    if (acc_precision == SVAMD_ACC_F32 && max_side > 20 && max_side <= PM_SIDE && !mfma_off) {
        // one wavefront per pair on the matrix cores; three workgroups of four wavefronts fit a CU (four up to 26 x 26)
        const int64_t groups = (npairs + PM_WAVES - 1) / PM_WAVES;
        const int64_t resident = 256 * (max_side <= 26 ? 4 : 3);
        const unsigned mgrid = (unsigned)(groups < resident ? groups : resident);
        if (max_side <= 26)
            hipLaunchKernelGGL((scan_pairs_mfma_kernel<26, 3>), dim3(mgrid), dim3(PM_WAVES * WAVE), 0, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        else
            hipLaunchKernelGGL((scan_pairs_mfma_kernel<32, 4>), dim3(mgrid), dim3(PM_WAVES * WAVE), 0, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    // Sides above 32 are covered by 48 x 48 or 64 x 64 tiles: the matrix cores win where the
    // side fills them (45 ... 48: 79 against 61 TFLOP/s; 56: 80 against 73; 60: 93 against 67;
    // 64: 107 against 78), the vector kernel where a third or more of the tiles would be
    // padding (40: 57 against 65; 50: 60 against 65).
    const bool wide_fits = (max_side >= 45 && max_side <= 48) || max_side >= 55;
    if (acc_precision == SVAMD_ACC_F32 && max_side > PM_SIDE && wide_fits && !mfma_off) {
        // one pair per workgroup, a wavefront per strip of 16 surface columns
        const int tiles = max_side <= 48 ? 3 : 4;
        const size_t lds = ((size_t)max_side * pw_tp(max_side) + (size_t)max_side * pw_rp(max_side, tiles)) * sizeof(float);
        const unsigned wgrid = (unsigned)(npairs < 256 * 8 ? npairs : 256 * 8);
        if (tiles == 3)
            hipLaunchKernelGGL(scan_pairs_mfma_wide_kernel<3>, dim3(wgrid), dim3(3 * WAVE), lds, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        else
            hipLaunchKernelGGL(scan_pairs_mfma_wide_kernel<4>, dim3(wgrid), dim3(4 * WAVE), lds, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    if (acc_precision == SVAMD_ACC_F32) {
        // columns per task: 8 unless 4 wastes fewer padded columns
        // Pairs with more than 32 tasks of 8 columns get a workgroup of two wavefronts
        // and tasks of 4 columns: their LDS slot leaves room for few workgroups per CU,
        // and a SIMD that holds a single wavefront cannot hide its LDS latency.
        const int cols4 = (max_side + 3) / 4 * 4, cols8 = (max_side + 7) / 8 * 8;
        const bool two_waves = (max_side + 1) / 2 * (cols8 / 8) > WAVE / 2;
        PairShape shape;
        shape.G = (cols8 == cols4 && !two_waves) ? 8 : 4;
        {
            const int tasks = (max_side + 1) / 2 * ((shape.G == 8 ? cols8 : cols4) / shape.G);
            shape.slots = two_waves ? 1 : WAVE / tasks;
            // four wavefronts when two would need more than one round of tasks: large pairs fill an
            // LDS slot that leaves room for one or two workgroups per CU
            shape.waves = !two_waves ? 1 : (tasks > 2 * WAVE ? 4 : 2);
        }
        // Sides above 16: what decides is how many of a workgroup's lanes have a task (a side's
        // row pairs x column groups rarely fill whole wavefronts) against how many workgroups
        // the LDS slots leave room for; the shapes below are the measured best per range of
        // sides (profiles/pair_shapes.py: +5 to +26 % over the rule above, none slower).
        if (max_side > 16) {
            const int tasks8 = (max_side + 1) / 2 * (cols8 / 8);
            if (max_side <= 17)
                shape = PairShape{8, 1, 2};
            else if (max_side <= 20)
                shape = PairShape{4, 1, 1};
            else if (max_side <= 26)
                shape = PairShape{8, 1, 1};
            else if (max_side <= 32)
                shape = max_side == 31 ? PairShape{4, 4, 2} : PairShape{4, 2, 1};
            else if (max_side <= 38)
                shape = PairShape{4, 3, 1};
            else if (max_side <= 40)
                shape = PairShape{8, 4, 2};
            else if (max_side <= 44)
                shape = PairShape{4, 4, 1};
            else
                shape = PairShape{8, tasks8 > 3 * WAVE ? 4 : 3, 1};
        }
        pair_shape_override(max_side, shape);
        const int G = shape.G;
        const int gcols = G == 8 ? cols8 : cols4;
        const int t_floats = max_side * (int)side4;
        const int rcols = pair_row_cols(gcols, (int)side4);
        const int rxp = G == 4 ? pair_row_pitch<4>(rcols) : pair_row_pitch<8>(rcols);
        const int s_floats = 2 * max_side * rxp;
        const int slot_floats = t_floats + s_floats + max_side * (int)side4;   // template, row pairs, plain region
        const int tasks_per_slot = (max_side + 1) / 2 * (gcols / G);
        int slots = shape.slots;
        if (slots < 1) slots = 1;
        const int fthreads = shape.waves * WAVE;
        if (slots > PAIR_MAX_SLOTS) slots = PAIR_MAX_SLOTS;
        if (slots > 1 && slots * tasks_per_slot > fthreads)      // pairs that share a workgroup get one round of tasks
            slots = fthreads / tasks_per_slot > 0 ? fthreads / tasks_per_slot : 1;
        const size_t lds = (size_t)slots * slot_floats * sizeof(float);
        SV_REQUIRE(lds <= 128 * 1024, SVAMD_ERR_LIMIT, "scan_pairs: %d pairs of side %d per workgroup exceed the LDS",
                   slots, max_side);
        const int64_t groups = (npairs + slots - 1) / slots;
        const unsigned fgrid = (unsigned)(groups < (1 << 20) ? groups : (1 << 20));
        // pairs share one wavefront: staged by rows of 16 lanes
        const bool shared = slots > 1 && max_side <= 16 && fthreads == WAVE;
        auto launch = [&](auto kernel) -> int {
            if (lds > 64 * 1024) {
                SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            }
            hipLaunchKernelGGL(kernel, dim3(fgrid), dim3(fthreads), lds, st, tbase, rbase, pd, npairs, max_side, slots,
                               tasks_per_slot, slot_floats, t_floats, s_floats, min_std, min_mean_diff, out_max,
                               out_argmax, surfaces, surf_off);
            SV_LAUNCH_CHECK();
            return SVAMD_OK;
        };
        int rc;
        if (G == 8)
            rc = shared ? launch(&scan_pairs_f32_kernel<8, true>) : launch(&scan_pairs_f32_kernel<8, false>);
        else
            rc = shared ? launch(&scan_pairs_f32_kernel<4, true>) : launch(&scan_pairs_f32_kernel<4, false>);
        if (rc != SVAMD_OK)
            return rc;
    } else {
        const size_t lds = elems * sizeof(double);
        // the LDS limit is per function and per device: raised before every launch that needs it
        if (w64 == 8) {
            if (lds > 64 * 1024) {
                SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_pairs_kernel<double, 8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            }
            hipLaunchKernelGGL((scan_pairs_kernel<double, 8>), dim3(grid), dim3(threads), lds, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off,
                               list, list_count);
        } else {
            if (lds > 64 * 1024) {
                SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_pairs_kernel<double, 4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            }
            hipLaunchKernelGGL((scan_pairs_kernel<double, 4>), dim3(grid), dim3(threads), lds, st, tbase, rbase, pd,
                               npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces, surf_off,
                               list, list_count);
        }
    }
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

}  // namespace svamd

extern "C" int svamd_scan_pairs(const double *tbase, const double *rbase, const int64_t *desc,
                                int64_t npairs, int max_side, double min_std, double min_mean_diff,
                                int acc_precision, double *out_max, int64_t *out_argmax, double *surfaces,
                                const int64_t *surf_off, void *stream)
{
    SV_REQUIRE(npairs >= 0, SVAMD_ERR_INVALID, "scan_pairs: negative pair count");
    if (npairs == 0)
        return SVAMD_OK;
    SV_REQUIRE(tbase && rbase && desc && out_max, SVAMD_ERR_INVALID, "scan_pairs: NULL pointer");
    SV_REQUIRE(!surfaces || surf_off, SVAMD_ERR_INVALID, "scan_pairs: surfaces given without offsets");
    SV_REQUIRE(max_side > 0 && max_side <= SVAMD_PAIR_MAX_SIDE, SVAMD_ERR_LIMIT,
               "scan_pairs: max_side %d outside 1..%d", max_side, SVAMD_PAIR_MAX_SIDE);
    SV_REQUIRE(acc_precision == SVAMD_ACC_F32 || acc_precision == SVAMD_ACC_F64, SVAMD_ERR_INVALID,
               "scan_pairs: unknown accumulation precision %d", acc_precision);
    return launch_pairs(tbase, rbase, desc, npairs, max_side, min_std, min_mean_diff, acc_precision, out_max,
                        out_argmax, surfaces, surf_off, nullptr, nullptr, 0, static_cast<hipStream_t>(stream));
}

extern "C" int svamd_scan_fold(const double *corr_max, const uint8_t *is_centre, const uint8_t *skip,
                               const int64_t *win_off, int64_t nwin, double threshold, double centre_tol,
                               int64_t *best_pair, double *best_val, double *min_margin, void *stream)
{
    SV_REQUIRE(nwin >= 0, SVAMD_ERR_INVALID, "scan_fold: negative window count");
    if (nwin == 0)
        return SVAMD_OK;
    SV_REQUIRE(corr_max && is_centre && win_off && best_pair && best_val, SVAMD_ERR_INVALID,
               "scan_fold: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(scan_fold_kernel, dim3((unsigned)((nwin + 255) / 256)), dim3(256), 0, st, corr_max,
                       is_centre, skip, win_off, nwin, threshold, centre_tol, best_pair, best_val,
                       min_margin);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

#ifdef SV_PAIR_STAMPS
extern "C" int svamd_debug_pair_stamps(unsigned long long *out8, int reset)
{
    if (reset) {
        unsigned long long zero[8] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(pair_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : 4;
    }
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(pair_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 4;
}
#endif
