// Dense band scan: zero-mean normalised cross-correlation of every win_w x win_h
// window of the left image against the same-row boxes of the right image at all
// horizontal offsets d in [-max_disp, +max_disp], with the per-window arg-max or
// selection rule fused in.
//
// Arithmetic.  With l = L - 128 and r = R - 128 (integers in [-128, 127]) and
// n = win_w * win_h <= 1024, every partial sum of sum(l * r) is an integer of
// magnitude < 2^24, so the float32 fused multiply-adds below are EXACT.  The
// score is then formed in float64 from exact integers,
//     ncc = (n * sum(l r) - sum(l) sum(r)) / sqrt(A_L * A_R),  A = n sum(x^2) - sum(x)^2,
// where sum(l), sum(r), A_L, A_R come from summed-area tables (kernel K1) through
// a per-position table of window sums and 1/sqrt(A).
//
// Mapping.  A workgroup is 16 wavefronts and owns a tile of 16 window rows x 256
// window columns.  Its left tile and the matching band of the right image are
// staged once in LDS as float32.  Wavefront w owns window row w; lane i owns the
// 4 consecutive windows 4i..4i+3 and, per pass, 8 consecutive offsets: 32
// accumulators per lane, held as 16 register pairs (two neighbouring windows
// each) and updated by packed FMAs.  For each image row the lane walks the 34 columns its 4
// windows span: one ds_read_b128 per 4 left pixels, one per 4 new right pixels,
// up to 128 FMAs in between (about 50 FMAs per LDS read instruction), so the
// loop is bound by the vector ALU, not by LDS.  Offsets are visited in ascending
// order inside a lane, so the sequential selection rule needs no cross-lane
// traffic at all; only the final results are written.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int BAND_X = 4;                 // windows per lane
constexpr int BAND_D = 8;                 // offsets per pass
constexpr int BAND_TX = WAVE * BAND_X;    // window columns per tile
#ifndef SVAMD_BAND_WAVES
#define SVAMD_BAND_WAVES 16
#endif
constexpr int BAND_WAVES = SVAMD_BAND_WAVES;  // window rows per tile
constexpr int BAND_THREADS = BAND_WAVES * WAVE;
constexpr int BAND_LDS_LIMIT = 160 * 1024 - 256;   // dynamic LDS; the rest holds the pace counters

struct BandGeom {
    int H, W, win_w, win_h, stride_y, max_disp, nd;
    int nwx, nwy, nsx;        // window grid; nsx = W - win_w + 1 = columns of the stats tables
    int mode;
    double thr, tol;
    int row_begin, row_end;
    int lpitch, rpitch, tile_rows, dchunk;
    int tile_wrows;           // window rows per tile (<= BAND_WAVES; fewer when rows are strided)
};

// ---- per-position window statistics from the summed-area tables -----------------
__global__ void __launch_bounds__(256)
window_stats_kernel(const uint32_t *__restrict__ s1, const uint32_t *__restrict__ s2, int H, int W, int win_w,
                    int win_h, int32_t *__restrict__ wsum, double *__restrict__ winv)
{
    const int nsx = W - win_w + 1, nsy = H - win_h + 1;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nsx * nsy)
        return;
    const int y = (int)(idx / nsx), x = (int)(idx - (int64_t)y * nsx);
    const int64_t tp = (int64_t)W + 1;
    const int64_t a = (int64_t)y * tp + x, b = (int64_t)(y + win_h) * tp + x;
    // unsigned wrap-around differences: exact because the true sums fit 32 bits
    const uint32_t d1 = s1[b + win_w] - s1[a + win_w] - s1[b] + s1[a];
    const uint32_t d2 = s2[b + win_w] - s2[a + win_w] - s2[b] + s2[a];
    const int64_t sum = (int32_t)d1;
    const int64_t n = (int64_t)win_w * win_h;
    const int64_t A = n * (int64_t)d2 - sum * sum;
    wsum[idx] = (int32_t)sum;
    winv[idx] = A > 0 ? 1.0 / sqrt((double)A) : NAN;  // constant window: std == 0 -> NaN score
}

// ---- LDS tile staging ---------------------------------------------------------------
// dst[r][c] = img[row0 + r][col0 + c] - 128 as float32, 0 outside the image.
__device__ __forceinline__ void stage_tile(const uint8_t *__restrict__ img, int H, int W, int row0,
                                           int nrows, int col0, int pitch, float *__restrict__ dst)
{
    const int quads = pitch / 4;
    for (int i = threadIdx.x; i < nrows * quads; i += BAND_THREADS) {
        const int r = i / quads, q = i - r * quads;
        const int y = row0 + r, x = col0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H) {
            const uint8_t *src = img + (int64_t)y * W;
            if (x >= 0 && x + 0 < W) v.x = (float)src[x + 0] - 128.f;
            if (x + 1 >= 0 && x + 1 < W) v.y = (float)src[x + 1] - 128.f;
            if (x + 2 >= 0 && x + 2 < W) v.z = (float)src[x + 2] - 128.f;
            if (x + 3 >= 0 && x + 3 < W) v.w = (float)src[x + 3] - 128.f;
        }
        *reinterpret_cast<float4 *>(dst + r * pitch + 4 * q) = v;
    }
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// acc.x += l * r; acc.y += l * r with l = half LH of lpair, r = half RH of rpair:
// one v_pk_fma_f32 whose two multiplicands are broadcast through op_sel, so the
// operands are used where the 16-byte LDS reads left them (no register moves).
template <int LH, int RH>
__device__ __forceinline__ void pk_fma_bcast(float2v &acc, const float2v &lpair, const float2v &rpair)
{
    if constexpr (LH == 0 && RH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
    else if constexpr (LH == 0 && RH == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
    else if constexpr (LH == 1 && RH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
}

struct Quad {           // four consecutive pixels as two register pairs
    float2v lo, hi;
};

template <bool ALIGNED>
__device__ __forceinline__ Quad load_quad(const float *src)
{
    Quad q;
    if constexpr (ALIGNED) {
        const float4v v = *reinterpret_cast<const float4v *>(src);
        q.lo = v.xy;
        q.hi = v.zw;
    } else {
        q.lo = float2v{src[0], src[1]};
        q.hi = float2v{src[2], src[3]};
    }
    return q;
}

template <int K>
__device__ __forceinline__ const float2v &pair_of(const Quad *q)
{
    return (K & 2) ? q[K / 4].hi : q[K / 4].lo;
}

template <int K>
__device__ __forceinline__ float elem_of(const Quad *q)
{
    const float2v &p = pair_of<K>(q);
    return (K & 1) ? p.y : p.x;
}

// One left column u against the D right columns u .. u+D-1, for window pair P.
template <int WIN_W, int D, int U, int P, int DD = 0>
__device__ __forceinline__ void column_step(float2v (&acc)[BAND_X / 2][D], const Quad &lq, const Quad *rq)
{
    if constexpr (DD < D) {
        constexpr bool lo = 2 * P <= U && U < 2 * P + WIN_W;          // window 2P covers column U
        constexpr bool hi = 2 * P + 1 <= U && U < 2 * P + 1 + WIN_W;  // window 2P+1 covers it
        constexpr int I = U & 3, K = U + DD;
        const float2v &lp2 = (I & 2) ? lq.hi : lq.lo;
        if constexpr (lo && hi) {
            pk_fma_bcast<(I & 1), (K & 1)>(acc[P][DD], lp2, pair_of<K>(rq));
        } else if constexpr (lo) {
            acc[P][DD].x = fmaf((I & 1) ? lp2.y : lp2.x, elem_of<K>(rq), acc[P][DD].x);
        } else if constexpr (hi) {
            acc[P][DD].y = fmaf((I & 1) ? lp2.y : lp2.x, elem_of<K>(rq), acc[P][DD].y);
        }
        column_step<WIN_W, D, U, P, DD + 1>(acc, lq, rq);
    }
}

template <int WIN_W, int D, int C, int I = 0>
__device__ __forceinline__ void quad_step(float2v (&acc)[BAND_X / 2][D], const Quad &lq, const Quad *rq)
{
    constexpr int NU = WIN_W + BAND_X - 1;
    if constexpr (I < 4) {
        constexpr int U = 4 * C + I;
        if constexpr (U < NU) {
            column_step<WIN_W, D, U, 0>(acc, lq, rq);
            column_step<WIN_W, D, U, 1>(acc, lq, rq);
        }
        quad_step<WIN_W, D, C, I + 1>(acc, lq, rq);
    }
}

template <int WIN_W, int D, bool ALIGNED, int C = 0>
__device__ __forceinline__ void row_step(float2v (&acc)[BAND_X / 2][D], const float *lp, const float *rp,
                                         const Quad &l_head, Quad *rq)
{
    constexpr int NU = WIN_W + BAND_X - 1;       // left columns touched by the lane's windows
    constexpr int NCL = (NU + 3) / 4;            // ... in groups of four
    constexpr int NCR = (NU + D - 1 + 3) / 4;    // right columns in groups of four
    constexpr int AHEAD = (D + 2) / 4;           // right groups needed beyond group C
    if constexpr (C < NCL) {
        Quad lq;
        if constexpr (C == 0)
            lq = l_head;                         // prefetched during the previous row
        else
            lq = load_quad<true>(lp + 4 * C);
        if constexpr (C + AHEAD < NCR && C + AHEAD >= AHEAD + (AHEAD == 0 ? 1 : 0))
            rq[C + AHEAD] = load_quad<ALIGNED>(rp + 4 * (C + AHEAD));
        quad_step<WIN_W, D, C>(acc, lq, rq);
        // keeps the scheduler from hoisting the LDS reads of later groups over this one:
        // with them in flight early the row needs more registers than there are (spills)
        __builtin_amdgcn_sched_barrier(0);
        row_step<WIN_W, D, ALIGNED, C + 1>(acc, lp, rp, l_head, rq);
    }
}

// acc[p][d] = (window 2p, window 2p+1) at offset d:
//   acc[x][d] += sum over `rows` image rows and the window's columns of
//                L[row][x + b] * R[row][x + b + d],  b < WIN_W
// lp / rp point at the lane's first left column / first right column (offset d = 0
// of this pass) in the first row.  Two neighbouring windows share every product
// except at their outer columns, so the bulk of the work is issued as packed
// v_pk_fma_f32 with both multiplicands broadcast, which lets a single wavefront
// keep its SIMD's FMA pipe full; the outer columns use scalar FMAs.  The first
// LDS reads of each row are issued during the previous row's arithmetic.
template <int WIN_W, int D, bool ALIGNED>
__device__ __forceinline__ void accumulate(const float *__restrict__ lp, const float *__restrict__ rp,
                                           int lpitch, int rpitch, int rows,
                                           float2v (&acc)[BAND_X / 2][D])
{
    constexpr int NU = WIN_W + BAND_X - 1;
    constexpr int NCR = (NU + D - 1 + 3) / 4;
    constexpr int AHEAD = (D + 2) / 4;
    constexpr int HEAD = AHEAD == 0 ? 1 : AHEAD;   // right quads needed before the first FMA
    // Two sets of row registers (A, B) take turns, two rows per loop iteration: while
    // one row is processed from set A, the first reads of the next row land in set B,
    // and no register has to be copied.  The last row re-reads its own head (no read
    // past the tile).
    Quad lA = load_quad<true>(lp), lB;
    Quad rqA[NCR], rqB[NCR];
#pragma unroll
    for (int c = 0; c < HEAD; ++c)
        rqA[c] = load_quad<ALIGNED>(rp + 4 * c);
    int a = 0;
#pragma unroll 1
    for (; a + 2 <= rows; a += 2) {
        lB = load_quad<true>(lp + lpitch);
#pragma unroll
        for (int c = 0; c < HEAD; ++c)
            rqB[c] = load_quad<ALIGNED>(rp + rpitch + 4 * c);
        row_step<WIN_W, D, ALIGNED>(acc, lp, rp, lA, rqA);
        lp += lpitch;
        rp += rpitch;
        const float *lnext = a + 2 < rows ? lp + lpitch : lp;
        const float *rnext = a + 2 < rows ? rp + rpitch : rp;
        lA = load_quad<true>(lnext);
#pragma unroll
        for (int c = 0; c < HEAD; ++c)
            rqA[c] = load_quad<ALIGNED>(rnext + 4 * c);
        row_step<WIN_W, D, ALIGNED>(acc, lp, rp, lB, rqB);
        lp = lnext;
        rp = rnext;
    }
    if (a < rows)
        row_step<WIN_W, D, ALIGNED>(acc, lp, rp, lA, rqA);
}

// acc.x += l.x * r; acc.y += l.y * r with r = half RH of rpair: the left operand
// is taken as a pair (one value per window), the right one is broadcast.
template <int RH>
__device__ __forceinline__ void pk_fma_rbcast(float2v &acc, const float2v &lpair, const float2v &rpair)
{
    if constexpr (RH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(lpair), "v"(rpair));
}

// Element K (0 .. 11) of three consecutive quads, as the register pair that holds it.
template <int K>
__device__ __forceinline__ const float2v &pair_of3(const Quad &q0, const Quad &q1, const Quad &q2)
{
    const Quad &q = K < 4 ? q0 : (K < 8 ? q1 : q2);
    return (K & 2) ? q.hi : q.lo;
}

template <int D, int DD = 0, typename F>
__device__ __forceinline__ void for_each_offset(F &f)
{
    if constexpr (DD < D) {
        f(std::integral_constant<int, DD>{});
        for_each_offset<D, DD + 1>(f);
    }
}

// Window width known only at run time: the four left columns ubase .. ubase + 3 of
// one image row against the D offsets.  r0, r1, r2: right columns ubase .. ubase + 11.
// INTERIOR: all four windows of the lane cover all four columns (both packed
// multiplicands broadcast, as in the fixed-width kernels).  Otherwise the left
// value is masked per window - window x covers columns x .. x + win_w - 1.
template <int D, bool INTERIOR>
__device__ __forceinline__ void quad_any(float2v (&acc)[BAND_X / 2][D], const Quad &lq, const Quad &r0,
                                         const Quad &r1, const Quad &r2, int ubase, int win_w)
{
    auto column = [&](auto I_) {
        constexpr int I = decltype(I_)::value;
        const float2v &lp2 = (I & 2) ? lq.hi : lq.lo;
        auto offsets = [&](auto fma) { for_each_offset<D>(fma); };
        if constexpr (INTERIOR) {
            offsets([&](auto DD_) {
                constexpr int DD = decltype(DD_)::value, K = I + DD;
                pk_fma_bcast<(I & 1), (K & 1)>(acc[0][DD], lp2, pair_of3<K>(r0, r1, r2));
                pk_fma_bcast<(I & 1), (K & 1)>(acc[1][DD], lp2, pair_of3<K>(r0, r1, r2));
            });
        } else {
            const int u = ubase + I;
            if (u < win_w + BAND_X - 1) {                       // wave-uniform
                const float l = (I & 1) ? lp2.y : lp2.x;
                float2v lv[BAND_X / 2];
#pragma unroll
                for (int p = 0; p < BAND_X / 2; ++p) {
                    lv[p].x = (u >= 2 * p && u < 2 * p + win_w) ? l : 0.f;
                    lv[p].y = (u >= 2 * p + 1 && u < 2 * p + 1 + win_w) ? l : 0.f;
                }
                offsets([&](auto DD_) {
                    constexpr int DD = decltype(DD_)::value, K = I + DD;
                    pk_fma_rbcast<(K & 1)>(acc[0][DD], lv[0], pair_of3<K>(r0, r1, r2));
                    pk_fma_rbcast<(K & 1)>(acc[1][DD], lv[1], pair_of3<K>(r0, r1, r2));
                });
            }
        }
    };
    column(std::integral_constant<int, 0>{});
    column(std::integral_constant<int, 1>{});
    column(std::integral_constant<int, 2>{});
    column(std::integral_constant<int, 3>{});
}

// accumulate() for a window width known only at run time (any width the tile
// geometry admits): the same register tile, the row walked in groups of four
// columns with the three right quads a group needs rotating through registers.
template <int D, bool ALIGNED>
__device__ __forceinline__ void accumulate_any(const float *__restrict__ lp, const float *__restrict__ rp,
                                               int lpitch, int rpitch, int rows, int win_w,
                                               float2v (&acc)[BAND_X / 2][D])
{
    static_assert(D <= 9, "three right quads cover offsets 0..8 only");
    const int groups = (win_w + BAND_X - 1 + 3) / 4;
    // groups c = 1 .. last_interior lie inside all four windows (3 <= 4c, 4c + 3 <= win_w - 1)
    const int last_interior = win_w >= 8 ? (win_w - 4) / 4 : 0;
    // The row has three phases, none with a branch inside a loop over accumulators:
    // the first group (masked), whole rotations of three interior groups (the three
    // right quads change roles without copies), and the remaining groups (masked
    // code, which also serves interior groups, with the quads moved down by one).
    const int rotations = last_interior / 3;
#pragma unroll 1
    for (int a = 0; a < rows; ++a) {
        Quad r0 = load_quad<ALIGNED>(rp), r1 = load_quad<ALIGNED>(rp + 4), r2 = load_quad<ALIGNED>(rp + 8);
        quad_any<D, false>(acc, load_quad<true>(lp), r0, r1, r2, 0, win_w);
        r0 = r1;
        r1 = r2;
        int c = 1;
        for (int t = 0; t < rotations; ++t, c += 3) {
            r2 = load_quad<ALIGNED>(rp + 4 * (c + 2));
            quad_any<D, true>(acc, load_quad<true>(lp + 4 * c), r0, r1, r2, 4 * c, win_w);
            r0 = load_quad<ALIGNED>(rp + 4 * (c + 3));
            quad_any<D, true>(acc, load_quad<true>(lp + 4 * c + 4), r1, r2, r0, 4 * c + 4, win_w);
            r1 = load_quad<ALIGNED>(rp + 4 * (c + 4));
            quad_any<D, true>(acc, load_quad<true>(lp + 4 * c + 8), r2, r0, r1, 4 * c + 8, win_w);
        }
        for (; c < groups; ++c) {
            r2 = load_quad<ALIGNED>(rp + 4 * (c + 2));
            quad_any<D, false>(acc, load_quad<true>(lp + 4 * c), r0, r1, r2, 4 * c, win_w);
            r0 = r1;
            r1 = r2;
        }
        lp += lpitch;
        rp += rpitch;
    }
}

// WIN_W > 0: the width is a compile-time constant (fully unrolled row); WIN_W == 0:
// the width comes from the geometry at run time.
template <int WIN_W, int D, bool ALIGNED>
__device__ __forceinline__ void accumulate_rows(const float *__restrict__ lp, const float *__restrict__ rp,
                                                int lpitch, int rpitch, int rows, int win_w,
                                                float2v (&acc)[BAND_X / 2][D])
{
    if constexpr (WIN_W > 0)
        accumulate<WIN_W, D, ALIGNED>(lp, rp, lpitch, rpitch, rows, acc);
    else
        accumulate_any<D, ALIGNED>(lp, rp, lpitch, rpitch, rows, win_w, acc);
}

struct LaneState {
    double best[BAND_X];
    int arg[BAND_X];
};

// Turns the accumulators of offsets D0 .. D0 + DN - 1 (of the D of a pass) of each
// window into scores and feeds the selection.  k0: offset index (0 .. nd-1) of
// acc[.][0]; centre_pass: this pass evaluates the centre offset ahead of the
// ascending sweep (selection-rule mode only).
template <int D, int D0, int DN>
__device__ __forceinline__ void finish_range(const BandGeom &g, const float2v (&acc)[BAND_X / 2][D], int k0,
                                             int kend, bool centre_pass, int wy, int xbase,
                                             const int32_t *__restrict__ wsum_l,
                                             const double *__restrict__ winv_l,
                                             const int32_t *__restrict__ wsum_r,
                                             const double *__restrict__ winv_r, LaneState &st,
                                             double *__restrict__ scores)
{
    const double n = (double)(g.win_w * g.win_h);
    const int64_t srow = (int64_t)wy * g.stride_y * g.nsx;
    // window x at offset d looks at the right-image box xbase + x + d + k0 - max_disp:
    // BAND_X + DN - 1 distinct boxes serve the lane's BAND_X * DN evaluations
    constexpr int NJ = BAND_X + DN - 1;
    int32_t r_sum[NJ];
    double r_inv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int xr = xbase + D0 + j + k0 - g.max_disp;
        const bool in = xr >= 0 && xr < g.nsx;
        r_sum[j] = in ? wsum_r[srow + xr] : 0;
        r_inv[j] = in ? winv_r[srow + xr] : NAN;       // outside the image: no score
    }
#pragma unroll
    for (int x = 0; x < BAND_X; ++x) {
        const int xw = xbase + x;
        const bool wvalid = xw < g.nwx;
        // the window's own statistics are re-read every time (cache hits) rather than
        // held in registers across the accumulation
        const double s1l = wvalid ? (double)wsum_l[srow + xw] : 0.0;   // sum of (L - 128) over the window
        const double invl = wvalid ? winv_l[srow + xw] : NAN;          // 1 / sqrt(A_L)
#pragma unroll
        for (int dd = 0; dd < DN; ++dd) {
            const int d = D0 + dd;
            const int k = k0 + d;
            const bool kvalid = k < kend;
            double s = NAN;
            if (wvalid && kvalid) {
                const float cross = (x & 1) ? acc[x / 2][d].y : acc[x / 2][d].x;
                const double num = fma(n, (double)cross, -(s1l * (double)r_sum[x + dd]));
                s = num * invl * r_inv[x + dd];
            }
            if (scores && wvalid && kvalid)
                scores[((int64_t)wy * g.nwx + xw) * g.nd + k] = s;
            if (!kvalid)
                continue;
            if (g.mode == SVAMD_BAND_ARGMAX) {
                if (s > st.best[x]) {
                    st.best[x] = s;
                    st.arg[x] = k - g.max_disp;
                }
            } else if (centre_pass) {
                if (s > g.thr && s >= st.best[x]) {
                    st.best[x] = s;
                    st.arg[x] = 0;
                }
            } else if (k != g.max_disp) {
                if (s > g.thr && s > st.best[x] + g.tol) {
                    st.best[x] = s;
                    st.arg[x] = k - g.max_disp;
                }
            }
        }
    }
}

// The scores of one pass.  Eight offsets are finished four at a time: each half
// needs the statistics of 7 right-image boxes instead of 11 at once, and half the
// accumulators are dead before the second half's statistics arrive, which keeps
// the epilogue inside the register budget the accumulation loop leaves (every
// window still sees its offsets in ascending order).
template <int D>
__device__ __forceinline__ void finish_pass(const BandGeom &g, const float2v (&acc)[BAND_X / 2][D], int k0,
                                            int kend, bool centre_pass, int wy, int xbase,
                                            const int32_t *__restrict__ wsum_l,
                                            const double *__restrict__ winv_l,
                                            const int32_t *__restrict__ wsum_r,
                                            const double *__restrict__ winv_r, LaneState &st,
                                            double *__restrict__ scores)
{
    if constexpr (D == 8) {
        finish_range<D, 0, 4>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, scores);
        __builtin_amdgcn_sched_barrier(0);
        finish_range<D, 4, 4>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, scores);
    } else {
        finish_range<D, 0, D>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, scores);
    }
}

// Pace keeping between the wavefronts that share a SIMD.  The hardware issues
// from the oldest ready wavefront first, so without intervention the four
// wavefronts of a SIMD run one after the other, and the last one ends up alone on
// the SIMD - where a single wavefront cannot keep the packed-FMA pipe full.  Each
// wavefront therefore publishes how many passes it has finished and lowers its own
// issue priority by its lead over the slowest wavefront of its SIMD, which keeps
// all four within a pass or two of each other until the end of the tile.
// Wavefronts w, w+4, w+8, ... of a workgroup share a SIMD.
__device__ __forceinline__ void keep_pace(volatile int *progress, int wave, int done)
{
    progress[wave] = done;
    int slowest = done;
#pragma unroll
    for (int m = (wave & 3); m < BAND_WAVES; m += 4)
        slowest = min(slowest, progress[m]);
    const int lead = done - slowest;
    if (lead <= 0)
        __builtin_amdgcn_s_setprio(3);
    else if (lead == 1)
        __builtin_amdgcn_s_setprio(2);
    else if (lead == 2)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(0);
}

// Diagnostic build only (-DSVAMD_BAND_STAMPS, see profiles/stamp_band.py): the
// `scores` argument is reinterpreted as a buffer of 40 wall-clock stamps per
// wavefront (kernel entry, then the end of every 8-offset pass) and no scores
// are written.
#ifdef SVAMD_BAND_STAMPS
#define SV_STAMP(slot)                                                                           \
    do {                                                                                         \
        if (scores && lane == 0)                                                                 \
            reinterpret_cast<long long *>(scores)[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * \
                                                   BAND_WAVES + wave) * 40 + (slot)] = wall_clock64(); \
    } while (0)
#define SV_SCORES_OUT nullptr
#else
#define SV_STAMP(slot) do { } while (0)
#define SV_SCORES_OUT scores
#endif

template <int WIN_W>
__global__ void __launch_bounds__(BAND_THREADS)
band_scan_kernel(BandGeom g, const uint8_t *__restrict__ left, const uint8_t *__restrict__ right,
                 const int32_t *__restrict__ wsum_l, const double *__restrict__ winv_l,
                 const int32_t *__restrict__ wsum_r, const double *__restrict__ winv_r,
                 int32_t *__restrict__ out_disp, double *__restrict__ out_val, double *__restrict__ scores)
{
    extern __shared__ __align__(16) float tile[];
    __shared__ int progress[BAND_WAVES];
    float *Lt = tile;
    float *Rt = tile + g.tile_rows * g.lpitch;

    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int x0 = blockIdx.x * BAND_TX;                      // first window column of the tile
    const int wy0 = g.row_begin + blockIdx.y * g.tile_wrows;  // first window row of the tile
    const int y0 = wy0 * g.stride_y;                          // first image row of the tile
    const int wy = wy0 + wave;
    const bool active = wave < g.tile_wrows && wy < g.row_end;  // wave-uniform
    const int xbase = x0 + BAND_X * lane;
    const int lrow = wave * g.stride_y;                       // tile row of this wave's first image row

    if (lane == 0)
        progress[wave] = active ? 0 : 0x7fffffff;   // idle wavefronts never hold anyone back
    int passes = 0;
    __builtin_amdgcn_s_setprio(3);                  // everyone starts level; leaders step down

    LaneState st;
#pragma unroll
    for (int x = 0; x < BAND_X; ++x) {
        st.best[x] = g.mode == SVAMD_BAND_ARGMAX ? -INFINITY : 0.0;
        st.arg[x] = g.mode == SVAMD_BAND_ARGMAX ? SVAMD_BAND_NO_MATCH : 0;
    }

    SV_STAMP(0);
    stage_tile(left, g.H, g.W, y0, g.tile_rows, x0, g.lpitch, Lt);
    const float *lp = Lt + lrow * g.lpitch + BAND_X * lane;

    // The centre offset is judged first (selection-rule mode).  When the whole offset
    // range sits in LDS at once it is read from there, otherwise the right image is
    // staged at d = 0 for it.
    const bool whole_band = g.dchunk >= g.nd;
    if (g.mode == SVAMD_BAND_FOLD && !whole_band) {
        stage_tile(right, g.H, g.W, y0, g.tile_rows, x0, g.lpitch, Rt);
        __syncthreads();
        if (active) {
            float2v acc[BAND_X / 2][1] = {};
            accumulate_rows<WIN_W, 1, true>(lp, Rt + lrow * g.lpitch + BAND_X * lane, g.lpitch, g.lpitch,
                                            g.win_h, g.win_w, acc);
            finish_pass<1>(g, acc, g.max_disp, g.nd, true, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, nullptr);
        }
    }

    for (int k0 = 0; k0 < g.nd; k0 += g.dchunk) {
        const int kend = min(k0 + g.dchunk, g.nd);
        __syncthreads();  // previous contents of Rt are no longer read
        stage_tile(right, g.H, g.W, y0, g.tile_rows, x0 - g.max_disp + k0, g.rpitch, Rt);
        __syncthreads();
        if (!active)
            continue;
        const float *rp = Rt + lrow * g.rpitch + BAND_X * lane;
        if (g.mode == SVAMD_BAND_FOLD && whole_band) {
            float2v acc[BAND_X / 2][1] = {};
            if (g.max_disp & 3)
                accumulate_rows<WIN_W, 1, false>(lp, rp + g.max_disp, g.lpitch, g.rpitch, g.win_h, g.win_w, acc);
            else
                accumulate_rows<WIN_W, 1, true>(lp, rp + g.max_disp, g.lpitch, g.rpitch, g.win_h, g.win_w, acc);
            finish_pass<1>(g, acc, g.max_disp, g.nd, true, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, nullptr);
        }
        int k = k0;
        for (; k + BAND_D <= kend; k += BAND_D) {
            float2v acc[BAND_X / 2][BAND_D] = {};
            accumulate_rows<WIN_W, BAND_D, true>(lp, rp + (k - k0), g.lpitch, g.rpitch, g.win_h, g.win_w, acc);
            finish_pass<BAND_D>(g, acc, k, kend, false, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, SV_SCORES_OUT);
            keep_pace(progress, wave, ++passes);
            SV_STAMP(1 + k / BAND_D);
        }
        for (; k < kend; ++k) {  // fewer than 8 offsets left in this chunk
            float2v acc[BAND_X / 2][1] = {};
            accumulate_rows<WIN_W, 1, false>(lp, rp + (k - k0), g.lpitch, g.rpitch, g.win_h, g.win_w, acc);
            finish_pass<1>(g, acc, k, kend, false, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st, SV_SCORES_OUT);
        }
    }

    if (lane == 0)
        progress[wave] = 0x7fffffff;
    if (active) {
#pragma unroll
        for (int x = 0; x < BAND_X; ++x) {
            const int xw = xbase + x;
            if (xw < g.nwx) {
                const int64_t o = (int64_t)wy * g.nwx + xw;
                out_disp[o] = st.arg[x];
                out_val[o] = (g.mode == SVAMD_BAND_ARGMAX && st.arg[x] == SVAMD_BAND_NO_MATCH) ? NAN
                                                                                               : st.best[x];
            }
        }
    }
}

struct BandWs {
    size_t n_sat, n_stat;
};
static BandWs band_ws(int H, int W, int win_w, int win_h)
{
    BandWs b;
    b.n_sat = (size_t)(H + 1) * (W + 1);
    b.n_stat = (size_t)(H - win_h + 1) * (W - win_w + 1);
    return b;
}

template <int WIN_W>
static int band_launch(const BandGeom &g, size_t lds, dim3 grid, hipStream_t st, const uint8_t *left,
                       const uint8_t *right, const int32_t *wl, const double *il, const int32_t *wr,
                       const double *ir, int32_t *out_disp, double *out_val, double *scores)
{
    // per function and per device: set before every launch (a process may drive several devices)
    SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&band_scan_kernel<WIN_W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BAND_LDS_LIMIT));
    hipLaunchKernelGGL((band_scan_kernel<WIN_W>), grid, dim3(BAND_THREADS), lds, st, g, left, right, wl, il,
                       wr, ir, out_disp, out_val, scores);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

}  // namespace svamd

using namespace svamd;

extern "C" size_t svamd_band_scan_u8_workspace_bytes(int H, int W, int win_w, int win_h)
{
    if (H <= 0 || W <= 0 || win_w <= 0 || win_h <= 0 || win_w > W || win_h > H)
        return 0;
    BandWs b = band_ws(H, W, win_w, win_h);
    return 4 * align_up(b.n_sat * 4, 256) + 2 * align_up(b.n_stat * 4, 256) + 2 * align_up(b.n_stat * 8, 256);
}

namespace {
struct BandTables {
    uint32_t *s1l, *s2l, *s1r, *s2r;
    int32_t *wl, *wr;
    double *il, *ir;
};
BandTables band_tables(void *workspace, int H, int W, int win_w, int win_h)
{
    BandWs b = band_ws(H, W, win_w, win_h);
    Carver ws(workspace);
    BandTables t;
    t.s1l = ws.take<uint32_t>(b.n_sat);
    t.s2l = ws.take<uint32_t>(b.n_sat);
    t.s1r = ws.take<uint32_t>(b.n_sat);
    t.s2r = ws.take<uint32_t>(b.n_sat);
    t.wl = ws.take<int32_t>(b.n_stat);
    t.wr = ws.take<int32_t>(b.n_stat);
    t.il = ws.take<double>(b.n_stat);
    t.ir = ws.take<double>(b.n_stat);
    return t;
}
}  // namespace

extern "C" int svamd_band_prepare_u8(const uint8_t *left, const uint8_t *right, int H, int W, int win_w,
                                     int win_h, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(left && right, SVAMD_ERR_INVALID, "band_prepare_u8: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && win_w > 0 && win_h > 0, SVAMD_ERR_INVALID, "band_prepare_u8: non-positive size");
    SV_REQUIRE(win_w <= W && win_h <= H, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE((int64_t)win_w * win_h <= 1024, SVAMD_ERR_LIMIT,
               "band_prepare_u8: window area %d exceeds 1024", win_w * win_h);
    SV_REQUIRE((int64_t)(H + 1) * (W + 1) < (1LL << 31), SVAMD_ERR_LIMIT, "band_prepare_u8: image too large");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_band_scan_u8_workspace_bytes(H, W, win_w, win_h),
               SVAMD_ERR_WORKSPACE, "band_prepare_u8: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BandWs b = band_ws(H, W, win_w, win_h);
    BandTables t = band_tables(workspace, H, W, win_w, win_h);
    int rc;
    // (the window statistics are written after the tables: until then their buffer is the tables' scratch)
    const size_t spare = b.n_stat * sizeof(double);
    if ((rc = sat2d_launch<uint8_t, uint32_t>(left, H, W, W, 128u, t.s1l, t.s2l, st, t.il, spare)) != SVAMD_OK) return rc;
    if ((rc = sat2d_launch<uint8_t, uint32_t>(right, H, W, W, 128u, t.s1r, t.s2r, st, t.ir, spare)) != SVAMD_OK) return rc;
    const unsigned sg = (unsigned)((b.n_stat + 255) / 256);
    hipLaunchKernelGGL(window_stats_kernel, dim3(sg), dim3(256), 0, st, t.s1l, t.s2l, H, W, win_w, win_h,
                       t.wl, t.il);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_stats_kernel, dim3(sg), dim3(256), 0, st, t.s1r, t.s2r, H, W, win_w, win_h,
                       t.wr, t.ir);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" int svamd_band_scan_u8(const uint8_t *left, const uint8_t *right, int H, int W, int win_w,
                                  int win_h, int stride_x, int stride_y, int max_disp, int mode,
                                  double threshold, double centre_tol, int row_begin, int row_end,
                                  int flags, int32_t *out_disp, double *out_val, double *scores,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(left && right && out_disp && out_val, SVAMD_ERR_INVALID, "band_scan_u8: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && win_w > 0 && win_h > 0 && max_disp >= 0 && stride_y > 0, SVAMD_ERR_INVALID,
               "band_scan_u8: non-positive size");
    SV_REQUIRE(win_w <= W && win_h <= H, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE(mode == SVAMD_BAND_ARGMAX || mode == SVAMD_BAND_FOLD, SVAMD_ERR_INVALID,
               "band_scan_u8: unknown mode %d", mode);
    SV_REQUIRE(stride_x == 1, SVAMD_ERR_LIMIT, "band_scan_u8: only stride_x == 1 is supported (got %d)",
               stride_x);
    SV_REQUIRE((int64_t)win_w * win_h <= 1024, SVAMD_ERR_LIMIT,
               "band_scan_u8: window area %d exceeds 1024 (float32 accumulation would not be exact)",
               win_w * win_h);
    SV_REQUIRE((int64_t)(H + 1) * (W + 1) < (1LL << 31), SVAMD_ERR_LIMIT, "band_scan_u8: image too large");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_band_scan_u8_workspace_bytes(H, W, win_w, win_h),
               SVAMD_ERR_WORKSPACE, "band_scan_u8: workspace missing, misaligned or too small");

    BandGeom g;
    g.H = H; g.W = W; g.win_w = win_w; g.win_h = win_h; g.stride_y = stride_y; g.max_disp = max_disp;
    g.nd = 2 * max_disp + 1;
    g.nwx = W - win_w + 1;
    g.nwy = (H - win_h) / stride_y + 1;
    g.nsx = W - win_w + 1;
    g.mode = mode; g.thr = threshold; g.tol = centre_tol;
    SV_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= g.nwy, SVAMD_ERR_INVALID,
               "band_scan_u8: row range [%d, %d) outside 0..%d", row_begin, row_end, g.nwy);
    g.row_begin = row_begin; g.row_end = row_end;
    if (row_begin == row_end)
        return SVAMD_OK;

    // Tile geometry.  Row pitches carry 4 floats of slack because the last
    // 4-wide LDS read of a lane may extend past the columns it uses.  As many
    // offsets as fit in LDS are staged per pass over the right image.
    const bool tuned = win_w == 7 || win_w == 11 || win_w == 15 || win_w == 21 || win_w == 25 || win_w == 31;
    const int slack = tuned ? 4 : 12;   // the run-time-width row walk reads up to three quads ahead
    g.lpitch = (int)align_up((size_t)(BAND_TX + win_w - 1), 4) + slack;
    size_t lds = 0;
    for (g.tile_wrows = BAND_WAVES; g.tile_wrows >= 1; --g.tile_wrows) {
        g.tile_rows = (g.tile_wrows - 1) * stride_y + win_h;
        g.dchunk = (int)align_up((size_t)g.nd, BAND_D);
        for (;;) {
            const int dcur = g.dchunk < g.nd ? g.dchunk : g.nd;
            g.rpitch = (int)align_up((size_t)(BAND_TX + win_w - 1 + dcur - 1), 4) + slack;
            if (g.rpitch < g.lpitch)
                g.rpitch = g.lpitch;  // the centre pass stages a left-shaped tile into Rt
            lds = (size_t)g.tile_rows * (g.lpitch + g.rpitch) * sizeof(float);
            if (lds <= (size_t)BAND_LDS_LIMIT || g.dchunk <= BAND_D)
                break;
            g.dchunk -= BAND_D;
        }
        if (lds <= (size_t)BAND_LDS_LIMIT)
            break;
    }
    SV_REQUIRE(lds <= (size_t)BAND_LDS_LIMIT, SVAMD_ERR_LIMIT,
               "band_scan_u8: window %dx%d with row stride %d does not fit the LDS tile", win_w, win_h,
               stride_y);

    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & SVAMD_BAND_PREPARED)) {
        int rc = svamd_band_prepare_u8(left, right, H, W, win_w, win_h, workspace, workspace_bytes, stream);
        if (rc != SVAMD_OK)
            return rc;
    }
    BandTables t = band_tables(workspace, H, W, win_w, win_h);
    const int32_t *wl = t.wl, *wr = t.wr;
    const double *il = t.il, *ir = t.ir;

    dim3 grid((g.nwx + BAND_TX - 1) / BAND_TX, (row_end - row_begin + g.tile_wrows - 1) / g.tile_wrows);
#define SV_BAND_CASE(WW)                                                                          \
    case WW:                                                                                      \
        return band_launch<WW>(g, lds, grid, st, left, right, wl, il, wr, ir, out_disp, out_val, scores)
    switch (win_w) {
        SV_BAND_CASE(7);
        SV_BAND_CASE(11);
        SV_BAND_CASE(15);
        SV_BAND_CASE(21);
        SV_BAND_CASE(25);
        SV_BAND_CASE(31);
    default:   // any other width: the kernel that takes it at run time
        return band_launch<0>(g, lds, grid, st, left, right, wl, il, wr, ir, out_disp, out_val, scores);
    }
#undef SV_BAND_CASE
}
