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
//
// Floating-point images (what the reference's downsample() produces for factors
// above 1) run through the same kernel with PIX = double: pixels are staged as
// float32 after the image mean is taken off, every lane additionally takes a constant
// of its own off the values it reads (about the mean of its windows / of the boxes
// of the pass), so that the float32 products it adds up are products of deviations,
// and the epilogue puts those constants back in float64.  Window sums come from
// float64 tables.  Each window records how close its closest decision came to
// flipping; windows within the caller's guard band are evaluated again in float64
// (band_recheck_kernel), so the selected offsets are those of a float64 evaluation.
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
// dst[r][c] = img[row0 + r][col0 + c] - centre as float32, 0 outside the image
// (centre = 128 for 8-bit pixels).
template <typename PIX>
__device__ __forceinline__ void stage_tile(const PIX *__restrict__ img, int H, int W, int row0,
                                           int nrows, int col0, int pitch, float *__restrict__ dst, double centre)
{
    const int quads = pitch / 4;
    for (int i = threadIdx.x; i < nrows * quads; i += BAND_THREADS) {
        const int r = i / quads, q = i - r * quads;
        const int y = row0 + r, x = col0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H) {
            const PIX *src = img + (int64_t)y * W;
            if constexpr (std::is_same<PIX, uint8_t>::value) {
                if (x >= 0 && x + 0 < W) v.x = (float)src[x + 0] - 128.f;
                if (x + 1 >= 0 && x + 1 < W) v.y = (float)src[x + 1] - 128.f;
                if (x + 2 >= 0 && x + 2 < W) v.z = (float)src[x + 2] - 128.f;
                if (x + 3 >= 0 && x + 3 < W) v.w = (float)src[x + 3] - 128.f;
            } else {
                if (x >= 0 && x + 0 < W) v.x = (float)(src[x + 0] - centre);
                if (x + 1 >= 0 && x + 1 < W) v.y = (float)(src[x + 1] - centre);
                if (x + 2 >= 0 && x + 2 < W) v.z = (float)(src[x + 2] - centre);
                if (x + 3 >= 0 && x + 3 < W) v.w = (float)(src[x + 3] - centre);
            }
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

// Floating-point images: the constants a lane takes off the left / right values it
// reads (FLT = false: 8-bit images, nothing is taken off and nothing is stored).
template <bool FLT>
struct LaneCentre {
};
template <>
struct LaneCentre<true> {
    float2v a, b;
};
template <bool FLT>
__device__ __forceinline__ void centre_left(Quad &q, const LaneCentre<FLT> &c)
{
    if constexpr (FLT) {
        q.lo -= c.a;
        q.hi -= c.a;
    }
}
template <bool FLT>
__device__ __forceinline__ void centre_right(Quad &q, const LaneCentre<FLT> &c)
{
    if constexpr (FLT) {
        q.lo -= c.b;
        q.hi -= c.b;
    }
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

template <int WIN_W, int D, bool ALIGNED, bool FLT, int C = 0>
__device__ __forceinline__ void row_step(float2v (&acc)[BAND_X / 2][D], const float *lp, const float *rp,
                                         const Quad &l_head, Quad *rq, const LaneCentre<FLT> &lc)
{
    constexpr int NU = WIN_W + BAND_X - 1;       // left columns touched by the lane's windows
    constexpr int NCL = (NU + 3) / 4;            // ... in groups of four
    constexpr int NCR = (NU + D - 1 + 3) / 4;    // right columns in groups of four
    constexpr int AHEAD = (D + 2) / 4;           // right groups needed beyond group C
    if constexpr (C < NCL) {
        Quad lq;
        if constexpr (C == 0) {
            lq = l_head;                         // prefetched during the previous row
            constexpr int HEAD = AHEAD == 0 ? 1 : AHEAD;
#pragma unroll
            for (int c = 0; c < HEAD; ++c)       // ... as were the first right quads
                centre_right<FLT>(rq[c], lc);
        } else {
            lq = load_quad<true>(lp + 4 * C);
        }
        centre_left<FLT>(lq, lc);
        if constexpr (C + AHEAD < NCR && C + AHEAD >= AHEAD + (AHEAD == 0 ? 1 : 0)) {
            rq[C + AHEAD] = load_quad<ALIGNED>(rp + 4 * (C + AHEAD));
            centre_right<FLT>(rq[C + AHEAD], lc);
        }
        quad_step<WIN_W, D, C>(acc, lq, rq);
        // keeps the scheduler from hoisting the LDS reads of later groups over this one:
        // with them in flight early the row needs more registers than there are (spills)
        __builtin_amdgcn_sched_barrier(0);
        row_step<WIN_W, D, ALIGNED, FLT, C + 1>(acc, lp, rp, l_head, rq, lc);
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
template <int WIN_W, int D, bool ALIGNED, bool FLT>
__device__ __forceinline__ void accumulate(const float *__restrict__ lp, const float *__restrict__ rp,
                                           int lpitch, int rpitch, int rows,
                                           float2v (&acc)[BAND_X / 2][D], const LaneCentre<FLT> &lc)
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
        row_step<WIN_W, D, ALIGNED, FLT>(acc, lp, rp, lA, rqA, lc);
        lp += lpitch;
        rp += rpitch;
        const float *lnext = a + 2 < rows ? lp + lpitch : lp;
        const float *rnext = a + 2 < rows ? rp + rpitch : rp;
        lA = load_quad<true>(lnext);
#pragma unroll
        for (int c = 0; c < HEAD; ++c)
            rqA[c] = load_quad<ALIGNED>(rnext + 4 * c);
        row_step<WIN_W, D, ALIGNED, FLT>(acc, lp, rp, lB, rqB, lc);
        lp = lnext;
        rp = rnext;
    }
    if (a < rows)
        row_step<WIN_W, D, ALIGNED, FLT>(acc, lp, rp, lA, rqA, lc);
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
template <int D, bool ALIGNED, bool FLT>
__device__ __forceinline__ void accumulate_any(const float *__restrict__ lp, const float *__restrict__ rp,
                                               int lpitch, int rpitch, int rows, int win_w,
                                               float2v (&acc)[BAND_X / 2][D], const LaneCentre<FLT> &lc)
{
    auto left = [&](const float *src) {
        Quad q = load_quad<true>(src);
        centre_left<FLT>(q, lc);
        return q;
    };
    auto right = [&](const float *src) {
        Quad q = load_quad<ALIGNED>(src);
        centre_right<FLT>(q, lc);
        return q;
    };
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
        Quad r0 = right(rp), r1 = right(rp + 4), r2 = right(rp + 8);
        quad_any<D, false>(acc, left(lp), r0, r1, r2, 0, win_w);
        r0 = r1;
        r1 = r2;
        int c = 1;
        for (int t = 0; t < rotations; ++t, c += 3) {
            r2 = right(rp + 4 * (c + 2));
            quad_any<D, true>(acc, left(lp + 4 * c), r0, r1, r2, 4 * c, win_w);
            r0 = right(rp + 4 * (c + 3));
            quad_any<D, true>(acc, left(lp + 4 * c + 4), r1, r2, r0, 4 * c + 4, win_w);
            r1 = right(rp + 4 * (c + 4));
            quad_any<D, true>(acc, left(lp + 4 * c + 8), r2, r0, r1, 4 * c + 8, win_w);
        }
        for (; c < groups; ++c) {
            r2 = right(rp + 4 * (c + 2));
            quad_any<D, false>(acc, left(lp + 4 * c), r0, r1, r2, 4 * c, win_w);
            r0 = r1;
            r1 = r2;
        }
        lp += lpitch;
        rp += rpitch;
    }
}

// WIN_W > 0: the width is a compile-time constant (fully unrolled row); WIN_W == 0:
// the width comes from the geometry at run time.
template <int WIN_W, int D, bool ALIGNED, bool FLT>
__device__ __forceinline__ void accumulate_rows(const float *__restrict__ lp, const float *__restrict__ rp,
                                                int lpitch, int rpitch, int rows, int win_w,
                                                float2v (&acc)[BAND_X / 2][D], const LaneCentre<FLT> &lc)
{
    if constexpr (WIN_W > 0)
        accumulate<WIN_W, D, ALIGNED, FLT>(lp, rp, lpitch, rpitch, rows, acc, lc);
    else
        accumulate_any<D, ALIGNED, FLT>(lp, rp, lpitch, rpitch, rows, win_w, acc, lc);
}

struct LaneState {
    double best[BAND_X];
    int arg[BAND_X];
};

// Turns the accumulators of offsets D0 .. D0 + DN - 1 (of the D of a pass) of each
// window into scores and feeds the selection.  k0: offset index (0 .. nd-1) of
// acc[.][0]; centre_pass: this pass evaluates the centre offset ahead of the
// ascending sweep (selection-rule mode only).
template <int D, int D0, int DN, bool FLT, typename TSum>
__device__ __forceinline__ void finish_range(const BandGeom &g, const float2v (&acc)[BAND_X / 2][D], int k0,
                                             int kend, bool centre_pass, int wy, int xbase,
                                             const TSum *__restrict__ wsum_l,
                                             const double *__restrict__ winv_l,
                                             const TSum *__restrict__ wsum_r,
                                             const double *__restrict__ winv_r, LaneState &st,
                                             double *__restrict__ scores, const LaneCentre<FLT> &lc,
                                             double (&margin)[BAND_X])
{
    const double n = (double)(g.win_w * g.win_h);
    const int64_t srow = (int64_t)wy * g.stride_y * g.nsx;
    // window x at offset d looks at the right-image box xbase + x + d + k0 - max_disp:
    // BAND_X + DN - 1 distinct boxes serve the lane's BAND_X * DN evaluations
    constexpr int NJ = BAND_X + DN - 1;
    TSum r_sum[NJ];
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
                double cross64 = (double)cross;
                if constexpr (FLT) {
                    // the lane's constants back in: sum (l - a)(r - b) = sum l r - b sum l - a sum r + n a b
                    const double a = (double)lc.a.x, b = (double)lc.b.x;
                    cross64 += a * (double)r_sum[x + dd] + b * s1l - n * a * b;
                }
                const double num = fma(n, cross64, -(s1l * (double)r_sum[x + dd]));
                s = num * invl * r_inv[x + dd];
            }
            if (scores && wvalid && kvalid)
                scores[((int64_t)wy * g.nwx + xw) * g.nd + k] = s;
            if (!kvalid)
                continue;
            if constexpr (FLT) {
                // how far this decision is from going the other way
                if (s == s) {
                    // Pixels are staged as float32 of (pixel - image mean): a window whose mean
                    // lies many of its own deviations away from the image mean has lost its
                    // texture to that rounding (relative error about 2^-23 mean / deviation per
                    // pixel).  Such an evaluation sends its window to the float64 second look.
                    const double lost = 1.2e-7 * (fabs(s1l) * invl + fabs((double)r_sum[x + dd]) * r_inv[x + dd]);
                    if (lost > 4e-6)
                        margin[x] = 0.0;
                    double m = INFINITY;
                    if (g.mode == SVAMD_BAND_ARGMAX)
                        m = fabs(s - st.best[x]);
                    else if (centre_pass)
                        m = fmin(fabs(s - g.thr), fabs(s - st.best[x]));
                    else if (k != g.max_disp)
                        m = fmin(fabs(s - g.thr), fabs(s - (st.best[x] + g.tol)));
                    margin[x] = fmin(margin[x], m);
                }
            }
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
template <int D, bool FLT, typename TSum>
__device__ __forceinline__ void finish_pass(const BandGeom &g, const float2v (&acc)[BAND_X / 2][D], int k0,
                                            int kend, bool centre_pass, int wy, int xbase,
                                            const TSum *__restrict__ wsum_l,
                                            const double *__restrict__ winv_l,
                                            const TSum *__restrict__ wsum_r,
                                            const double *__restrict__ winv_r, LaneState &st,
                                            double *__restrict__ scores, const LaneCentre<FLT> &lc,
                                            double (&margin)[BAND_X])
{
    if constexpr (D == 8) {
        finish_range<D, 0, 4, FLT>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                   scores, lc, margin);
        __builtin_amdgcn_sched_barrier(0);
        finish_range<D, 4, 4, FLT>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                   scores, lc, margin);
    } else {
        finish_range<D, 0, D, FLT>(g, acc, k0, kend, centre_pass, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                   scores, lc, margin);
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

template <int WIN_W, typename PIX>
__global__ void __launch_bounds__(BAND_THREADS)
band_scan_kernel(BandGeom g, const PIX *__restrict__ left, const PIX *__restrict__ right,
                 const typename std::conditional<std::is_same<PIX, uint8_t>::value, int32_t, double>::type *__restrict__ wsum_l,
                 const double *__restrict__ winv_l,
                 const typename std::conditional<std::is_same<PIX, uint8_t>::value, int32_t, double>::type *__restrict__ wsum_r,
                 const double *__restrict__ winv_r,
                 int32_t *__restrict__ out_disp, double *__restrict__ out_val, double *__restrict__ scores,
                 const Stats *__restrict__ image_stats, double *__restrict__ out_margin)
{
    constexpr bool FLT = !std::is_same<PIX, uint8_t>::value;
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
    double margin[BAND_X];
#pragma unroll
    for (int x = 0; x < BAND_X; ++x) {
        st.best[x] = g.mode == SVAMD_BAND_ARGMAX ? -INFINITY : 0.0;
        st.arg[x] = g.mode == SVAMD_BAND_ARGMAX ? SVAMD_BAND_NO_MATCH : 0;
        margin[x] = INFINITY;
    }

    // floating-point images: the image means the staged pixels are centred by, and the
    // constants this lane takes off what it reads (see the head of the file)
    double centre_l = 128.0, centre_r = 128.0;
    LaneCentre<FLT> lc;
    const double inv_n = 1.0 / (double)(g.win_w * g.win_h);
    const int64_t srow = active ? (int64_t)wy * g.stride_y * g.nsx : 0;
    auto right_constant = [&](int k) {           // about the mean of the boxes around offset index k
        if constexpr (FLT) {
            const int xr = min(max(xbase + BAND_X / 2 + k - g.max_disp, 0), g.nsx - 1);
            float b = active ? (float)((double)wsum_r[srow + xr] * inv_n) : 0.f;
            if (!(fabsf(b) < 3.0e38f))
                b = 0.f;
            lc.b = float2v{b, b};
        }
    };
    if constexpr (FLT) {
        centre_l = image_stats[0].mean;
        centre_r = image_stats[1].mean;
        float a = active ? (float)((double)wsum_l[srow + min(xbase + BAND_X / 2, g.nwx - 1)] * inv_n) : 0.f;
        if (!(fabsf(a) < 3.0e38f))
            a = 0.f;
        lc.a = float2v{a, a};
        lc.b = float2v{0.f, 0.f};
    }

    SV_STAMP(0);
    stage_tile(left, g.H, g.W, y0, g.tile_rows, x0, g.lpitch, Lt, centre_l);
    const float *lp = Lt + lrow * g.lpitch + BAND_X * lane;

    // The centre offset is judged first (selection-rule mode).  When the whole offset
    // range sits in LDS at once it is read from there, otherwise the right image is
    // staged at d = 0 for it.
    const bool whole_band = g.dchunk >= g.nd;
    if (g.mode == SVAMD_BAND_FOLD && !whole_band) {
        stage_tile(right, g.H, g.W, y0, g.tile_rows, x0, g.lpitch, Rt, centre_r);
        __syncthreads();
        if (active) {
            float2v acc[BAND_X / 2][1] = {};
            right_constant(g.max_disp);
            accumulate_rows<WIN_W, 1, true, FLT>(lp, Rt + lrow * g.lpitch + BAND_X * lane, g.lpitch, g.lpitch,
                                                 g.win_h, g.win_w, acc, lc);
            finish_pass<1, FLT>(g, acc, g.max_disp, g.nd, true, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                nullptr, lc, margin);
        }
    }

    for (int k0 = 0; k0 < g.nd; k0 += g.dchunk) {
        const int kend = min(k0 + g.dchunk, g.nd);
        __syncthreads();  // previous contents of Rt are no longer read
        stage_tile(right, g.H, g.W, y0, g.tile_rows, x0 - g.max_disp + k0, g.rpitch, Rt, centre_r);
        __syncthreads();
        if (!active)
            continue;
        const float *rp = Rt + lrow * g.rpitch + BAND_X * lane;
        if (g.mode == SVAMD_BAND_FOLD && whole_band) {
            float2v acc[BAND_X / 2][1] = {};
            right_constant(g.max_disp);
            if (g.max_disp & 3)
                accumulate_rows<WIN_W, 1, false, FLT>(lp, rp + g.max_disp, g.lpitch, g.rpitch, g.win_h, g.win_w,
                                                      acc, lc);
            else
                accumulate_rows<WIN_W, 1, true, FLT>(lp, rp + g.max_disp, g.lpitch, g.rpitch, g.win_h, g.win_w,
                                                     acc, lc);
            finish_pass<1, FLT>(g, acc, g.max_disp, g.nd, true, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                nullptr, lc, margin);
        }
        int k = k0;
        for (; k + BAND_D <= kend; k += BAND_D) {
            float2v acc[BAND_X / 2][BAND_D] = {};
            right_constant(k + BAND_D / 2);
            accumulate_rows<WIN_W, BAND_D, true, FLT>(lp, rp + (k - k0), g.lpitch, g.rpitch, g.win_h, g.win_w, acc,
                                                      lc);
            finish_pass<BAND_D, FLT>(g, acc, k, kend, false, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                     SV_SCORES_OUT, lc, margin);
            keep_pace(progress, wave, ++passes);
            SV_STAMP(1 + k / BAND_D);
        }
        for (; k < kend; ++k) {  // fewer than 8 offsets left in this chunk
            float2v acc[BAND_X / 2][1] = {};
            right_constant(k);
            accumulate_rows<WIN_W, 1, false, FLT>(lp, rp + (k - k0), g.lpitch, g.rpitch, g.win_h, g.win_w, acc, lc);
            finish_pass<1, FLT>(g, acc, k, kend, false, wy, xbase, wsum_l, winv_l, wsum_r, winv_r, st,
                                SV_SCORES_OUT, lc, margin);
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
                if constexpr (FLT)
                    out_margin[o] = margin[x];
            }
        }
    }
}

// ---- floating-point images: statistics, and the float64 second look -------------------
// Window sums and 1 / sqrt(A) from float64 tables of the centred pixels.  A is a
// difference of table entries whose rounding follows the image's size, not the window's:
// a window whose A is not far above that rounding (a flat patch of a large image) takes
// its sums from its own pixels, and counts as constant (NaN scores) only when all of them
// are equal.
__global__ void __launch_bounds__(256)
window_stats_f64_kernel(const double *__restrict__ img, const Stats *__restrict__ image_stats,
                        const double *__restrict__ s1, const double *__restrict__ s2, int H, int W, int win_w,
                        int win_h, double *__restrict__ wsum, double *__restrict__ winv)
{
    const int nsx = W - win_w + 1, nsy = H - win_h + 1;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nsx * nsy)
        return;
    const int y = (int)(idx / nsx), x = (int)(idx - (int64_t)y * nsx);
    const int64_t tp = (int64_t)W + 1;
    const int64_t a = (int64_t)y * tp + x, b = (int64_t)(y + win_h) * tp + x;
    double sum = s1[b + win_w] - s1[a + win_w] - s1[b] + s1[a];
    const double sq = s2[b + win_w] - s2[a + win_w] - s2[b] + s2[a];
    const double n = (double)win_w * (double)win_h;
    double A = n * sq - sum * sum;
    const double c1 = fmax(fmax(fabs(s1[a]), fabs(s1[a + win_w])), fmax(fabs(s1[b]), fabs(s1[b + win_w])));
    const double c2 = fmax(fmax(fabs(s2[a]), fabs(s2[a + win_w])), fmax(fabs(s2[b]), fabs(s2[b + win_w])));
    const double ulp = 2.220446049250313e-16 * (n * c2 + 2.0 * fabs(sum) * c1);
    bool flat = false;
    if (!(A > 1e7 * ulp)) {
        const double centre = image_stats->mean;
        double lo = INFINITY, hi = -INFINITY;
        sum = 0.0;
        for (int r = 0; r < win_h; ++r)
            for (int c = 0; c < win_w; ++c) {
                const double p = img[(int64_t)(y + r) * W + x + c];
                sum += p - centre;
                lo = fmin(lo, p);
                hi = fmax(hi, p);
            }
        const double mean = sum / n;
        double dev = 0.0;
        for (int r = 0; r < win_h; ++r)
            for (int c = 0; c < win_w; ++c) {
                const double d = img[(int64_t)(y + r) * W + x + c] - centre - mean;
                dev += d * d;
            }
        A = n * dev;                      // n * sum (p - mean)^2 = n * sq - sum^2
        flat = lo == hi;
    }
    wsum[idx] = sum;
    winv[idx] = (A > 0.0 && !flat) ? 1.0 / sqrt(A) : NAN;
}

__global__ void __launch_bounds__(256)
band_collect_close_kernel(const double *__restrict__ margin, int nwx, int row_begin, int row_end, double guard,
                          int32_t *__restrict__ list, int32_t *__restrict__ count)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t first = (int64_t)row_begin * nwx, total = (int64_t)(row_end - row_begin) * nwx;
    if (i >= total || !(margin[first + i] < guard))
        return;
    list[atomicAdd(count, 1)] = (int32_t)(first + i);
}

// The windows on the list once more, every offset in float64 straight from the images,
// then the same selection.  One wavefront per window: the centred left window sits in
// LDS; the right image's row segment under all offsets (nd + win_w - 1 columns) is staged
// in LDS a row ahead (coalesced loads held in registers while the current row is worked
// on; staging the left rows the same way instead of the whole window measured slower);
// a lane owns OPL consecutive offsets and slides a ring of OPL right values along the
// row (two LDS reads per OPL multiply-adds, no global load in the loop).  The window
// statistics are the tables' (float64 already); only the cross term differs from the
// first look.
constexpr int RECHECK_AHEAD = 6;      // staged values a lane holds in registers: 384 columns; beyond, loaded in place

template <int OPL>
__global__ void __launch_bounds__(WAVE)
band_recheck_kernel(BandGeom g, const double *__restrict__ left, const double *__restrict__ right,
                    const Stats *__restrict__ image_stats, const double *__restrict__ wsum_l,
                    const double *__restrict__ winv_l, const double *__restrict__ wsum_r,
                    const double *__restrict__ winv_r, const int32_t *__restrict__ list,
                    const int32_t *__restrict__ count, int32_t *__restrict__ out_disp,
                    double *__restrict__ out_val, double *__restrict__ scores)
{
    extern __shared__ __align__(16) double recheck_lds[];
    const int seg = g.nd + g.win_w - 1;               // right columns under the window at all offsets
    const int segp = (seg + OPL + 1) & ~1;            // + the ring's reach past the last offset
    double *Lw = recheck_lds;                         // win_h x win_w, centred
    double *Rs = Lw + ((g.win_w * g.win_h + 1) & ~1); // two staged row segments of segp values
    const int lane = threadIdx.x;
    const double centre_l = image_stats[0].mean, centre_r = image_stats[1].mean;
    const double n = (double)(g.win_w * g.win_h);
    const int todo = *count;
    for (int item = blockIdx.x; item < todo; item += gridDim.x) {
        const int64_t o = list[item];
        const int wy = (int)(o / g.nwx), wx = (int)(o - (int64_t)wy * g.nwx);
        const int y = wy * g.stride_y;
        const int64_t srow = (int64_t)y * g.nsx;
        const double s1l = wsum_l[srow + wx], invl = winv_l[srow + wx];
        const int x0 = wx - g.max_disp;                       // image column of the segment's first value
        // columns outside the image are read at the border: they only reach offsets whose
        // box leaves the image, and those are discarded below
        auto fetch = [&](int r, double (&held)[RECHECK_AHEAD]) {
            const double *rrow = right + (int64_t)(y + r) * g.W;
#pragma unroll
            for (int u = 0; u < RECHECK_AHEAD; ++u) {
                const int i = lane + u * WAVE;
                held[u] = i < segp ? rrow[min(max(x0 + i, 0), g.W - 1)] : 0.0;
            }
        };
        auto stage = [&](int r, int buf, const double (&held)[RECHECK_AHEAD]) {
            double *dst = Rs + buf * segp;
#pragma unroll
            for (int u = 0; u < RECHECK_AHEAD; ++u) {
                const int i = lane + u * WAVE;
                if (i < segp)
                    dst[i] = held[u] - centre_r;
            }
            const double *rrow = right + (int64_t)(y + r) * g.W;
            for (int i = lane + RECHECK_AHEAD * WAVE; i < segp; i += WAVE)
                dst[i] = rrow[min(max(x0 + i, 0), g.W - 1)] - centre_r;
        };
        double held[RECHECK_AHEAD];
        fetch(0, held);
        __syncthreads();                       // the previous window's LDS contents have been read
        // (row = i / win_w by a multiply: exact for i * win_w < 2^32, and i < 4096 here)
        const uint32_t inv_w = (uint32_t)((((uint64_t)1 << 32) + g.win_w - 1) / g.win_w);
        const double *lwin = left + (int64_t)y * g.W + wx;
#pragma unroll 4
        for (int i = lane; i < g.win_w * g.win_h; i += WAVE) {
            const int r = (int)__umulhi((uint32_t)i, inv_w), c = i - r * g.win_w;
            Lw[i] = lwin[r * g.W + c] - centre_l;
        }
        stage(0, 0, held);
        __syncthreads();
        const int k_first = lane * OPL;                       // this lane's first offset index
        const int x_first = x0 + k_first;                     // ... and its box's first column
        double acc[OPL];
#pragma unroll
        for (int j = 0; j < OPL; ++j)
            acc[j] = 0.0;
        for (int r = 0; r < g.win_h; ++r) {
            if (r + 1 < g.win_h)
                fetch(r + 1, held);
            if (k_first < g.nd) {
                const double *rr = Rs + (r & 1) * segp + k_first;
                const double *lrow = Lw + r * g.win_w;
                double ring[OPL];                             // R[c + j] lives in slot (c + j) % OPL
#pragma unroll
                for (int j = 0; j + 1 < OPL; ++j)
                    ring[j] = rr[j];
                // whole chunks of OPL columns, the next chunk's 2 OPL values read from LDS while
                // this chunk's OPL x OPL multiply-adds run (one wait per chunk)
                const int whole = g.win_w / OPL * OPL;
                double lv[OPL], nv[OPL], lv2[OPL], nv2[OPL];
                auto read_chunk = [&](int c0, double (&l)[OPL], double (&v)[OPL]) {
#pragma unroll
                    for (int u = 0; u < OPL; ++u) {
                        l[u] = lrow[c0 + u];
                        v[u] = rr[c0 + u + OPL - 1];
                    }
                };
                auto work_chunk = [&](const double (&l)[OPL], const double (&v)[OPL]) {
#pragma unroll
                    for (int u = 0; u < OPL; ++u) {
                        ring[(u + OPL - 1) % OPL] = v[u];
#pragma unroll
                        for (int j = 0; j < OPL; ++j)
                            acc[j] = fma(l[u], ring[(u + j) % OPL], acc[j]);
                    }
                };
                int c0 = 0;
                if (whole > 0)
                    read_chunk(0, lv, nv);
                for (; c0 + 2 * OPL <= whole; c0 += 2 * OPL) {
                    read_chunk(c0 + OPL, lv2, nv2);
                    __builtin_amdgcn_sched_barrier(0);
                    work_chunk(lv, nv);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c0 + 2 * OPL < whole)
                        read_chunk(c0 + 2 * OPL, lv, nv);
                    __builtin_amdgcn_sched_barrier(0);
                    work_chunk(lv2, nv2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c0 < whole) {
                    work_chunk(lv, nv);
                    c0 += OPL;
                }
                // the columns left over (win_w is not a multiple of OPL)
#pragma unroll
                for (int u = 0; u < OPL; ++u) {
                    const int c = c0 + u;
                    if (c < g.win_w) {                        // wave-uniform
                        ring[(u + OPL - 1) % OPL] = rr[c + OPL - 1];
                        const double l = lrow[c];
#pragma unroll
                        for (int j = 0; j < OPL; ++j)
                            acc[j] = fma(l, ring[(u + j) % OPL], acc[j]);
                    }
                }
            }
            if (r + 1 < g.win_h)
                stage(r + 1, (r + 1) & 1, held);
            __syncthreads();
        }
        // scores stay in the lanes' registers; invalid ones (box outside the image) are NaN
        double s[OPL];
#pragma unroll
        for (int j = 0; j < OPL; ++j) {
            const int k = k_first + j;
            const int xr = x_first + j;
            s[j] = NAN;
            if (k < g.nd && xr >= 0 && xr < g.nsx)
                s[j] = fma(n, acc[j], -(s1l * wsum_r[srow + xr])) * invl * winv_r[srow + xr];
            if (scores && k < g.nd)
                scores[o * g.nd + k] = s[j];
        }
        double best;
        int arg;
        if (g.mode == SVAMD_BAND_ARGMAX) {
            // first maximum in offset order: within the lane, then across lanes (ties to the lower offset)
            best = -INFINITY;
            arg = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < OPL; ++j)
                if (s[j] > best) {
                    best = s[j];
                    arg = k_first + j;
                }
#pragma unroll
            for (int m = 1; m < WAVE; m <<= 1) {
                const double ob = __shfl_xor(best, m);
                const int oa = __shfl_xor(arg, m);
                if (ob > best || (ob == best && oa < arg)) {
                    best = ob;
                    arg = oa;
                }
            }
            if (arg == 0x7fffffff) {
                best = NAN;
                arg = SVAMD_BAND_NO_MATCH;
            } else {
                arg -= g.max_disp;
            }
        } else {
            // image_scan's rule walks the offsets in order and moves on each candidate that beats
            // the threshold and the running best by more than the tolerance.  Every lane replays
            // the walk over the lanes' blocks; a block whose largest eligible score cannot beat
            // the running best is passed over whole.
            double centre_score = NAN, block_max = -INFINITY;
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int k = k_first + j;
                if (k == g.max_disp)
                    centre_score = s[j];
                else if (s[j] > g.thr && s[j] > block_max)
                    block_max = s[j];
            }
            best = 0.0;
            arg = 0;
            const double c = __shfl(centre_score, g.max_disp / OPL);
            if (c > g.thr && c >= best)
                best = c;
            const int blocks = (g.nd + OPL - 1) / OPL;
            for (int b = 0; b < blocks; ++b) {
                if (!(__shfl(block_max, b) > best + g.tol))
                    continue;                                  // wave-uniform
#pragma unroll
                for (int j = 0; j < OPL; ++j) {
                    const double v = __shfl(s[j], b);
                    const int k = b * OPL + j;
                    if (k != g.max_disp && v > g.thr && v > best + g.tol) {
                        best = v;
                        arg = k - g.max_disp;
                    }
                }
            }
        }
        if (lane == 0) {
            out_disp[o] = arg;
            out_val[o] = best;
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

template <int WIN_W, typename PIX, typename TSum>
static int band_launch(const BandGeom &g, size_t lds, dim3 grid, hipStream_t st, const PIX *left,
                       const PIX *right, const TSum *wl, const double *il, const TSum *wr,
                       const double *ir, int32_t *out_disp, double *out_val, double *scores,
                       const Stats *image_stats = nullptr, double *out_margin = nullptr)
{
    // per function and per device: set before every launch (a process may drive several devices)
    SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&band_scan_kernel<WIN_W, PIX>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BAND_LDS_LIMIT));
    hipLaunchKernelGGL((band_scan_kernel<WIN_W, PIX>), grid, dim3(BAND_THREADS), lds, st, g, left, right, wl, il,
                       wr, ir, out_disp, out_val, scores, image_stats, out_margin);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// Tile geometry.  Row pitches carry 4 floats of slack because the last 4-wide LDS
// read of a lane may extend past the columns it uses.  As many offsets as fit in LDS
// are staged per pass over the right image.  Returns the dynamic LDS size, 0 when
// the window does not fit.
static size_t band_tile_geometry(BandGeom &g, bool tuned)
{
    const int slack = tuned ? 4 : 12;   // the run-time-width row walk reads up to three quads ahead
    g.lpitch = (int)align_up((size_t)(BAND_TX + g.win_w - 1), 4) + slack;
    size_t lds = 0;
    for (g.tile_wrows = BAND_WAVES; g.tile_wrows >= 1; --g.tile_wrows) {
        g.tile_rows = (g.tile_wrows - 1) * g.stride_y + g.win_h;
        g.dchunk = (int)align_up((size_t)g.nd, BAND_D);
        for (;;) {
            const int dcur = g.dchunk < g.nd ? g.dchunk : g.nd;
            g.rpitch = (int)align_up((size_t)(BAND_TX + g.win_w - 1 + dcur - 1), 4) + slack;
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
    return lds <= (size_t)BAND_LDS_LIMIT ? lds : 0;
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

extern "C" int svamd_band_box_sums_supported(int win_w, int win_h, int stride_y, int max_disp)
{
    if (win_w <= 0 || win_h <= 0 || stride_y <= 0 || max_disp < 0 || (int64_t)win_w * win_h > 4096)
        return 0;
    return band_sums_fits(win_w, win_h, stride_y, max_disp) ? 1 : 0;
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
    SV_REQUIRE((int64_t)win_w * win_h <= 4096, SVAMD_ERR_LIMIT,
               "band_prepare_u8: window area %d exceeds 4096", win_w * win_h);
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
    // (the box-sum kernel adds integers: its bound is the tables', which a tile's LDS size keeps far away)
    if (flags & SVAMD_BAND_BOX_SUMS)
        SV_REQUIRE((int64_t)win_w * win_h <= 4096, SVAMD_ERR_LIMIT, "band_scan_u8: window area %d exceeds 4096",
                   win_w * win_h);
    else
        SV_REQUIRE((int64_t)win_w * win_h <= 1024, SVAMD_ERR_LIMIT,
                   "band_scan_u8: window area %d exceeds 1024 (float32 accumulation would not be exact; "
                   "SVAMD_BAND_BOX_SUMS takes up to 4096)",
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

    // every width up to 32 has a kernel with a fully unrolled row (the widths the reference's
    // scan configurations ship - 5, 10, 12, 20, 24, 25 - run 20-40 % faster in it than in the
    // kernel that takes the width at run time, which serves the wider windows)
    const bool tuned = win_w >= 3 && win_w <= 32;
    const size_t lds = band_tile_geometry(g, tuned);
    SV_REQUIRE(lds > 0 || (flags & SVAMD_BAND_BOX_SUMS), SVAMD_ERR_LIMIT,
               "band_scan_u8: window %dx%d with row stride %d does not fit the LDS tile", win_w, win_h,
               stride_y);                          // (the box-sum kernel checks its own tile)

    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & SVAMD_BAND_PREPARED)) {
        int rc = svamd_band_prepare_u8(left, right, H, W, win_w, win_h, workspace, workspace_bytes, stream);
        if (rc != SVAMD_OK)
            return rc;
    }
    BandTables t = band_tables(workspace, H, W, win_w, win_h);
    const int32_t *wl = t.wl, *wr = t.wr;
    const double *il = t.il, *ir = t.ir;

    if (flags & SVAMD_BAND_BOX_SUMS)
        return band_sums_launch(left, right, H, W, win_w, win_h, stride_y, max_disp, mode, threshold, centre_tol,
                                row_begin, row_end, wl, il, wr, ir, out_disp, out_val, scores, st);

    dim3 grid((g.nwx + BAND_TX - 1) / BAND_TX, (row_end - row_begin + g.tile_wrows - 1) / g.tile_wrows);
#define SV_BAND_CASE(WW)                                                                          \
    case WW:                                                                                      \
        return band_launch<WW>(g, lds, grid, st, left, right, wl, il, wr, ir, out_disp, out_val, scores)
    switch (win_w) {
        SV_BAND_CASE(3);
        SV_BAND_CASE(4);
        SV_BAND_CASE(5);
        SV_BAND_CASE(6);
        SV_BAND_CASE(7);
        SV_BAND_CASE(8);
        SV_BAND_CASE(9);
        SV_BAND_CASE(10);
        SV_BAND_CASE(11);
        SV_BAND_CASE(12);
        SV_BAND_CASE(13);
        SV_BAND_CASE(14);
        SV_BAND_CASE(15);
        SV_BAND_CASE(16);
        SV_BAND_CASE(17);
        SV_BAND_CASE(18);
        SV_BAND_CASE(19);
        SV_BAND_CASE(20);
        SV_BAND_CASE(21);
        SV_BAND_CASE(22);
        SV_BAND_CASE(23);
        SV_BAND_CASE(24);
        SV_BAND_CASE(25);
        SV_BAND_CASE(26);
        SV_BAND_CASE(27);
        SV_BAND_CASE(28);
        SV_BAND_CASE(29);
        SV_BAND_CASE(30);
        SV_BAND_CASE(31);
        SV_BAND_CASE(32);
    default:   // any other width: the kernel that takes it at run time
        return band_launch<0>(g, lds, grid, st, left, right, wl, il, wr, ir, out_disp, out_val, scores);
    }
#undef SV_BAND_CASE
}

// ---- the same scan for float64 images ------------------------------------------------
namespace {
struct BandTablesF {
    Stats *stats;          // [0] left image, [1] right image
    double *sscr;
    double *s1l, *s2l, *s1r, *s2r, *wl, *wr, *il, *ir, *margin;
    int32_t *list, *count;
    size_t bytes;
};
BandTablesF band_tables_f(void *workspace, int H, int W, int win_w, int win_h)
{
    BandWs b = band_ws(H, W, win_w, win_h);
    Carver ws(workspace);
    BandTablesF t;
    t.stats = ws.take<Stats>(2);
    t.sscr = ws.take<double>(stats_scratch_doubles());
    t.s1l = ws.take<double>(b.n_sat);
    t.s2l = ws.take<double>(b.n_sat);
    t.s1r = ws.take<double>(b.n_sat);
    t.s2r = ws.take<double>(b.n_sat);
    t.wl = ws.take<double>(b.n_stat);
    t.wr = ws.take<double>(b.n_stat);
    t.il = ws.take<double>(b.n_stat);
    t.ir = ws.take<double>(b.n_stat);
    t.margin = ws.take<double>(b.n_stat);
    t.list = ws.take<int32_t>(b.n_stat);
    t.count = ws.take<int32_t>(64);
    t.bytes = ws.used;
    return t;
}
}  // namespace

extern "C" size_t svamd_band_scan_f64_workspace_bytes(int H, int W, int win_w, int win_h)
{
    if (H <= 0 || W <= 0 || win_w <= 0 || win_h <= 0 || win_w > W || win_h > H)
        return 0;
    return band_tables_f(nullptr, H, W, win_w, win_h).bytes;
}

extern "C" size_t svamd_band_scan_f64_recheck_count_offset(int H, int W, int win_w, int win_h)
{
    if (H <= 0 || W <= 0 || win_w <= 0 || win_h <= 0 || win_w > W || win_h > H)
        return 0;
    BandTablesF t = band_tables_f(nullptr, H, W, win_w, win_h);
    return (size_t)(reinterpret_cast<char *>(t.count) - static_cast<char *>(nullptr));
}

extern "C" int svamd_band_prepare_f64(const double *left, const double *right, int H, int W, int win_w,
                                      int win_h, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(left && right, SVAMD_ERR_INVALID, "band_prepare_f64: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && win_w > 0 && win_h > 0, SVAMD_ERR_INVALID, "band_prepare_f64: non-positive size");
    SV_REQUIRE(win_w <= W && win_h <= H, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE((int64_t)(H + 1) * (W + 1) < (1LL << 31), SVAMD_ERR_LIMIT, "band_prepare_f64: image too large");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_band_scan_f64_workspace_bytes(H, W, win_w, win_h),
               SVAMD_ERR_WORKSPACE, "band_prepare_f64: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BandWs b = band_ws(H, W, win_w, win_h);
    BandTablesF t = band_tables_f(workspace, H, W, win_w, win_h);
    int rc;
    if ((rc = stats_launch(left, SVAMD_F64, H, W, W, false, t.stats, t.sscr, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(right, SVAMD_F64, H, W, W, false, t.stats + 1, t.sscr, st)) != SVAMD_OK) return rc;
    // (the window statistics are written after the tables: until then their buffer is the tables' scratch)
    const size_t spare = b.n_stat * sizeof(double);
    if ((rc = sat2d_launch<double, double>(left, H, W, W, 0.0, t.s1l, t.s2l, st, t.il, spare, &t.stats[0].mean)) !=
        SVAMD_OK)
        return rc;
    if ((rc = sat2d_launch<double, double>(right, H, W, W, 0.0, t.s1r, t.s2r, st, t.ir, spare, &t.stats[1].mean)) !=
        SVAMD_OK)
        return rc;
    const unsigned sg = (unsigned)((b.n_stat + 255) / 256);
    hipLaunchKernelGGL(window_stats_f64_kernel, dim3(sg), dim3(256), 0, st, left, t.stats, t.s1l, t.s2l, H, W, win_w,
                       win_h, t.wl, t.il);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_stats_f64_kernel, dim3(sg), dim3(256), 0, st, right, t.stats + 1, t.s1r, t.s2r, H, W,
                       win_w, win_h, t.wr, t.ir);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" int svamd_band_scan_f64(const double *left, const double *right, int H, int W, int win_w, int win_h,
                                   int stride_x, int stride_y, int max_disp, int mode, double threshold,
                                   double centre_tol, double guard_band, int row_begin, int row_end, int flags,
                                   int32_t *out_disp, double *out_val, double *scores, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(left && right && out_disp && out_val, SVAMD_ERR_INVALID, "band_scan_f64: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && win_w > 0 && win_h > 0 && max_disp >= 0 && stride_y > 0, SVAMD_ERR_INVALID,
               "band_scan_f64: non-positive size");
    SV_REQUIRE(win_w <= W && win_h <= H, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE(mode == SVAMD_BAND_ARGMAX || mode == SVAMD_BAND_FOLD, SVAMD_ERR_INVALID,
               "band_scan_f64: unknown mode %d", mode);
    SV_REQUIRE(stride_x == 1, SVAMD_ERR_LIMIT, "band_scan_f64: only stride_x == 1 is supported (got %d)",
               stride_x);
    SV_REQUIRE((int64_t)win_w * win_h <= 4096, SVAMD_ERR_LIMIT, "band_scan_f64: window area %d exceeds 4096",
               win_w * win_h);
    SV_REQUIRE((int64_t)(H + 1) * (W + 1) < (1LL << 31), SVAMD_ERR_LIMIT, "band_scan_f64: image too large");
    SV_REQUIRE(guard_band >= 0.0, SVAMD_ERR_INVALID, "band_scan_f64: negative guard band");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_band_scan_f64_workspace_bytes(H, W, win_w, win_h),
               SVAMD_ERR_WORKSPACE, "band_scan_f64: workspace missing, misaligned or too small");
    BandGeom g;
    g.H = H; g.W = W; g.win_w = win_w; g.win_h = win_h; g.stride_y = stride_y; g.max_disp = max_disp;
    g.nd = 2 * max_disp + 1;
    g.nwx = W - win_w + 1;
    g.nwy = (H - win_h) / stride_y + 1;
    g.nsx = W - win_w + 1;
    g.mode = mode; g.thr = threshold; g.tol = centre_tol;
    SV_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= g.nwy, SVAMD_ERR_INVALID,
               "band_scan_f64: row range [%d, %d) outside 0..%d", row_begin, row_end, g.nwy);
    g.row_begin = row_begin; g.row_end = row_end;
    if (row_begin == row_end)
        return SVAMD_OK;
    // (fewer fixed widths than for 8-bit images: the shipped configurations' and the odd sizes up to 31)
    static const int FIXED_WIDTHS[] = {5, 7, 9, 10, 11, 12, 15, 16, 20, 21, 24, 25, 31};
    bool tuned = false;
    for (int w : FIXED_WIDTHS)
        tuned |= win_w == w;
    const size_t lds = band_tile_geometry(g, tuned);
    SV_REQUIRE(lds > 0, SVAMD_ERR_LIMIT, "band_scan_f64: window %dx%d with row stride %d does not fit the LDS tile",
               win_w, win_h, stride_y);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    if (!(flags & SVAMD_BAND_PREPARED) &&
        (rc = svamd_band_prepare_f64(left, right, H, W, win_w, win_h, workspace, workspace_bytes, stream)) != SVAMD_OK)
        return rc;
    BandTablesF t = band_tables_f(workspace, H, W, win_w, win_h);
    // the margins are indexed by window (row of the window grid), the statistics by image position
    dim3 grid((g.nwx + BAND_TX - 1) / BAND_TX, (row_end - row_begin + g.tile_wrows - 1) / g.tile_wrows);
#define SV_BAND_CASE(WW)                                                                                      \
    case WW:                                                                                                  \
        rc = band_launch<WW>(g, lds, grid, st, left, right, t.wl, t.il, t.wr, t.ir, out_disp, out_val, scores, \
                             t.stats, t.margin);                                                              \
        break
    switch (win_w) {
        SV_BAND_CASE(5);
        SV_BAND_CASE(7);
        SV_BAND_CASE(9);
        SV_BAND_CASE(10);
        SV_BAND_CASE(11);
        SV_BAND_CASE(12);
        SV_BAND_CASE(15);
        SV_BAND_CASE(16);
        SV_BAND_CASE(20);
        SV_BAND_CASE(21);
        SV_BAND_CASE(24);
        SV_BAND_CASE(25);
        SV_BAND_CASE(31);
#undef SV_BAND_CASE
    default:
        rc = band_launch<0>(g, lds, grid, st, left, right, t.wl, t.il, t.wr, t.ir, out_disp, out_val, scores, t.stats,
                            t.margin);
    }
    if (rc != SVAMD_OK || guard_band == 0.0)
        return rc;
    SV_HIP(hipMemsetAsync(t.count, 0, 64 * sizeof(int32_t), st));
    const int64_t total = (int64_t)(row_end - row_begin) * g.nwx;
    hipLaunchKernelGGL(band_collect_close_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, t.margin,
                       g.nwx, row_begin, row_end, guard_band, t.list, t.count);
    SV_LAUNCH_CHECK();
    const unsigned rgrid = (unsigned)(total < 32768 ? total : 32768);
    const int opl = (g.nd + WAVE - 1) / WAVE;                 // offsets per lane of the second look
    // the window and two staged right-row segments (as band_recheck_kernel lays them out)
    const size_t rsegp = (size_t)((g.nd + win_w - 1 + opl + 1) & ~1);
    const size_t rlds = ((size_t)((win_w * win_h + 1) & ~1) + 2 * rsegp) * sizeof(double);
    SV_REQUIRE(opl <= 8 && rlds <= 60 * 1024, SVAMD_ERR_LIMIT,
               "band_scan_f64: %d offsets are more than the float64 second look handles (512)", g.nd);
#define SV_RECHECK_CASE(N)                                                                                     \
    case N:                                                                                                    \
        hipLaunchKernelGGL(band_recheck_kernel<N>, dim3(rgrid), dim3(WAVE), rlds, st, g, left, right, t.stats, \
                           t.wl, t.il, t.wr, t.ir, t.list, t.count, out_disp, out_val, scores);                \
        break
    switch (opl) {
        SV_RECHECK_CASE(1);
        SV_RECHECK_CASE(2);
        SV_RECHECK_CASE(3);
        SV_RECHECK_CASE(4);
        SV_RECHECK_CASE(5);
        SV_RECHECK_CASE(6);
        SV_RECHECK_CASE(7);
    default:
        SV_RECHECK_CASE(8);
    }
#undef SV_RECHECK_CASE
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
