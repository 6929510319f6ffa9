// The dense band scan of 8-bit images by box sums instead of one multiply-add per window
// pixel: for every offset the tile's product image L[y][x] * R[y][x + d] is turned into a
// summed-area table in LDS, a window's cross term is four table reads, and the score is
// the same float64 expression of exact integers as in band_scan.hip - so scores, selected
// offsets and correlations are identical bit for bit, at a few dozen instructions per
// (window, offset) instead of 2 * win_w * win_h flops.
//
// Same job as band_scan_kernel (reference: the spatial NCC of correlation.py:104-116 at
// full overlap, then utils.max_pos (utils.py:17-26) or image_scan's rule
// (sv_image_comparison.py:249-255) per window); svamd_band_scan_u8 takes this kernel when
// asked to (SVAMD_BAND_BOX_SUMS).
//
// Workgroup = 16 wavefronts = a tile of 32 (or 16) window rows x 256 window columns; the
// left tile and the right band under all offsets sit in LDS as bytes.  Per offset, three
// phases separated by barriers:
//   A  a wavefront owns a block of consecutive image rows: per row each lane forms the
//      running sums of 8 consecutive pixel products (dot products of masked bytes), the
//      lanes' totals are prefixed across the wavefront (DPP), and the row is added to the
//      block's rows above it - the table a block writes is its own 2-D prefix sum;
//   B  sixteen lanes per column prefix the blocks' last rows: each block's carry, i.e.
//      what the blocks above it add to its columns;
//   C  a thread per 8 (or 4) windows: four table corners and the four carries of their
//      blocks, the right window's statistics from global memory, the score in float64,
//      the selection kept in registers for the whole tile.
// Integer arithmetic wraps around modulo 2^32; window sums are differences of table
// entries and fit 32 bits, so they are exact.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svamd {

namespace {

constexpr int SUMS_WAVES = 16;
constexpr int SUMS_THREADS = SUMS_WAVES * WAVE;
constexpr int SUMS_WC = 256;                 // window columns per tile
constexpr int SUMS_COLS_PER_THREAD = SUMS_WC / WAVE;
constexpr int SUMS_CPL = 8;                  // product columns per lane in phase A
constexpr int SUMS_LDS_LIMIT = 160 * 1024;

struct SumsGeom {
    int H, W, win_w, win_h, stride_y, max_disp, nd;
    int nwx, nsx;          // window columns; columns of the statistics tables (the same number)
    int mode;
    double thr, tol;
    int row_begin, row_end;
    int tr;                // image rows of a tile
    int rows_a;            // ... per wavefront in phase A
    int lanes_a;           // lanes of phase A that hold product columns
    int lpitch, rpitch;    // bytes per staged row of the left / right tile
    int spitch;            // table entries per row: 3 unused, the zero column, then the product columns
};

// Inclusive prefix sums over the 64 lanes (wrap-around arithmetic: differences of table
// entries stay exact).  Within rows of 16 lanes by shifts of 1, 2, 4, 8; then lane 15 of
// rows 0 and 2 into rows 1 and 3, and lane 31 into rows 2 and 3.
__device__ __forceinline__ uint32_t wave_prefix_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Diagnostic build only (-DSVAMD_SUMS_STAMPS, see profiles/stamp_sums.py): `scores` is taken
// as eight 64-bit counters to which lane 0 of every wavefront adds the clocks it spent in
// each phase (no scores are written).
#ifdef SVAMD_SUMS_STAMPS
#define SUMS_STAMP(slot)                                        \
    do {                                                        \
        const long long now_ = clock64();                       \
        stamp_acc_[slot] += now_ - stamp_last_;                 \
        stamp_last_ = now_;                                     \
    } while (0)
#else
#define SUMS_STAMP(slot) do { } while (0)
#endif

// RPT: window rows per thread (tile = 16 * RPT window rows).
template <int RPT>
__global__ void __launch_bounds__(SUMS_THREADS)
band_sums_kernel(SumsGeom g, const uint8_t *__restrict__ left, const uint8_t *__restrict__ right,
                 const int32_t *__restrict__ wsum_l, const double *__restrict__ winv_l,
                 const int32_t *__restrict__ wsum_r, const double *__restrict__ winv_r,
                 int32_t *__restrict__ out_disp, double *__restrict__ out_val, double *__restrict__ scores)
{
    extern __shared__ __align__(16) uint32_t sums_lds[];
    uint32_t *sat = sums_lds;                                         // (tr + 1) x spitch
    uint32_t *carry = sat + (g.tr + 1) * g.spitch;                    // one row per block of rows_a image rows
    uint8_t *Lb = reinterpret_cast<uint8_t *>(carry + SUMS_WAVES * g.spitch);  // tr x lpitch
    uint8_t *Rb = Lb + g.tr * g.lpitch;                               // tr x rpitch

    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int x0 = blockIdx.x * SUMS_WC;                              // first window column of the tile
    const int wy0 = g.row_begin + blockIdx.y * (SUMS_WAVES * RPT);    // first window row
    const int y0 = wy0 * g.stride_y;                                  // first image row

    // ---- staging: the left tile, the right band under all offsets, the table's zero row and column
    for (int i = threadIdx.x; i < g.tr * g.lpitch; i += SUMS_THREADS) {
        const int r = i / g.lpitch, c = i - r * g.lpitch;
        const int y = y0 + r, x = x0 + c;
        Lb[i] = (y < g.H && x < g.W) ? left[(int64_t)y * g.W + x] : (uint8_t)0;
    }
    for (int i = threadIdx.x; i < g.tr * g.rpitch; i += SUMS_THREADS) {
        const int r = i / g.rpitch, c = i - r * g.rpitch;
        const int y = y0 + r, x = x0 - g.max_disp + c;
        Rb[i] = (y < g.H && x >= 0 && x < g.W) ? right[(int64_t)y * g.W + x] : (uint8_t)0;
    }
    for (int i = threadIdx.x; i < g.spitch; i += SUMS_THREADS)
        sat[i] = 0;
    for (int r = threadIdx.x; r <= g.tr; r += SUMS_THREADS)
        sat[r * g.spitch + 3] = 0;
    for (int i = threadIdx.x; i < SUMS_WAVES * g.spitch; i += SUMS_THREADS)
        carry[i] = 0;                                                 // block 0 and the zero column take nothing
    const uint32_t inv_rows = (uint32_t)((((uint64_t)1 << 32) + g.rows_a - 1) / g.rows_a);   // r / rows_a by a multiply (r is small)

    // ---- the windows of this thread: rows wave, wave + 16; columns lane, lane + 64, ...
    constexpr int NW = RPT * SUMS_COLS_PER_THREAD;
    double best[NW], invl[NW];
    int arg[NW], s1l[NW], r_first[RPT];
    const double n = (double)(g.win_w * g.win_h);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int wy = wy0 + wave + SUMS_WAVES * rr;
        const int64_t srow = (int64_t)min(wy, g.row_end - 1) * g.stride_y * g.nsx;
        r_first[rr] = (int)srow;                                      // (the tables hold fewer than 2^31 entries)
#pragma unroll
        for (int j = 0; j < SUMS_COLS_PER_THREAD; ++j) {
            const int w = rr * SUMS_COLS_PER_THREAD + j;
            const int xw = min(x0 + lane + WAVE * j, g.nwx - 1);
            best[w] = g.mode == SVAMD_BAND_ARGMAX ? -INFINITY : 0.0;
            arg[w] = g.mode == SVAMD_BAND_ARGMAX ? SVAMD_BAND_NO_MATCH : 0;
            s1l[w] = wsum_l[srow + xw];
            invl[w] = winv_l[srow + xw];
        }
    }
    __syncthreads();

#ifdef SVAMD_SUMS_STAMPS
    long long stamp_acc_[8] = {}, stamp_last_ = clock64();
    double *scores_out = nullptr;
#else
    double *scores_out = scores;
#endif
    // selection-rule mode judges the centre offset first, then all others in ascending order
    const int steps = g.mode == SVAMD_BAND_FOLD ? g.nd + 1 : g.nd;
    for (int it = 0; it < steps; ++it) {
        int k = it;
        bool centre_rule = false;
        if (g.mode == SVAMD_BAND_FOLD) {
            centre_rule = it == 0;
            k = centre_rule ? g.max_disp : it - 1;
            if (!centre_rule && k == g.max_disp)
                continue;                                             // uniform over the workgroup
        }

        SUMS_STAMP(0);      // loop head, barrier wait of the previous step
        const double thr_k = g.mode == SVAMD_BAND_FOLD ? g.thr : -INFINITY;
        const double tol_k = g.mode == SVAMD_BAND_FOLD && !centre_rule ? g.tol : 0.0;
        // ---- phase A: a wavefront owns a block of consecutive image rows and leaves the
        // block's own two-dimensional prefix sums of the products of offset index k (right
        // column = left column + k): a row's running sums plus the block's rows above it
        const int shift = k & 3;
        {
            uint32_t above[SUMS_CPL];
#pragma unroll
            for (int i = 0; i < SUMS_CPL; ++i)
                above[i] = 0;
            const int y_end = min((wave + 1) * g.rows_a, g.tr);
            for (int y = wave * g.rows_a; y < y_end; ++y) {
                // (lanes without columns read the last lane's bytes and multiply by zero: no branch)
                uint32_t p[SUMS_CPL];
                const int la = min(lane, g.lanes_a - 1);
                uint2 lw = *reinterpret_cast<const uint2 *>(Lb + y * g.lpitch + SUMS_CPL * la);
                lw.x = lane < g.lanes_a ? lw.x : 0u;
                lw.y = lane < g.lanes_a ? lw.y : 0u;
                const uint32_t *rw =
                    reinterpret_cast<const uint32_t *>(Rb + y * g.rpitch + SUMS_CPL * la + (k & ~3));
                const uint32_t r0 = rw[0], r1 = rw[1], r2 = rw[2];
                const uint32_t ra = __builtin_amdgcn_alignbyte(r1, r0, shift);
                const uint32_t rb = __builtin_amdgcn_alignbyte(r2, r1, shift);
                // running sums of the four byte products of a word: dot products with the
                // left bytes masked off from the top
                p[0] = __builtin_amdgcn_udot4(lw.x & 0xffu, ra, 0u, false);
                p[1] = __builtin_amdgcn_udot4(lw.x & 0xffffu, ra, 0u, false);
                p[2] = __builtin_amdgcn_udot4(lw.x & 0xffffffu, ra, 0u, false);
                p[3] = __builtin_amdgcn_udot4(lw.x, ra, 0u, false);
                p[4] = __builtin_amdgcn_udot4(lw.y & 0xffu, rb, p[3], false);
                p[5] = __builtin_amdgcn_udot4(lw.y & 0xffffu, rb, p[3], false);
                p[6] = __builtin_amdgcn_udot4(lw.y & 0xffffffu, rb, p[3], false);
                p[7] = __builtin_amdgcn_udot4(lw.y, rb, p[3], false);
                const uint32_t before = wave_prefix_u32(p[SUMS_CPL - 1]) - p[SUMS_CPL - 1];
#pragma unroll
                for (int i = 0; i < SUMS_CPL; ++i)
                    above[i] += p[i] + before;
                if (lane < g.lanes_a) {
                    uint4 *dst = reinterpret_cast<uint4 *>(sat + (y + 1) * g.spitch + 4 + SUMS_CPL * lane);
                    dst[0] = uint4{above[0], above[1], above[2], above[3]};
                    dst[1] = uint4{above[4], above[5], above[6], above[7]};
                }
            }
        }
        SUMS_STAMP(1);      // phase A
        __syncthreads();
        SUMS_STAMP(2);      // waiting for the others' phase A

        // ---- phase B: what the blocks above a block add to each of its columns (a thread per
        // column walks the blocks' last rows), then every entry takes its block's share
        {
            // sixteen lanes = the sixteen blocks of one column: a prefix sum within the rows
            // of 16 lanes of the blocks' last table rows gives what each block hands down
            const int ncols = SUMS_CPL * g.lanes_a, b = lane & 15;
            for (int c = 4 * wave + (lane >> 4); c - (lane >> 4) < ncols; c += 4 * SUMS_WAVES) {
                const bool has = c < ncols && (b + 1) * g.rows_a < g.tr;      // a block below takes it
                uint32_t v = has ? sat[(b + 1) * g.rows_a * g.spitch + 4 + c] : 0u;
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
                if (has)
                    carry[(b + 1) * g.spitch + 4 + c] = v;
            }
        }
        __syncthreads();
        SUMS_STAMP(3);      // block carries (and the wait for them)

        // ---- phase C: the windows' scores at this offset and the selection (the variant that
        // also writes the score volume is a separate copy of the loop)
        auto phase_c = [&](auto with_scores) {
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr) {
                const int wrow = wave + SUMS_WAVES * rr;
                const int wy = wy0 + wrow;
                if (wy >= g.row_end)
                    continue;                                         // wave-uniform
                const int ytop = wrow * g.stride_y, ybot = ytop + g.win_h;   // table rows of the window's corners
                const uint32_t *top = sat + ytop * g.spitch + 3;
                const uint32_t *bot = sat + ybot * g.spitch + 3;
                // table row r holds image row r - 1, which lies in block (r - 1) / rows_a; row 0 is all zero
                const uint32_t *ctop =
                    carry + (ytop > 0 ? (int)__umulhi((uint32_t)(ytop - 1), inv_rows) : 0) * g.spitch + 3;
                const uint32_t *cbot = carry + (int)__umulhi((uint32_t)(ybot - 1), inv_rows) * g.spitch + 3;
#pragma unroll
                for (int j = 0; j < SUMS_COLS_PER_THREAD; ++j) {
                    const int w = rr * SUMS_COLS_PER_THREAD + j;
                    const int x = lane + WAVE * j;
                    const int xw = x0 + x;
                    const int xr = xw + k - g.max_disp;               // the right window's column
                    // (plain & and |: an invalid evaluation becomes NaN through its factor)
                    const bool valid = (xw < g.nwx) & (xr >= 0) & (xr < g.nsx);
                    const int at = r_first[rr] + min(max(xr, 0), g.nsx - 1);
                    const int rs = wsum_r[at];
                    const double ri = valid ? winv_r[at] : NAN;
                    // sum of L * R over the window (the table is block-local: each corner takes what
                    // the blocks above its row add), then of (L - 128)(R - 128): all exact integers
                    const uint32_t lr = (bot[x + g.win_w] - top[x + g.win_w] - bot[x] + top[x]) +
                                        (cbot[x + g.win_w] - ctop[x + g.win_w] - cbot[x] + ctop[x]);
                    const int cross = (int)lr - 128 * (s1l[w] + rs) - 16384 * g.win_w * g.win_h;
                    const double num = fma(n, (double)cross, -((double)s1l[w] * (double)rs));
                    const double s = num * invl[w] * ri;
                    if constexpr (decltype(with_scores)::value) {
                        if (xw < g.nwx)
                            scores_out[((int64_t)wy * g.nwx + xw) * g.nd + k] = s;
                    }
                    // one rule for both modes: arg-max is the rule without a threshold and a tolerance
                    const double lim = best[w] + tol_k;
                    const bool take = (s > thr_k) & ((s > lim) | (centre_rule & (s == lim)));
                    best[w] = take ? s : best[w];
                    arg[w] = take ? k - g.max_disp : arg[w];
                }
            }
        };
        if (scores_out)
            phase_c(std::true_type{});
        else
            phase_c(std::false_type{});
        SUMS_STAMP(6);      // phase C
        __syncthreads();                                              // the table is rebuilt for the next offset
    }
#ifdef SVAMD_SUMS_STAMPS
    if (scores && lane == 0)
        for (int i = 0; i < 8; ++i)
            atomicAdd(reinterpret_cast<unsigned long long *>(scores) + i, (unsigned long long)stamp_acc_[i]);
#endif

#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int wy = wy0 + wave + SUMS_WAVES * rr;
        if (wy >= g.row_end)
            continue;
#pragma unroll
        for (int j = 0; j < SUMS_COLS_PER_THREAD; ++j) {
            const int w = rr * SUMS_COLS_PER_THREAD + j;
            const int xw = x0 + lane + WAVE * j;
            if (xw < g.nwx) {
                const int64_t o = (int64_t)wy * g.nwx + xw;
                out_disp[o] = arg[w];
                out_val[o] = (g.mode == SVAMD_BAND_ARGMAX && arg[w] == SVAMD_BAND_NO_MATCH) ? NAN : best[w];
            }
        }
    }
}

// LDS bytes of a tile of 16 * rpt window rows; 0 when it does not fit.
size_t sums_geometry(SumsGeom &g, int rpt)
{
    const int tcol = SUMS_WC + g.win_w - 1;                           // product columns of a tile
    g.lanes_a = (tcol + SUMS_CPL - 1) / SUMS_CPL;
    if (g.lanes_a > WAVE)
        return 0;
    g.tr = (SUMS_WAVES * rpt - 1) * g.stride_y + g.win_h;
    g.rows_a = (g.tr + SUMS_WAVES - 1) / SUMS_WAVES;
    g.lpitch = SUMS_CPL * g.lanes_a;
    g.rpitch = (int)align_up((size_t)(g.lpitch + g.nd + 8), 8);        // phase A reads 12 bytes from column + (k & ~3)
    g.spitch = 4 + SUMS_CPL * g.lanes_a;
    const size_t lds = (size_t)(g.tr + 1 + SUMS_WAVES) * g.spitch * 4 + (size_t)g.tr * (g.lpitch + g.rpitch);
    return lds <= (size_t)SUMS_LDS_LIMIT ? lds : 0;
}

}  // namespace

bool band_sums_fits(int win_w, int win_h, int stride_y, int max_disp)
{
    SumsGeom g{};
    g.win_w = win_w; g.win_h = win_h; g.stride_y = stride_y; g.max_disp = max_disp; g.nd = 2 * max_disp + 1;
    return sums_geometry(g, 1) != 0;
}

int band_sums_launch(const uint8_t *left, const uint8_t *right, int H, int W, int win_w, int win_h, int stride_y,
                     int max_disp, int mode, double thr, double tol, int row_begin, int row_end,
                     const int32_t *wl, const double *il, const int32_t *wr, const double *ir, int32_t *out_disp,
                     double *out_val, double *scores, hipStream_t st)
{
    SumsGeom g{};
    g.H = H; g.W = W; g.win_w = win_w; g.win_h = win_h; g.stride_y = stride_y; g.max_disp = max_disp;
    g.nd = 2 * max_disp + 1;
    g.nwx = g.nsx = W - win_w + 1;
    g.mode = mode; g.thr = thr; g.tol = tol;
    g.row_begin = row_begin; g.row_end = row_end;
    int rpt = 2;
    size_t lds = sums_geometry(g, rpt);
    if (!lds) {
        rpt = 1;
        lds = sums_geometry(g, rpt);
    }
    SV_REQUIRE(lds > 0, SVAMD_ERR_LIMIT, "band_scan_u8: window %dx%d with %d offsets does not fit the box-sum tile",
               win_w, win_h, g.nd);
    dim3 grid((g.nwx + SUMS_WC - 1) / SUMS_WC, (row_end - row_begin + SUMS_WAVES * rpt - 1) / (SUMS_WAVES * rpt));
    if (rpt == 2) {
        SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&band_sums_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SUMS_LDS_LIMIT));
        hipLaunchKernelGGL(band_sums_kernel<2>, grid, dim3(SUMS_THREADS), lds, st, g, left, right, wl, il, wr, ir,
                           out_disp, out_val, scores);
    } else {
        SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&band_sums_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SUMS_LDS_LIMIT));
        hipLaunchKernelGGL(band_sums_kernel<1>, grid, dim3(SUMS_THREADS), lds, st, g, left, right, wl, il, wr, ir,
                           out_disp, out_val, scores);
    }
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

}  // namespace svamd
