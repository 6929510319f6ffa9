// First stage of the stereo scan without a host-side list of pairs: the windows tile
// the left image on a regular grid and every window's candidate regions are the cross
// product of per-axis candidate positions, so the host uploads two small per-axis
// tables and this file derives each (window, candidate) pair's boxes on the device,
// runs the pair kernels and the selection rule over a range of window rows and leaves
// one (candidate slot, correlation) per window.  Memory is proportional to the pairs of
// the row range the caller picks, not to the scan.
#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int CAND_FIELDS = SVAMD_GRID_CAND_FIELDS;
constexpr int XWIN_FIELDS = SVAMD_GRID_XWIN_FIELDS;
constexpr int YWIN_FIELDS = SVAMD_GRID_YWIN_FIELDS;
constexpr int FLAG_REJECTED = 1, FLAG_AT_CENTRE = 2;

struct GridTables {
    const int32_t *xcand, *xwin, *ycand, *ywin;
    const int64_t *row_first;
    int ncol, nx, nrow, ny;
};

// One thread per (window of the row range, candidate slot).  Slot 0 is the window's own
// position, slot 1 + iy * nx + ix the scheme grid row by row; candidates that the
// reference's boundary test rejects, and the grid candidate that coincides with the
// window's own centre, take no pair.  The surviving candidates of a window are numbered
// in slot order from the window's first pair, which follows from the per-axis counts.
__global__ void __launch_bounds__(256)
grid_enumerate_kernel(GridTables g, int row_begin, int64_t nwin, int64_t npairs, int64_t left_pitch,
                      int64_t right_pitch, int64_t *__restrict__ desc, uint8_t *__restrict__ is_centre,
                      int32_t *__restrict__ pair_slot, int64_t *__restrict__ win_off)
{
    const int slots = 1 + g.nx * g.ny;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nwin * slots)
        return;
    const int64_t w = t / slots;
    const int slot = (int)(t - w * slots);
    const int r = row_begin + (int)(w / g.ncol), c = (int)(w % g.ncol);
    const int32_t *xw = g.xwin + (size_t)c * XWIN_FIELDS, *yw = g.ywin + (size_t)r * YWIN_FIELDS;
    const int own_ok = xw[2] & yw[2];
    const int64_t first = g.row_first[r] - g.row_first[row_begin] + (int64_t)yw[2] * xw[5] +
                          (int64_t)yw[3] * xw[6] - (int64_t)yw[4] * xw[7];
    const int32_t *xc = g.xcand + ((size_t)c * (1 + g.nx)) * CAND_FIELDS;
    const int32_t *yc = g.ycand + ((size_t)r * (1 + g.ny)) * CAND_FIELDS;
    bool valid;
    int rank = 0;
    if (slot == 0) {
        win_off[w] = first;
        if (w == nwin - 1)
            win_off[nwin] = first + own_ok + (int64_t)yw[3] * xw[3] - (int64_t)yw[4] * xw[4];
        valid = own_ok != 0;
    } else {
        const int iy = (slot - 1) / g.nx, ix = (slot - 1) - iy * g.nx;
        xc += (size_t)(1 + ix) * CAND_FIELDS;
        yc += (size_t)(1 + iy) * CAND_FIELDS;
        const bool xrej = xc[2] & FLAG_REJECTED, yrej = yc[2] & FLAG_REJECTED;
        const bool xac = xc[2] & FLAG_AT_CENTRE, yac = yc[2] & FLAG_AT_CENTRE;
        valid = !(xrej || yrej) && !(xac && yac);
        // surviving candidates before this one: whole grid rows above, the columns to the
        // left, less the own-centre duplicates among them
        rank = own_ok + yc[3] * xw[3] + xc[3] - (yc[4] * xw[4] + (yac ? xc[4] : 0));
    }
    if (!valid)
        return;
    const int64_t p = first + rank;
    if (p < 0 || p >= npairs)        // tables that do not add up: write nothing out of bounds
        return;
    int64_t *d = desc + 8 * p;
    d[0] = (int64_t)yw[0] * left_pitch + xw[0];
    d[1] = left_pitch;
    d[2] = yw[1] - yw[0];
    d[3] = xw[1] - xw[0];
    d[4] = (int64_t)yc[0] * right_pitch + xc[0];
    d[5] = right_pitch;
    d[6] = yc[1] - yc[0];
    d[7] = xc[1] - xc[0];
    is_centre[p] = slot == 0;
    pair_slot[p] = slot;
}

// Windows whose selection came within `guard` of flipping: all their pairs go onto the
// list that the float64 kernel works through (in any order: results land by pair index).
__global__ void __launch_bounds__(256)
grid_collect_close_kernel(const double *__restrict__ margin, const int64_t *__restrict__ win_off, int64_t nwin,
                          double guard, int32_t *__restrict__ list, int32_t *__restrict__ count)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= nwin || !(margin[w] < guard))
        return;
    const int64_t p0 = win_off[w], n = win_off[w + 1] - p0;
    if (n <= 0)
        return;
    const int32_t at = atomicAdd(count, (int32_t)n);
    for (int64_t j = 0; j < n; ++j)
        list[at + j] = (int32_t)(p0 + j);
}

// The selected pair of every window that selected one.
__global__ void __launch_bounds__(256)
grid_collect_chosen_kernel(const int64_t *__restrict__ best_pair, int64_t nwin, int32_t *__restrict__ list,
                           int32_t *__restrict__ count)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= nwin || best_pair[w] < 0)
        return;
    list[atomicAdd(count, 1)] = (int32_t)best_pair[w];
}

__global__ void __launch_bounds__(256)
grid_finish_kernel(const int64_t *__restrict__ best_pair, const double *__restrict__ corr,
                   const int32_t *__restrict__ pair_slot, int64_t nwin, int32_t *__restrict__ best_slot,
                   double *__restrict__ best_val)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= nwin)
        return;
    const int64_t p = best_pair[w];
    best_slot[w] = p < 0 ? -1 : pair_slot[p];
    best_val[w] = p < 0 ? 0.0 : corr[p];
}

struct GridSpace {
    int64_t *desc;
    double *corr;
    uint8_t *is_centre;
    int32_t *pair_slot, *list;
    int64_t *win_off, *best_pair;
    double *margin, *fold_val;
    int32_t *counts;
    size_t bytes;
};

static GridSpace grid_carve(void *workspace, int64_t nwin, int64_t npairs)
{
    Carver c(workspace);
    GridSpace s;
    const size_t np = (size_t)(npairs > 0 ? npairs : 1), nw = (size_t)(nwin > 0 ? nwin : 1);
    s.desc = c.take<int64_t>(8 * np);
    s.corr = c.take<double>(np);
    s.is_centre = c.take<uint8_t>(np);
    s.pair_slot = c.take<int32_t>(np);
    s.list = c.take<int32_t>(np);
    s.win_off = c.take<int64_t>(nw + 1);
    s.best_pair = c.take<int64_t>(nw);
    s.margin = c.take<double>(nw);
    s.fold_val = c.take<double>(nw);
    s.counts = c.take<int32_t>(64);
    s.bytes = c.used;
    return s;
}

}  // namespace svamd

using namespace svamd;

extern "C" size_t svamd_scan_grid_workspace_bytes(int64_t nwin, int64_t npairs)
{
    return grid_carve(nullptr, nwin, npairs).bytes;
}

extern "C" int svamd_scan_grid(const double *left, int64_t left_pitch, const double *right, int64_t right_pitch,
                               const int32_t *xcand, const int32_t *xwin, int ncol, int nx,
                               const int32_t *ycand, const int32_t *ywin, int nrow, int ny,
                               const int64_t *row_first, int row_begin, int row_end, int64_t npairs,
                               int max_side, double min_std, double min_mean_diff, double threshold,
                               double centre_tol, int precision, double guard_band, int32_t *best_slot,
                               double *best_val, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(ncol > 0 && nrow > 0 && nx > 0 && ny > 0, SVAMD_ERR_INVALID, "scan_grid: empty window grid or scheme");
    SV_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= nrow, SVAMD_ERR_INVALID,
               "scan_grid: window rows [%d, %d) outside 0..%d", row_begin, row_end, nrow);
    const int64_t nwin = (int64_t)(row_end - row_begin) * ncol;
    if (nwin == 0)
        return SVAMD_OK;
    SV_REQUIRE(left && right && xcand && xwin && ycand && ywin && row_first && best_slot && best_val,
               SVAMD_ERR_INVALID, "scan_grid: NULL pointer");
    const int64_t slots = 1 + (int64_t)nx * ny;
    SV_REQUIRE(npairs >= 0 && npairs <= nwin * slots, SVAMD_ERR_INVALID,
               "scan_grid: %lld pairs for %lld windows of %lld candidates", (long long)npairs, (long long)nwin,
               (long long)slots);
    SV_REQUIRE(npairs < ((int64_t)1 << 31), SVAMD_ERR_LIMIT,
               "scan_grid: %lld pairs in one call; scan fewer window rows at a time", (long long)npairs);
    SV_REQUIRE(max_side > 0 && max_side <= SVAMD_PAIR_MAX_SIDE, SVAMD_ERR_LIMIT,
               "scan_grid: max_side %d outside 1..%d", max_side, SVAMD_PAIR_MAX_SIDE);
    SV_REQUIRE(precision == SVAMD_ACC_F32 || precision == SVAMD_ACC_F64 || precision == SVAMD_ACC_MIXED,
               SVAMD_ERR_INVALID, "scan_grid: unknown precision %d", precision);
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, SVAMD_ERR_WORKSPACE,
               "scan_grid: workspace missing or not 256-byte aligned");
    const GridSpace s = grid_carve(workspace, nwin, npairs);
    SV_REQUIRE(workspace_bytes >= s.bytes, SVAMD_ERR_WORKSPACE, "scan_grid: workspace of %zu bytes, %zu needed",
               workspace_bytes, s.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t *out_slot = best_slot + (int64_t)row_begin * ncol;
    double *out_val = best_val + (int64_t)row_begin * ncol;
    const unsigned wblocks = (unsigned)((nwin + 255) / 256);

    GridTables g{xcand, xwin, ycand, ywin, row_first, ncol, nx, nrow, ny};
    const int64_t threads = nwin * slots;
    SV_REQUIRE((threads + 255) / 256 < ((int64_t)1 << 31), SVAMD_ERR_LIMIT,
               "scan_grid: too many candidates in one call; scan fewer window rows at a time");
    // tables whose counts do not add up to `npairs` leave descriptors unwritten: those
    // must read as empty boxes (a NaN pair), not as whatever the workspace held
    if (npairs > 0)
        SV_HIP(hipMemsetAsync(s.desc, 0, (size_t)npairs * 8 * sizeof(int64_t), st));
    hipLaunchKernelGGL(grid_enumerate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, g, row_begin,
                       nwin, npairs, left_pitch, right_pitch, s.desc, s.is_centre, s.pair_slot, s.win_off);
    SV_LAUNCH_CHECK();
    int rc = SVAMD_OK;
    if (npairs > 0) {
        rc = launch_pairs(left, right, s.desc, npairs, max_side, min_std, min_mean_diff,
                          precision == SVAMD_ACC_F64 ? SVAMD_ACC_F64 : SVAMD_ACC_F32, s.corr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, 0, st);
        if (rc != SVAMD_OK)
            return rc;
    }
    auto fold = [&]() {
        return svamd_scan_fold(s.corr, s.is_centre, nullptr, s.win_off, nwin, threshold, centre_tol, s.best_pair,
                               s.fold_val, s.margin, stream);
    };
    if ((rc = fold()) != SVAMD_OK)
        return rc;
    if (precision == SVAMD_ACC_MIXED && npairs > 0) {
        SV_HIP(hipMemsetAsync(s.counts, 0, 64 * sizeof(int32_t), st));
        // decisions within the guard band of flipping: their windows again in float64
        hipLaunchKernelGGL(grid_collect_close_kernel, dim3(wblocks), dim3(256), 0, st, s.margin, s.win_off, nwin,
                           guard_band, s.list, s.counts);
        SV_LAUNCH_CHECK();
        const unsigned redo_grid = (unsigned)(npairs < 16384 ? npairs : 16384);
        rc = launch_pairs(left, right, s.desc, npairs, max_side, min_std, min_mean_diff, SVAMD_ACC_F64, s.corr,
                          nullptr, nullptr, nullptr, s.list, s.counts, redo_grid, st);
        if (rc != SVAMD_OK || (rc = fold()) != SVAMD_OK)
            return rc;
        // the value a window reports is its selected pair's float64 maximum
        hipLaunchKernelGGL(grid_collect_chosen_kernel, dim3(wblocks), dim3(256), 0, st, s.best_pair, nwin, s.list,
                           s.counts + 32);
        SV_LAUNCH_CHECK();
        const unsigned chosen_grid = (unsigned)(nwin < 16384 ? nwin : 16384);
        rc = launch_pairs(left, right, s.desc, npairs, max_side, min_std, min_mean_diff, SVAMD_ACC_F64, s.corr,
                          nullptr, nullptr, nullptr, s.list, s.counts + 32, chosen_grid, st);
        if (rc != SVAMD_OK)
            return rc;
    }
    hipLaunchKernelGGL(grid_finish_kernel, dim3(wblocks), dim3(256), 0, st, s.best_pair, s.corr, s.pair_slot, nwin,
                       out_slot, out_val);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
