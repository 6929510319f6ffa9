// Displacement grids: the step after a scan.  Every window paints its dp_x, dp_y,
// |dp| and a contribution count over its box of the full-resolution grids; the
// grids of one sequence are then added to running totals, and the totals are
// finally divided by the contribution counts.
//
// Two producers feed it: the box list of a sequence_scan result (any boxes, any
// order: svamd_accumulate_boxes) and the disparity map of the dense band scan
// (one box per window origin: svamd_band_accumulate).
//
// svamd_accumulate_boxes keeps the reference's floating-point result bit for bit:
// a pixel's value is the sum of its boxes' values in list order, starting from
// zero, and only then is it added to the total.  To find a pixel's boxes in list
// order without a sort, the list is filtered twice by workgroups that split it
// into contiguous chunks (count, prefix sum, fill - so order is preserved):
// first into 256 x 256 pixel coarse tiles, then each coarse tile's list into its
// 32 x 32 pixel fine tiles.  A workgroup per fine tile then walks its list.
#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int FINE = 32, COARSE = 256, FPC = COARSE / FINE;   // fine tiles per coarse tile side
constexpr int BIN_THREADS = 256;

struct BinLevel {
    int tile;            // tile side in pixels
    int tiles_x, tiles_y;
    int parent_tile;     // parent tile side (0: the parent is the whole list)
    int parent_tiles_x;
};

__device__ __forceinline__ bool box_hits(const int32_t *__restrict__ b, int x0, int y0, int side)
{
    // b = {bx0, bx1, by0, by1}, half open; empty boxes hit nothing
    return b[0] < b[1] && b[2] < b[3] && b[0] < x0 + side && b[1] > x0 && b[2] < y0 + side && b[3] > y0;
}

// Block-wide exclusive prefix sum of one int per thread (BIN_THREADS threads).
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    int inc = v;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int up = __shfl_up(inc, o, WAVE);
        if (lane >= o)
            inc += up;
    }
    __syncthreads();
    if (lane == WAVE - 1)
        lds[wid] = inc;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < BIN_THREADS / WAVE; ++w) {
        if (w < wid)
            base += lds[w];
        sum += lds[w];
    }
    *total = sum;
    return base + inc - v;
}

// One workgroup per tile.  The tile's source list (its parent's list, or 0..n-1)
// is split into BIN_THREADS contiguous chunks; fill == 0: count the entries that
// hit the tile; fill == 1: write them, in order, at tile_off[tile].
__global__ void __launch_bounds__(BIN_THREADS)
bin_kernel(BinLevel lv, const int32_t *__restrict__ boxes, int64_t n, const int32_t *__restrict__ parent_list,
           const int64_t *__restrict__ parent_off, int fill, int64_t *__restrict__ tile_cnt,
           const int64_t *__restrict__ tile_off, int32_t *__restrict__ list)
{
    __shared__ int lds[BIN_THREADS / WAVE];
    const int tile = blockIdx.x;
    const int ty = tile / lv.tiles_x, tx = tile - ty * lv.tiles_x;
    const int x0 = tx * lv.tile, y0 = ty * lv.tile;
    int64_t src0 = 0, src_n = n;
    if (parent_list) {
        const int p = (y0 / lv.parent_tile) * lv.parent_tiles_x + x0 / lv.parent_tile;
        src0 = parent_off[p];
        src_n = parent_off[p + 1] - src0;
    }
    const int64_t chunk = (src_n + BIN_THREADS - 1) / BIN_THREADS;
    const int64_t k0 = min((int64_t)threadIdx.x * chunk, src_n), k1 = min(k0 + chunk, src_n);
    int mine = 0;
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t b = parent_list ? parent_list[src0 + k] : k;
        mine += box_hits(boxes + 4 * b, x0, y0, lv.tile) ? 1 : 0;
    }
    int total;
    const int before = block_exclusive_scan(mine, lds, &total);
    if (!fill) {
        if (threadIdx.x == 0)
            tile_cnt[tile] = total;
        return;
    }
    int64_t o = tile_off[tile] + before;
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t b = parent_list ? parent_list[src0 + k] : k;
        if (box_hits(boxes + 4 * b, x0, y0, lv.tile))
            list[o++] = (int32_t)b;
    }
}

// off[0..count] = exclusive prefix sums of cnt[0..count-1]; one block.
__global__ void __launch_bounds__(1024)
offsets_kernel(const int64_t *__restrict__ cnt, int count, int64_t *__restrict__ off)
{
    __shared__ int64_t carry;
    __shared__ int64_t wsum[16];
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (int base = 0; base < count; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int64_t v = i < count ? cnt[i] : 0;
        // inclusive scan inside the block by wave scans + wave totals
        const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
        int64_t inc = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int64_t up = __shfl_up(inc, o, WAVE);
            if (lane >= o)
                inc += up;
        }
        if (lane == WAVE - 1)
            wsum[wid] = inc;
        __syncthreads();
        int64_t wbase = 0;
        for (int w = 0; w < wid; ++w)
            wbase += wsum[w];
        const int64_t start = carry;
        if (i < count)
            off[i] = start + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1)
            carry = start + wbase + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        off[count] = carry;
}

// One workgroup per fine tile, 256 threads x 4 pixels (thread t: row t / 8, columns
// 4 * (t % 8) .. + 3).  List entries are brought into LDS 256 at a time and every
// thread walks them in order.
__global__ void __launch_bounds__(256)
paint_kernel(int tiles_x, int H, int W, const int32_t *__restrict__ boxes, const double *__restrict__ vals,
             const int32_t *__restrict__ list, const int64_t *__restrict__ tile_off,
             double *__restrict__ depth, double *__restrict__ dp_x, double *__restrict__ dp_y,
             double *__restrict__ contrib)
{
    __shared__ int32_t sbox[256][4];
    __shared__ double sval[256][3];
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y = ty * FINE + threadIdx.x / 8;
    const int xb = tx * FINE + 4 * (threadIdx.x % 8);
    double a_depth[4] = {}, a_x[4] = {}, a_y[4] = {}, a_n[4] = {};
    const int64_t begin = tile_off[tile], end = tile_off[tile + 1];
    for (int64_t base = begin; base < end; base += 256) {
        const int m = (int)min((int64_t)256, end - base);
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const int64_t b = list[base + threadIdx.x];
            sbox[threadIdx.x][0] = boxes[4 * b + 0];
            sbox[threadIdx.x][1] = boxes[4 * b + 1];
            sbox[threadIdx.x][2] = boxes[4 * b + 2];
            sbox[threadIdx.x][3] = boxes[4 * b + 3];
            sval[threadIdx.x][0] = vals[3 * b + 0];
            sval[threadIdx.x][1] = vals[3 * b + 1];
            sval[threadIdx.x][2] = vals[3 * b + 2];
        }
        __syncthreads();
        for (int k = 0; k < m; ++k) {
            if (y < sbox[k][2] || y >= sbox[k][3])
                continue;
            const double vx = sval[k][0], vy = sval[k][1], vm = sval[k][2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (xb + p >= sbox[k][0] && xb + p < sbox[k][1]) {
                    a_x[p] += vx;
                    a_y[p] += vy;
                    a_depth[p] += vm;
                    if (vm > 0)
                        a_n[p] += 1.0;
                }
            }
        }
    }
    if (y >= H)
        return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (xb + p < W) {
            const int64_t o = (int64_t)y * W + xb + p;
            depth[o] += a_depth[p];
            dp_x[o] += a_x[p];
            dp_y[o] += a_y[p];
            contrib[o] += a_n[p];
        }
    }
}

// ---- dense band scan: one box per window origin ---------------------------------------
// hs[wy][x] = sums over the windows of row wy that cover column x of d, |d| and (d != 0)
__global__ void __launch_bounds__(256)
band_rows_kernel(const int32_t *__restrict__ disp, int nwy, int nwx, int win_w, int W,
                 int32_t *__restrict__ hs)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nwy * W)
        return;
    const int wy = (int)(idx / W), x = (int)(idx - (int64_t)wy * W);
    const int lo = max(0, x - win_w + 1), hi = min(x, nwx - 1);
    int s = 0, a = 0, c = 0;
    for (int wx = lo; wx <= hi; ++wx) {
        int d = disp[(int64_t)wy * nwx + wx];
        if (d == SVAMD_BAND_NO_MATCH)
            d = 0;
        s += d;
        a += d < 0 ? -d : d;
        c += d != 0;
    }
    const int64_t plane = (int64_t)nwy * W;
    hs[idx] = s;
    hs[plane + idx] = a;
    hs[2 * plane + idx] = c;
}

__global__ void __launch_bounds__(256)
band_cols_kernel(const int32_t *__restrict__ hs, int nwy, int win_h, int stride_y, int H, int W,
                 double *__restrict__ depth, double *__restrict__ dp_x, double *__restrict__ contrib)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W)
        return;
    const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
    // window rows wy with wy * stride_y <= y < wy * stride_y + win_h
    int lo = y - win_h + 1;
    lo = lo <= 0 ? 0 : (lo + stride_y - 1) / stride_y;
    const int hi = min(y / stride_y, nwy - 1);
    const int64_t plane = (int64_t)nwy * W;
    int64_t s = 0, a = 0, c = 0;
    for (int wy = lo; wy <= hi; ++wy) {
        const int64_t o = (int64_t)wy * W + x;
        s += hs[o];
        a += hs[plane + o];
        c += hs[2 * plane + o];
    }
    depth[idx] += (double)a;
    dp_x[idx] += (double)s;
    contrib[idx] += (double)c;
}

__global__ void __launch_bounds__(256)
grid_average_kernel(double *__restrict__ depth, double *__restrict__ dp_x, double *__restrict__ dp_y,
                    double *__restrict__ contrib, int64_t total)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    double c = contrib[idx];
    if (c == 0.0)
        c = INFINITY;
    contrib[idx] = c;
    depth[idx] = depth[idx] / c;
    dp_x[idx] = dp_x[idx] / c;
    dp_y[idx] = dp_y[idx] / c;
}

struct BoxLayout {
    int ctx, cty, ftx, fty;
    size_t coarse_entries, fine_entries;
};
static BoxLayout box_layout(int64_t n, int H, int W, int max_w, int max_h)
{
    BoxLayout L;
    L.ctx = (W + COARSE - 1) / COARSE;
    L.cty = (H + COARSE - 1) / COARSE;
    L.ftx = L.ctx * FPC;              // fine tiles cover whole coarse tiles
    L.fty = L.cty * FPC;
    auto span = [](int len, int side, int tiles) {
        int s = (len + side - 2) / side + 1;   // tiles a run of `len` pixels can touch
        return (size_t)(s < tiles ? s : tiles);
    };
    L.coarse_entries = (size_t)n * span(max_w, COARSE, L.ctx) * span(max_h, COARSE, L.cty);
    L.fine_entries = (size_t)n * span(max_w, FINE, L.ftx) * span(max_h, FINE, L.fty);
    return L;
}

}  // namespace svamd

using namespace svamd;

extern "C" size_t svamd_accumulate_boxes_workspace_bytes(int64_t n, int H, int W, int max_box_w, int max_box_h)
{
    if (n < 0 || H <= 0 || W <= 0 || max_box_w <= 0 || max_box_h <= 0)
        return 0;
    BoxLayout L = box_layout(n, H, W, max_box_w, max_box_h);
    const size_t nc = (size_t)L.ctx * L.cty, nf = (size_t)L.ftx * L.fty;
    return 2 * align_up((nc + 1) * 8, 256) + 2 * align_up((nf + 1) * 8, 256) +
           align_up(L.coarse_entries * 4, 256) + align_up(L.fine_entries * 4, 256) + 256;
}

extern "C" int svamd_accumulate_boxes(const int32_t *boxes, const double *vals, int64_t n, int H, int W,
                                      int max_box_w, int max_box_h, double *depth, double *dp_x,
                                      double *dp_y, double *contrib, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(depth && dp_x && dp_y && contrib, SVAMD_ERR_INVALID, "accumulate_boxes: NULL grid");
    SV_REQUIRE(n >= 0 && n < (1LL << 31) && H > 0 && W > 0 && max_box_w > 0 && max_box_h > 0,
               SVAMD_ERR_INVALID, "accumulate_boxes: bad sizes");
    if (n == 0)
        return SVAMD_OK;
    SV_REQUIRE(boxes && vals, SVAMD_ERR_INVALID, "accumulate_boxes: NULL box list");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_accumulate_boxes_workspace_bytes(n, H, W, max_box_w, max_box_h),
               SVAMD_ERR_WORKSPACE, "accumulate_boxes: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BoxLayout L = box_layout(n, H, W, max_box_w, max_box_h);
    const int nc = L.ctx * L.cty, nf = L.ftx * L.fty;
    Carver ws(workspace);
    int64_t *c_cnt = ws.take<int64_t>(nc + 1);
    int64_t *c_off = ws.take<int64_t>(nc + 1);
    int64_t *f_cnt = ws.take<int64_t>(nf + 1);
    int64_t *f_off = ws.take<int64_t>(nf + 1);
    int32_t *c_list = ws.take<int32_t>(L.coarse_entries);
    int32_t *f_list = ws.take<int32_t>(L.fine_entries);

    BinLevel coarse{COARSE, L.ctx, L.cty, 0, 0};
    BinLevel fine{FINE, L.ftx, L.fty, COARSE, L.ctx};
    for (int fill = 0; fill < 2; ++fill) {
        hipLaunchKernelGGL(bin_kernel, dim3(nc), dim3(BIN_THREADS), 0, st, coarse, boxes, n,
                           (const int32_t *)nullptr, (const int64_t *)nullptr, fill, c_cnt, c_off, c_list);
        SV_LAUNCH_CHECK();
        if (!fill) {
            hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, st, c_cnt, nc, c_off);
            SV_LAUNCH_CHECK();
        }
    }
    for (int fill = 0; fill < 2; ++fill) {
        hipLaunchKernelGGL(bin_kernel, dim3(nf), dim3(BIN_THREADS), 0, st, fine, boxes, n, c_list, c_off, fill,
                           f_cnt, f_off, f_list);
        SV_LAUNCH_CHECK();
        if (!fill) {
            hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, st, f_cnt, nf, f_off);
            SV_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(paint_kernel, dim3(nf), dim3(256), 0, st, L.ftx, H, W, boxes, vals, f_list, f_off, depth,
                       dp_x, dp_y, contrib);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" size_t svamd_band_accumulate_workspace_bytes(int H, int W, int win_h, int stride_y)
{
    if (H <= 0 || W <= 0 || win_h <= 0 || win_h > H || stride_y <= 0)
        return 0;
    const size_t nwy = (size_t)(H - win_h) / stride_y + 1;
    return align_up(3 * nwy * (size_t)W * 4, 256);
}

extern "C" int svamd_band_accumulate(const int32_t *disp, int H, int W, int win_w, int win_h, int stride_y,
                                     double *depth, double *dp_x, double *contrib, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(disp && depth && dp_x && contrib, SVAMD_ERR_INVALID, "band_accumulate: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && win_w > 0 && win_h > 0 && stride_y > 0, SVAMD_ERR_INVALID,
               "band_accumulate: non-positive size");
    SV_REQUIRE(win_w <= W && win_h <= H, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_band_accumulate_workspace_bytes(H, W, win_h, stride_y),
               SVAMD_ERR_WORKSPACE, "band_accumulate: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nwy = (H - win_h) / stride_y + 1, nwx = W - win_w + 1;
    int32_t *hs = static_cast<int32_t *>(workspace);
    const int64_t t1 = (int64_t)nwy * W, t2 = (int64_t)H * W;
    hipLaunchKernelGGL(band_rows_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, disp, nwy, nwx,
                       win_w, W, hs);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(band_cols_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, hs, nwy, win_h,
                       stride_y, H, W, depth, dp_x, contrib);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" int svamd_grid_average(double *depth, double *dp_x, double *dp_y, double *contrib, int64_t total,
                                  void *stream)
{
    SV_REQUIRE(depth && dp_x && dp_y && contrib && total > 0, SVAMD_ERR_INVALID, "grid_average: bad argument");
    hipLaunchKernelGGL(grid_average_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), depth, dp_x, dp_y, contrib, total);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
