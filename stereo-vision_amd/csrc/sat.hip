// K1: summed-area tables (integral images) by two prefix-scan passes.
//
// Pass 1 scans every image row with one 64-lane wavefront per row (shuffle-based
// inclusive scan, 64 pixels per step, running carry in a register) and writes
// the row prefix of v and v^2 into rows 1..h of the (h+1) x (w+1) tables.
// Pass 2 scans the table columns in place, one thread per column, so that
// neighbouring lanes touch neighbouring addresses (coalesced) at every row.
// Integer inputs use integer accumulators, so window sums derived from the
// tables are exact.  The unsigned 32-bit variant is allowed to wrap: a
// rectangle sum taken as a 4-corner difference in unsigned arithmetic is exact
// whenever the true sum fits in 32 bits.
#include "common.h"
#include "kernels.h"

namespace svamd {

template <typename TAcc>
__device__ __forceinline__ TAcc wave_inclusive_scan(TAcc v, int lane)
{
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        TAcc n = __shfl_up(v, o, WAVE);
        if (lane >= o)
            v += n;
    }
    return v;
}

template <typename TIn, typename TAcc>
__global__ void __launch_bounds__(256)
sat_rows_kernel(const TIn *__restrict__ img, int h, int w, int64_t pitch, TAcc offset,
                const double *__restrict__ device_offset, TAcc *__restrict__ s1, TAcc *__restrict__ s2)
{
    if (device_offset)                 // an offset the host does not know (a mean computed on the stream)
        offset = (TAcc)*device_offset;
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const int64_t tp = (int64_t)w + 1;
    if (row > h)
        return;
    if (row == 0) {  // zero first table row
        for (int x = lane; x <= w; x += WAVE) {
            s1[x] = TAcc(0);
            s2[x] = TAcc(0);
        }
        return;
    }
    const TIn *src = img + (int64_t)(row - 1) * pitch;
    TAcc *d1 = s1 + (int64_t)row * tp;
    TAcc *d2 = s2 + (int64_t)row * tp;
    if (lane == 0) {
        d1[0] = TAcc(0);
        d2[0] = TAcc(0);
    }
    TAcc c1 = TAcc(0), c2 = TAcc(0);
    for (int x0 = 0; x0 < w; x0 += WAVE) {
        int x = x0 + lane;
        TAcc v = x < w ? (TAcc)src[x] - offset : TAcc(0);
        TAcc p1 = wave_inclusive_scan<TAcc>(v, lane) + c1;
        TAcc p2 = wave_inclusive_scan<TAcc>(v * v, lane) + c2;
        if (x < w) {
            d1[x + 1] = p1;
            d2[x + 1] = p2;
        }
        c1 = __shfl(p1, WAVE - 1, WAVE);
        c2 = __shfl(p2, WAVE - 1, WAVE);
    }
}

template <typename TAcc>
__global__ void __launch_bounds__(256)
sat_cols_kernel(int h, int w, TAcc *__restrict__ s1, TAcc *__restrict__ s2)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tp = (int64_t)w + 1;
    if (x > w)
        return;
    TAcc *t = blockIdx.y == 0 ? s1 : s2;
    TAcc run = TAcc(0);
    int y = 1;
    // the scan is a chain of memory round trips: the more rows are requested at a
    // time, the fewer of them
    for (; y + 32 <= h + 1; y += 32) {
        TAcc v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            v[k] = t[(int64_t)(y + k) * tp + x];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            run += v[k];
            t[(int64_t)(y + k) * tp + x] = run;
        }
    }
    for (; y + 8 <= h + 1; y += 8) {
        TAcc v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = t[(int64_t)(y + k) * tp + x];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            run += v[k];
            t[(int64_t)(y + k) * tp + x] = run;
        }
    }
    for (; y <= h; ++y) {
        run += t[(int64_t)y * tp + x];
        t[(int64_t)y * tp + x] = run;
    }
}

// Column pass over row blocks, for tables tall enough that one thread per column
// leaves the chip idle: pass A adds up every (block, column)'s rows, pass B starts each
// block's scan from the sum of the blocks above it.  Integer tables come out
// identical to the single-kernel pass (wrapping arithmetic is associative); floating
// tables differ from it by the rounding of a different summation order.
template <typename TAcc>
__global__ void __launch_bounds__(256)
sat_col_blocksum_kernel(int h, int w, int rows_per_block, const TAcc *__restrict__ s1,
                        const TAcc *__restrict__ s2, TAcc *__restrict__ sums)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tp = (int64_t)w + 1;
    if (x > w)
        return;
    const TAcc *t = blockIdx.z == 0 ? s1 : s2;
    const int y0 = 1 + blockIdx.y * rows_per_block, y1 = min(h, y0 + rows_per_block - 1);
    TAcc run = TAcc(0);
    int y = y0;
    for (; y + 15 <= y1; y += 16) {
        TAcc v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            v[k] = t[(int64_t)(y + k) * tp + x];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            run += v[k];
    }
    for (; y <= y1; ++y)
        run += t[(int64_t)y * tp + x];
    sums[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * tp + x] = run;
}

template <typename TAcc>
__global__ void __launch_bounds__(256)
sat_cols_blocked_kernel(int h, int w, int rows_per_block, TAcc *__restrict__ s1, TAcc *__restrict__ s2,
                        const TAcc *__restrict__ sums)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tp = (int64_t)w + 1;
    if (x > w)
        return;
    TAcc *t = blockIdx.z == 0 ? s1 : s2;
    const TAcc *above = sums + (int64_t)blockIdx.z * gridDim.y * tp + x;
    TAcc run = TAcc(0);
    for (int b = 0; b < (int)blockIdx.y; ++b)
        run += above[(int64_t)b * tp];
    const int y0 = 1 + blockIdx.y * rows_per_block, y1 = min(h, y0 + rows_per_block - 1);
    int y = y0;
    for (; y + 15 <= y1; y += 16) {
        TAcc v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            v[k] = t[(int64_t)(y + k) * tp + x];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            run += v[k];
            t[(int64_t)(y + k) * tp + x] = run;
        }
    }
    for (; y <= y1; ++y) {
        run += t[(int64_t)y * tp + x];
        t[(int64_t)y * tp + x] = run;
    }
}

template <typename TIn, typename TAcc>
int sat2d_launch(const TIn *img, int h, int w, int64_t pitch, TAcc offset, TAcc *s1, TAcc *s2,
                 hipStream_t stream, void *scratch, size_t scratch_bytes, const double *device_offset)
{
    const int waves_per_block = 4;
    dim3 g1((h + 1 + waves_per_block - 1) / waves_per_block);
    hipLaunchKernelGGL((sat_rows_kernel<TIn, TAcc>), g1, dim3(waves_per_block * WAVE), 0, stream,
                       img, h, w, pitch, offset, device_offset, s1, s2);
    SV_LAUNCH_CHECK();
    // row blocks: enough of them for ~64 K threads, at least 8 rows each, 32 at most,
    // and no more than the scratch buffer has room for
    const int64_t tp = (int64_t)w + 1;
    int64_t blocks = (65536 + 2 * tp - 1) / (2 * tp);
    if (blocks > 32) blocks = 32;
    if (blocks > h / 8) blocks = h / 8;
    if (h < 256) blocks = 1;           // short tables: two launches cost more than the chain they shorten
    const int64_t room = scratch ? (int64_t)(scratch_bytes / (2 * tp * sizeof(TAcc))) : 0;
    if (blocks > room) blocks = room;
    if (blocks >= 2) {
        const int rows_per_block = (int)((h + blocks - 1) / blocks);
        const int nb = (h + rows_per_block - 1) / rows_per_block;
        dim3 g((unsigned)((tp + 255) / 256), nb, 2);
        TAcc *sums = static_cast<TAcc *>(scratch);
        hipLaunchKernelGGL((sat_col_blocksum_kernel<TAcc>), g, dim3(256), 0, stream, h, w, rows_per_block, s1, s2,
                           sums);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL((sat_cols_blocked_kernel<TAcc>), g, dim3(256), 0, stream, h, w, rows_per_block, s1, s2,
                           sums);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }
    dim3 g2((w + 1 + 255) / 256, 2);
    hipLaunchKernelGGL((sat_cols_kernel<TAcc>), g2, dim3(256), 0, stream, h, w, s1, s2);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

template int sat2d_launch<uint8_t, uint32_t>(const uint8_t *, int, int, int64_t, uint32_t, uint32_t *,
                                             uint32_t *, hipStream_t, void *, size_t, const double *);
template int sat2d_launch<uint8_t, int64_t>(const uint8_t *, int, int, int64_t, int64_t, int64_t *,
                                            int64_t *, hipStream_t, void *, size_t, const double *);
template int sat2d_launch<float, double>(const float *, int, int, int64_t, double, double *, double *,
                                         hipStream_t, void *, size_t, const double *);
template int sat2d_launch<double, double>(const double *, int, int, int64_t, double, double *, double *,
                                          hipStream_t, void *, size_t, const double *);

}  // namespace svamd

namespace svamd {
int sat2d_any(const void *img, int dtype, int h, int w, int64_t pitch, double offset, void *s1, void *s2,
              hipStream_t st, void *scratch, size_t scratch_bytes, const double *device_offset)
{
    switch (dtype) {
    case SVAMD_U8:
        SV_REQUIRE(offset == std::floor(offset), SVAMD_ERR_INVALID,
                   "svamd_sat2d: offset must be integral for uint8 input");
        return sat2d_launch<uint8_t, int64_t>(static_cast<const uint8_t *>(img), h, w, pitch,
                                              (int64_t)offset, static_cast<int64_t *>(s1),
                                              static_cast<int64_t *>(s2), st, scratch, scratch_bytes, device_offset);
    case SVAMD_F32:
        return sat2d_launch<float, double>(static_cast<const float *>(img), h, w, pitch, offset,
                                           static_cast<double *>(s1), static_cast<double *>(s2), st, scratch,
                                           scratch_bytes, device_offset);
    case SVAMD_F64:
        return sat2d_launch<double, double>(static_cast<const double *>(img), h, w, pitch, offset,
                                            static_cast<double *>(s1), static_cast<double *>(s2), st, scratch,
                                            scratch_bytes, device_offset);
    default:
        SV_REQUIRE(false, SVAMD_ERR_INVALID, "svamd_sat2d: unsupported dtype %d", dtype);
    }
}
}  // namespace svamd

using namespace svamd;

extern "C" int svamd_sat2d(const void *img, int dtype, int h, int w, int64_t pitch, double offset,
                           void *s1, void *s2, void *stream)
{
    SV_REQUIRE(img && s1 && s2, SVAMD_ERR_INVALID, "svamd_sat2d: NULL pointer");
    SV_REQUIRE(h > 0 && w > 0 && pitch >= w, SVAMD_ERR_INVALID, "svamd_sat2d: bad shape %dx%d pitch %lld",
               h, w, (long long)pitch);
    // no workspace in this signature: the column pass runs as one thread per column
    return sat2d_any(img, dtype, h, w, pitch, offset, s1, s2, static_cast<hipStream_t>(stream), nullptr, 0, nullptr);
}
