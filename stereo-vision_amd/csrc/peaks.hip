// Local-extremum mask of a correlation surface: the device part of the
// calibration peak finder.  A pixel is kept when it equals the maximum of its
// size x size neighbourhood and that neighbourhood's maximum exceeds its minimum
// by more than a threshold.  Neighbourhoods follow scipy.ndimage's conventions
// for maximum_filter / minimum_filter with a scalar size: offsets
// -(size / 2) .. size - size / 2 - 1 along each axis, borders reflected about the
// edge (d c b a | a b c d | d c b a).  The square filter is separable: a row
// pass into the workspace, then a column pass fused with the test.
#include "common.h"
#include "kernels.h"

namespace svamd {

__device__ __forceinline__ int reflect_index(int i, int n)
{
    // period 2n: ... 1 0 | 0 1 ... n-1 | n-1 n-2 ...
    const int p = 2 * n;
    i %= p;
    if (i < 0)
        i += p;
    return i < n ? i : p - 1 - i;
}

__global__ void __launch_bounds__(256)
extrema_rows_kernel(const double *__restrict__ data, int H, int W, int size, double *__restrict__ rmax,
                    double *__restrict__ rmin)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W)
        return;
    const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
    const double *row = data + (int64_t)y * W;
    const int lo = -(size / 2);
    double mx = -INFINITY, mn = INFINITY;
    for (int k = 0; k < size; ++k) {
        const double v = row[reflect_index(x + lo + k, W)];
        mx = fmax(mx, v);
        mn = fmin(mn, v);
    }
    rmax[idx] = mx;
    rmin[idx] = mn;
}

__global__ void __launch_bounds__(256)
extrema_cols_kernel(const double *__restrict__ data, const double *__restrict__ rmax,
                    const double *__restrict__ rmin, int H, int W, int size, double threshold,
                    uint8_t *__restrict__ mask)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W)
        return;
    const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
    const int lo = -(size / 2);
    double mx = -INFINITY, mn = INFINITY;
    for (int k = 0; k < size; ++k) {
        const int64_t o = (int64_t)reflect_index(y + lo + k, H) * W + x;
        mx = fmax(mx, rmax[o]);
        mn = fmin(mn, rmin[o]);
    }
    mask[idx] = (data[idx] == mx && (mx - mn) > threshold) ? 1 : 0;
}

}  // namespace svamd

using namespace svamd;

extern "C" size_t svamd_peak_mask_workspace_bytes(int H, int W)
{
    if (H <= 0 || W <= 0)
        return 0;
    return 2 * align_up((size_t)H * W * sizeof(double), 256);
}

extern "C" int svamd_peak_mask(const double *data, int H, int W, int size, double threshold, uint8_t *mask,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(data && mask, SVAMD_ERR_INVALID, "peak_mask: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && size > 0, SVAMD_ERR_INVALID, "peak_mask: non-positive size");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_peak_mask_workspace_bytes(H, W),
               SVAMD_ERR_WORKSPACE, "peak_mask: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver ws(workspace);
    double *rmax = ws.take<double>((size_t)H * W);
    double *rmin = ws.take<double>((size_t)H * W);
    const unsigned grid = (unsigned)(((int64_t)H * W + 255) / 256);
    hipLaunchKernelGGL(extrema_rows_kernel, dim3(grid), dim3(256), 0, st, data, H, W, size, rmax, rmin);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(extrema_cols_kernel, dim3(grid), dim3(256), 0, st, data, rmax, rmin, H, W, size, threshold,
                       mask);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
