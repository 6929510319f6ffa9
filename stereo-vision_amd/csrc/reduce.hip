// NaN-ignoring arg-max of a float64 array (first occurrence), the reduction
// behind utils.max_pos.
#include "common.h"
#include "kernels.h"

namespace svamd {

__global__ void __launch_bounds__(1024)
nanargmax_kernel(const double *__restrict__ data, int64_t n, int64_t *__restrict__ out)
{
    __shared__ double sv[16];
    __shared__ int64_t si[16];
    double best = -INFINITY;
    int64_t arg = INT64_MAX;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = data[i];
        // ascending i per thread: strict > keeps the first occurrence
        if (!isnan(v) && (v > best || arg == INT64_MAX)) {
            best = v;
            arg = i;
        }
    }
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o, WAVE);
        const int64_t oi = __shfl_xor(arg, o, WAVE);
        if (oi != INT64_MAX && (arg == INT64_MAX || ov > best || (ov == best && oi < arg))) {
            best = ov;
            arg = oi;
        }
    }
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (lane == 0) {
        sv[wid] = best;
        si[wid] = arg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / WAVE); ++w) {
            if (si[w] != INT64_MAX && (arg == INT64_MAX || sv[w] > best || (sv[w] == best && si[w] < arg))) {
                best = sv[w];
                arg = si[w];
            }
        }
        *out = arg == INT64_MAX ? -1 : arg;
    }
}

}  // namespace svamd

using namespace svamd;

extern "C" int svamd_nanargmax(const double *data, int64_t n, int64_t *out_index, void *stream)
{
    SV_REQUIRE(data && out_index && n > 0, SVAMD_ERR_INVALID, "nanargmax: NULL pointer or empty array");
    hipLaunchKernelGGL(nanargmax_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), data, n,
                       out_index);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
