// Shared host/device helpers of libsvamd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/svamd.h"

namespace svamd {

void set_error(const char *fmt, ...);

#define SV_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::svamd::set_error(__VA_ARGS__); \
            return (code);                   \
        }                                    \
    } while (0)

#define SV_HIP(expr)                                                                    \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::svamd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                               __FILE__, __LINE__);                                     \
            return SVAMD_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

#define SV_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::svamd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                     \
            return SVAMD_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

constexpr int WAVE = 64;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves 256-byte aligned sub-buffers out of a caller-provided workspace.
struct Carver {
    char *base;
    size_t used = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T>
    T *take(size_t count)
    {
        T *p = reinterpret_cast<T *>(base + used);
        used += align_up(count * sizeof(T), 256);
        return p;
    }
};

// Statistics of one array, produced on the device and consumed by later kernels
// of the same call without a host round trip.
struct Stats {
    double mean;
    double std;    // population standard deviation (ddof 0)
    double count;  // number of elements that entered (non-NaN when NaNs are ignored)
    double pad;
};

// ---- device helpers ---------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, WAVE);
    return v;
}

// The float64 sum of a wavefront through the vector ALU's lane-shift operands
// (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast 15 and 31 across
// the rows: the total arrives in lane 63 and is broadcast from there).  Shuffles
// (ds_bpermute) would occupy the LDS pipeline, which the kernels that call this
// between their staging phases keep busy already.  Lanes without a source add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ double wave_sum<double>(double v)
{
    v += dpp_shifted<0x111, 0xF>(v);
    v += dpp_shifted<0x112, 0xF>(v);
    v += dpp_shifted<0x114, 0xF>(v);
    v += dpp_shifted<0x118, 0xF>(v);
    v += dpp_shifted<0x142, 0xA>(v);
    v += dpp_shifted<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Sum over a block of up to 1024 threads; every thread receives the total.
// `red` points to at least 17 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    int nw = (blockDim.x + WAVE - 1) / WAVE;
    __syncthreads();
    if (lane == 0)
        red[wid] = v;
    __syncthreads();
    if (wid == 0) {
        double t = lane < nw ? red[lane] : 0.0;
        t = wave_sum(t);
        if (lane == 0)
            red[16] = t;
    }
    __syncthreads();
    return red[16];
}

__device__ __forceinline__ double load_as_double(const void *p, int dtype, int64_t i)
{
    if (dtype == SVAMD_U8)
        return (double)static_cast<const uint8_t *>(p)[i];
    if (dtype == SVAMD_F32)
        return (double)static_cast<const float *>(p)[i];
    return static_cast<const double *>(p)[i];
}

}  // namespace svamd
