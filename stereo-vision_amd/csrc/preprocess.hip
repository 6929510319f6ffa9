// Image preparation ahead of the scans, on the device: colour pixels to grey
// levels (with transparent pixels marked NaN) and anti-aliased resizing.
//
// Grey levels follow PIL's "L" conversion, an integer formula:
//     L = (19595 R + 38470 G + 7471 B + 32768) >> 16.
//
// Resizing follows scikit-image 0.19's `resize` with its defaults, which is
// three scipy.ndimage steps: values of integer images scaled by 1 / 255; when
// the output is smaller along an axis, a Gaussian filter with sigma =
// (input / output - 1) / 2 along that axis (kernel radius int(4 sigma + 0.5),
// borders mirrored about the edge pixel: c b | a b c | b a); then linear
// interpolation at the grid positions (k + 0.5) * input / output - 0.5 with the
// same mirrored borders; the result clipped to the value range of the scaled
// input.  The interpolation adds its four products in scipy's order, without
// fused multiply-adds, so it is bit-identical to scipy's on the same input; the
// Gaussian weights come from the device's exp, so the filtered values agree to
// rounding.
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace svamd {

constexpr int PRE_MAX_RADIUS = 255;
constexpr int PRE_PARTIALS = 2048;

template <typename T> __device__ __forceinline__ double as_unit(T v);
template <> __device__ __forceinline__ double as_unit<uint8_t>(uint8_t v)
{
    return __dmul_rn((double)v, 1.0 / 255.0);
}
template <> __device__ __forceinline__ double as_unit<double>(double v) { return v; }

// period 2n - 2: ... 2 1 | 0 1 ... n-1 | n-2 n-3 ...
__device__ __forceinline__ int mirror_index(int i, int n)
{
    if (n == 1)
        return 0;
    const int p = 2 * n - 2;
    i %= p;
    if (i < 0)
        i += p;
    return i < n ? i : p - i;
}

__global__ void __launch_bounds__(256)
greyscale_kernel(const uint8_t *__restrict__ pix, int channels, int64_t total, int nan_alpha,
                 uint8_t *__restrict__ out8, double *__restrict__ out64)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    uint32_t grey, alpha = 255;
    if (channels == 4) {
        const uchar4 p = reinterpret_cast<const uchar4 *>(pix)[idx];
        grey = (19595u * p.x + 38470u * p.y + 7471u * p.z + 32768u) >> 16;
        alpha = p.w;
    } else if (channels == 3) {
        const uint8_t *p = pix + 3 * idx;
        grey = (19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 32768u) >> 16;
    } else if (channels == 2) {          // grey + alpha
        grey = pix[2 * idx];
        alpha = pix[2 * idx + 1];
    } else {
        grey = pix[idx];
    }
    if (out8)
        out8[idx] = (uint8_t)grey;
    if (out64)
        out64[idx] = (nan_alpha && alpha == 0) ? NAN : (double)grey;
}

// weights[0 .. 2r]: exp(-x^2 / (2 sigma^2)) for x = -r .. r, divided by their sum
__global__ void __launch_bounds__(256) gaussian_weights_kernel(double sigma, int radius, double *__restrict__ weights)
{
    __shared__ double phi[2 * PRE_MAX_RADIUS + 1];
    __shared__ double total;
    const int n = 2 * radius + 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)(i - radius);
        phi[i] = exp(-0.5 / (sigma * sigma) * x * x);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < n; ++i)
            s += phi[i];
        total = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        weights[i] = phi[i] / total;
}

// One axis of the Gaussian filter.  The sum runs as scipy's symmetric correlate1d
// does: the centre tap first, then the pairs from the farthest inwards.
template <typename T, int AXIS>
__global__ void __launch_bounds__(256)
gaussian_axis_kernel(const T *__restrict__ in, int H, int W, const double *__restrict__ weights, int radius,
                     double *__restrict__ out)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W)
        return;
    const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
    const int n = AXIS == 0 ? H : W;
    const int c = AXIS == 0 ? y : x;
    const int64_t stride = AXIS == 0 ? W : 1;
    const T *line = AXIS == 0 ? in + x : in + (int64_t)y * W;
    double acc = __dmul_rn(as_unit<T>(line[c * stride]), weights[radius]);
    const bool inside = c - radius >= 0 && c + radius < n;
    for (int k = radius; k >= 1; --k) {
        const int a = inside ? c - k : mirror_index(c - k, n);
        const int b = inside ? c + k : mirror_index(c + k, n);
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(as_unit<T>(line[a * stride]), as_unit<T>(line[b * stride])),
                                       weights[radius - k]));
    }
    out[idx] = acc;
}

// Smallest and largest non-NaN value of the image (as the scaled values the resize
// works on): per-block partials from 16-byte loads, then one block folds them into
// partial[0], partial[1].
template <typename T> struct Pack16;
template <> struct Pack16<uint8_t> {
    static constexpr int N = 16;
    uint4 raw;
    __device__ __forceinline__ double get(int i) const
    {
        const uint32_t w = i < 4 ? raw.x : i < 8 ? raw.y : i < 12 ? raw.z : raw.w;
        return as_unit<uint8_t>((uint8_t)(w >> (8 * (i & 3))));
    }
};
template <> struct Pack16<double> {
    static constexpr int N = 2;
    double2 raw;
    __device__ __forceinline__ double get(int i) const { return i ? raw.y : raw.x; }
};

template <typename T>
__global__ void __launch_bounds__(256)
range_partial_kernel(const T *__restrict__ in, int64_t total, double *__restrict__ partial)
{
    double lo = INFINITY, hi = -INFINITY;
    constexpr int N = Pack16<T>::N;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const int64_t packs = total / N;
        const Pack16<T> *pin = reinterpret_cast<const Pack16<T> *>(in);
        for (int64_t i = tid; i < packs; i += nthreads) {
            const Pack16<T> p = pin[i];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double v = p.get(k);
                lo = fmin(lo, v);           // fmin / fmax drop NaNs
                hi = fmax(hi, v);
            }
        }
        done = packs * N;
    }
    for (int64_t i = done + tid; i < total; i += nthreads) {
        const double v = as_unit<T>(in[i]);
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    __shared__ double slo[256], shi[256];
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = slo[0];
        partial[2 * blockIdx.x + 1] = shi[0];
    }
}

__global__ void __launch_bounds__(256) range_final_kernel(double *__restrict__ partial, int npartial)
{
    double lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < npartial; i += blockDim.x) {
        lo = fmin(lo, partial[2 * i]);
        hi = fmax(hi, partial[2 * i + 1]);
    }
    __shared__ double slo[256], shi[256];
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[0] = slo[0];
        partial[1] = shi[0];
    }
}

// scipy's coordinate folding for mirrored borders (ni_interpolation.c, map_coordinate)
__device__ __forceinline__ double mirror_coordinate(double in, int len)
{
    if (in < 0) {
        if (len <= 1)
            return 0;
        const int sz2 = 2 * len - 2;
        in = sz2 * (double)(int64_t)(-in / sz2) + in;
        return in <= 1 - len ? in + sz2 : -in;
    }
    if (in > len - 1) {
        if (len <= 1)
            return 0;
        const int sz2 = 2 * len - 2;
        in -= sz2 * (double)(int64_t)(in / sz2);
        if (in > len - 1)
            in = sz2 - in;
    }
    return in;
}

template <typename T>
__global__ void __launch_bounds__(256)
zoom_linear_kernel(const T *__restrict__ in, int H, int W, int OH, int OW, double zoom_y, double zoom_x,
                   const double *__restrict__ partial, int npartial, double *__restrict__ out)
{
    __shared__ double range[2];
    if (threadIdx.x < 2) {
        double v = threadIdx.x == 0 ? INFINITY : -INFINITY;
        for (int i = 0; i < npartial; ++i)
            v = threadIdx.x == 0 ? fmin(v, partial[2 * i]) : fmax(v, partial[2 * i + 1]);
        range[threadIdx.x] = v;
    }
    __syncthreads();
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)OH * OW)
        return;
    const int oy = (int)(idx / OW), ox = (int)(idx - (int64_t)oy * OW);
    const double cy = mirror_coordinate(__dadd_rn(__dmul_rn(oy + 0.5, zoom_y), -0.5), H);
    const double cx = mirror_coordinate(__dadd_rn(__dmul_rn(ox + 0.5, zoom_x), -0.5), W);
    const double fy = floor(cy), fx = floor(cx);
    const double wy0 = 1.0 - (cy - fy), wy1 = 1.0 - wy0;
    const double wx0 = 1.0 - (cx - fx), wx1 = 1.0 - wx0;
    const int y0 = mirror_index((int)fy, H), y1 = mirror_index((int)fy + 1, H);
    const int x0 = mirror_index((int)fx, W), x1 = mirror_index((int)fx + 1, W);
    const T *r0 = in + (int64_t)y0 * W, *r1 = in + (int64_t)y1 * W;
    double t = 0.0;
    t = __dadd_rn(t, __dmul_rn(__dmul_rn(as_unit<T>(r0[x0]), wy0), wx0));
    t = __dadd_rn(t, __dmul_rn(__dmul_rn(as_unit<T>(r0[x1]), wy0), wx1));
    t = __dadd_rn(t, __dmul_rn(__dmul_rn(as_unit<T>(r1[x0]), wy1), wx0));
    t = __dadd_rn(t, __dmul_rn(__dmul_rn(as_unit<T>(r1[x1]), wy1), wx1));
    // numpy.clip keeps NaNs and leaves everything alone when a bound is NaN
    if (t < range[0])
        t = range[0];
    else if (t > range[1])
        t = range[1];
    out[idx] = t;
}

// The filter and the interpolation in one kernel, for images that shrink along both
// axes: a workgroup produces a tile of the output from an input tile staged in LDS
// (mirrored at the image borders while loading, so the passes inside need no border
// logic), filtered along axis 0, then along axis 1, then sampled.  The operations
// and their order are those of the separate passes: results are identical.
struct FusedGeom {
    int H, W, OH, OW;
    double zoom_y, zoom_x;
    int ry, rx;          // filter radii
    int to_y, to_x;      // output tile
    int fr_y, fr_x;      // filtered rows / columns a tile can need at most
};

template <typename T>
__global__ void __launch_bounds__(256)
downsample_fused_kernel(const T *__restrict__ in, FusedGeom g, const double *__restrict__ wy,
                        const double *__restrict__ wx, const double *__restrict__ partial, int npartial,
                        double *__restrict__ out)
{
    extern __shared__ __align__(16) double fused_lds[];
    __shared__ double range[2];
    double *A = fused_lds;                                          // input tile, later the filtered tile
    double *B = A + (size_t)(g.fr_y + 2 * g.ry) * (g.fr_x + 2 * g.rx);   // after the axis-0 pass
    if (threadIdx.x < 2) {
        double v = threadIdx.x == 0 ? INFINITY : -INFINITY;
        for (int i = 0; i < npartial; ++i)
            v = threadIdx.x == 0 ? fmin(v, partial[2 * i]) : fmax(v, partial[2 * i + 1]);
        range[threadIdx.x] = v;
    }
    const int oy0 = blockIdx.y * g.to_y, ox0 = blockIdx.x * g.to_x;
    const int oy1 = min(oy0 + g.to_y, g.OH) - 1, ox1 = min(ox0 + g.to_x, g.OW) - 1;
    // filtered rows / columns this tile samples: floor(first coordinate) .. floor(last) + 1
    const int fy0 = (int)floor(__dadd_rn(__dmul_rn(oy0 + 0.5, g.zoom_y), -0.5));
    const int fx0 = (int)floor(__dadd_rn(__dmul_rn(ox0 + 0.5, g.zoom_x), -0.5));
    const int nfy = (int)floor(__dadd_rn(__dmul_rn(oy1 + 0.5, g.zoom_y), -0.5)) + 2 - fy0;
    const int nfx = (int)floor(__dadd_rn(__dmul_rn(ox1 + 0.5, g.zoom_x), -0.5)) + 2 - fx0;
    const int ih = nfy + 2 * g.ry, iw = nfx + 2 * g.rx;

    for (int i = threadIdx.x; i < ih * iw; i += blockDim.x) {
        const int r = i / iw, c = i - r * iw;
        const int gy = mirror_index(fy0 - g.ry + r, g.H), gx = mirror_index(fx0 - g.rx + c, g.W);
        A[i] = as_unit<T>(in[(int64_t)gy * g.W + gx]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfy * iw; i += blockDim.x) {
        const int r = i / iw, c = i - r * iw;
        const double *col = A + (r + g.ry) * iw + c;
        double acc = __dmul_rn(col[0], wy[g.ry]);
        for (int k = g.ry; k >= 1; --k)
            acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(col[-k * iw], col[k * iw]), wy[g.ry - k]));
        B[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfy * nfx; i += blockDim.x) {
        const int r = i / nfx, c = i - r * nfx;
        const double *row = B + r * iw + c + g.rx;
        double acc = __dmul_rn(row[0], wx[g.rx]);
        for (int k = g.rx; k >= 1; --k)
            acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(row[-k], row[k]), wx[g.rx - k]));
        A[i] = acc;
    }
    __syncthreads();
    const int tw = ox1 - ox0 + 1, tcount = (oy1 - oy0 + 1) * tw;
    for (int i = threadIdx.x; i < tcount; i += blockDim.x) {
        const int oy = oy0 + i / tw, ox = ox0 + i % tw;
        const double cy = __dadd_rn(__dmul_rn(oy + 0.5, g.zoom_y), -0.5);
        const double cx = __dadd_rn(__dmul_rn(ox + 0.5, g.zoom_x), -0.5);
        const double fy = floor(cy), fx = floor(cx);
        const double wy0 = 1.0 - (cy - fy), wy1 = 1.0 - wy0;
        const double wx0 = 1.0 - (cx - fx), wx1 = 1.0 - wx0;
        const double *r0 = A + ((int)fy - fy0) * nfx + ((int)fx - fx0), *r1 = r0 + nfx;
        double t = 0.0;
        t = __dadd_rn(t, __dmul_rn(__dmul_rn(r0[0], wy0), wx0));
        t = __dadd_rn(t, __dmul_rn(__dmul_rn(r0[1], wy0), wx1));
        t = __dadd_rn(t, __dmul_rn(__dmul_rn(r1[0], wy1), wx0));
        t = __dadd_rn(t, __dmul_rn(__dmul_rn(r1[1], wy1), wx1));
        if (t < range[0])
            t = range[0];
        else if (t > range[1])
            t = range[1];
        out[(int64_t)oy * g.OW + ox] = t;
    }
}

constexpr size_t FUSED_LDS_LIMIT = 64 * 1024;

// Output tile for the fused kernel, or false when even a 4 x 4 tile's input does not
// fit (very large reduction factors): the separate passes serve those.
static bool fused_geometry(int H, int W, int OH, int OW, int ry, int rx, FusedGeom &g, size_t &lds)
{
    g.H = H; g.W = W; g.OH = OH; g.OW = OW;
    g.zoom_y = (double)H / OH; g.zoom_x = (double)W / OW;
    g.ry = ry; g.rx = rx;
    g.to_y = 16; g.to_x = 32;
    for (;;) {
        g.fr_y = (int)ceil(g.to_y * g.zoom_y) + 2;
        g.fr_x = (int)ceil(g.to_x * g.zoom_x) + 2;
        lds = sizeof(double) * ((size_t)(g.fr_y + 2 * ry) * (g.fr_x + 2 * rx) + (size_t)g.fr_y * (g.fr_x + 2 * rx));
        if (lds <= FUSED_LDS_LIMIT)
            return true;
        if (g.to_x <= 4 && g.to_y <= 4)
            return false;
        if (g.to_x >= g.to_y)
            g.to_x /= 2;
        else
            g.to_y /= 2;
    }
}

static int gaussian_radius(double sigma)
{
    return (int)(4.0 * sigma + 0.5);
}

template <typename T>
static int downsample_launch(const T *image, int H, int W, int OH, int OW, double *out, void *workspace,
                             hipStream_t st)
{
    const int64_t total = (int64_t)H * W;
    const double fy = (double)H / OH, fx = (double)W / OW;
    const double sy = fmax(0.0, (fy - 1) / 2), sx = fmax(0.0, (fx - 1) / 2);
    const bool anti_alias = OH < H || OW < W;
    Carver ws(workspace);
    double *partial = ws.take<double>(2 * PRE_PARTIALS);
    double *weights = ws.take<double>(2 * (2 * PRE_MAX_RADIUS + 1));
    double *tmp0 = ws.take<double>((size_t)total);
    double *tmp1 = ws.take<double>((size_t)total);

    const unsigned grid_in = (unsigned)((total + 255) / 256);
    const int nblocks = (int)(grid_in < (unsigned)PRE_PARTIALS ? grid_in : (unsigned)PRE_PARTIALS);
    hipLaunchKernelGGL(range_partial_kernel<T>, dim3(nblocks), dim3(256), 0, st, image, total, partial);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(range_final_kernel, dim3(1), dim3(256), 0, st, partial, nblocks);
    SV_LAUNCH_CHECK();
    const int npartial = 1;     // the consumers read the folded pair

    FusedGeom fg;
    size_t fused_lds = 0;
    if (OH < H && OW < W && fused_geometry(H, W, OH, OW, gaussian_radius(sy), gaussian_radius(sx), fg, fused_lds)) {
        // weights of the two axes side by side (an axis with sigma 0 gets the single weight 1)
        double *wy = weights, *wx = weights + (2 * PRE_MAX_RADIUS + 1);
        hipLaunchKernelGGL(gaussian_weights_kernel, dim3(1), dim3(256), 0, st, sy, fg.ry, wy);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL(gaussian_weights_kernel, dim3(1), dim3(256), 0, st, sx, fg.rx, wx);
        SV_LAUNCH_CHECK();
        dim3 grid((unsigned)((OW + fg.to_x - 1) / fg.to_x), (unsigned)((OH + fg.to_y - 1) / fg.to_y));
        hipLaunchKernelGGL(downsample_fused_kernel<T>, grid, dim3(256), fused_lds, st, image, fg, wy, wx, partial,
                           npartial, out);
        SV_LAUNCH_CHECK();
        return SVAMD_OK;
    }

    const double *filtered = nullptr;       // null: the input itself
    // scipy skips an axis whose sigma is below 1e-15
    if (anti_alias && sy > 1e-15) {
        const int r = gaussian_radius(sy);
        hipLaunchKernelGGL(gaussian_weights_kernel, dim3(1), dim3(256), 0, st, sy, r, weights);
        SV_LAUNCH_CHECK();
        hipLaunchKernelGGL((gaussian_axis_kernel<T, 0>), dim3(grid_in), dim3(256), 0, st, image, H, W, weights, r, tmp0);
        SV_LAUNCH_CHECK();
        filtered = tmp0;
    }
    if (anti_alias && sx > 1e-15) {
        const int r = gaussian_radius(sx);
        hipLaunchKernelGGL(gaussian_weights_kernel, dim3(1), dim3(256), 0, st, sx, r, weights);
        SV_LAUNCH_CHECK();
        if (filtered)
            hipLaunchKernelGGL((gaussian_axis_kernel<double, 1>), dim3(grid_in), dim3(256), 0, st, filtered, H, W,
                               weights, r, tmp1);
        else
            hipLaunchKernelGGL((gaussian_axis_kernel<T, 1>), dim3(grid_in), dim3(256), 0, st, image, H, W, weights, r,
                               tmp1);
        SV_LAUNCH_CHECK();
        filtered = tmp1;
    }
    const unsigned grid_out = (unsigned)(((int64_t)OH * OW + 255) / 256);
    if (filtered)
        hipLaunchKernelGGL(zoom_linear_kernel<double>, dim3(grid_out), dim3(256), 0, st, filtered, H, W, OH, OW, fy, fx,
                           partial, npartial, out);
    else
        hipLaunchKernelGGL(zoom_linear_kernel<T>, dim3(grid_out), dim3(256), 0, st, image, H, W, OH, OW, fy, fx,
                           partial, npartial, out);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

}  // namespace svamd

using namespace svamd;

extern "C" int svamd_greyscale(const uint8_t *pixels, int channels, int H, int W, int nan_alpha, uint8_t *grey_u8,
                               double *grey_f64, void *stream)
{
    SV_REQUIRE(pixels && (grey_u8 || grey_f64), SVAMD_ERR_INVALID, "greyscale: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0, SVAMD_ERR_INVALID, "greyscale: non-positive size");
    SV_REQUIRE(channels >= 1 && channels <= 4, SVAMD_ERR_INVALID, "greyscale: 1 to 4 channels per pixel");
    SV_REQUIRE(channels != 4 || (reinterpret_cast<uintptr_t>(pixels) & 3) == 0, SVAMD_ERR_INVALID,
               "greyscale: four-channel pixels must be 4-byte aligned");
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(greyscale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pixels, channels, total, nan_alpha, grey_u8, grey_f64);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" size_t svamd_downsample_workspace_bytes(int H, int W)
{
    if (H <= 0 || W <= 0)
        return 0;
    return align_up(2 * PRE_PARTIALS * sizeof(double), 256) + align_up(2 * (2 * PRE_MAX_RADIUS + 1) * sizeof(double), 256) +
           2 * align_up((size_t)H * W * sizeof(double), 256);
}

extern "C" int svamd_downsample(const void *image, int dtype, int H, int W, int out_h, int out_w, double *out,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(image && out, SVAMD_ERR_INVALID, "downsample: NULL pointer");
    SV_REQUIRE(H > 0 && W > 0 && out_h > 0 && out_w > 0, SVAMD_ERR_INVALID, "downsample: non-positive size");
    SV_REQUIRE(dtype == SVAMD_U8 || dtype == SVAMD_F64, SVAMD_ERR_INVALID,
               "downsample: images are uint8 or float64");
    SV_REQUIRE(gaussian_radius(((double)H / out_h - 1) / 2) <= PRE_MAX_RADIUS &&
                   gaussian_radius(((double)W / out_w - 1) / 2) <= PRE_MAX_RADIUS,
               SVAMD_ERR_LIMIT, "downsample: reduction factor above 128");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_downsample_workspace_bytes(H, W),
               SVAMD_ERR_WORKSPACE, "downsample: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SVAMD_U8)
        return downsample_launch(static_cast<const uint8_t *>(image), H, W, out_h, out_w, out, workspace, st);
    return downsample_launch(static_cast<const double *>(image), H, W, out_h, out_w, out, workspace, st);
}
