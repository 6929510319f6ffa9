// Sliding-dot-product cross-correlation: the spatial 2D NCC (a strided, NaN
// masked, edge-truncated template scan), the large-shape circular 2D NCC and the
// two 1D correlators all reduce to
//     raw[i][j] = sum_{m,n} t'[m][n] * ext[i*sy + m][j*sx + n]
// over a mean-removed float32 template t' and a float32 image `ext` that has been
// extended (zero padded or wrapped circularly) so that no index leaves it.  The
// normalisation is an O(1)-per-output epilogue in float64 fed by summed-area
// tables (spatial) or global statistics (circular / 1D).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svamd {

// ---- statistics ---------------------------------------------------------------
// One block; two passes (mean, then squared deviations), all in float64, the
// same two-pass definition numpy's mean/std and nanmean/nanstd use.
__global__ void __launch_bounds__(1024)
stats_kernel(const void *__restrict__ src, int dtype, int h, int w, int64_t pitch, int ignore_nan,
             Stats *__restrict__ out)
{
    __shared__ double red[17];
    const int64_t total = (int64_t)h * w;
    double s = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
        int64_t y = i / w, x = i - y * w;
        double v = load_as_double(src, dtype, y * pitch + x);
        if (!ignore_nan || !isnan(v)) {
            s += v;
            c += 1.0;
        }
    }
    s = block_sum(s, red);
    c = block_sum(c, red);
    const double mean = s / c;
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
        int64_t y = i / w, x = i - y * w;
        double v = load_as_double(src, dtype, y * pitch + x);
        if (!ignore_nan || !isnan(v)) {
            double d = v - mean;
            q += d * d;
        }
    }
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        out->mean = mean;
        out->std = sqrt(q / c);
        out->count = c;
        out->pad = 0.0;
    }
}

int stats_launch(const void *src, int dtype, int h, int w, int64_t pitch, bool ignore_nan, Stats *out,
                 hipStream_t stream)
{
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, stream, src, dtype, h, w, pitch,
                       ignore_nan ? 1 : 0, out);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// The constant subtracted from region pixels before they are rounded to
// float32: the region mean, rounded to an integer for uint8 data so that the
// centred pixels stay exactly representable.
__device__ __forceinline__ double centre_value(const Stats *st, int dtype)
{
    return dtype == SVAMD_U8 ? rint(st->mean) : st->mean;
}

// ---- operand preparation ----------------------------------------------------------
// Mean-removed template in float32 (masked pixels -> 0) and, optionally, the
// float64 summed-area table of the same values ((th+1) x (tw+1)), which gives the
// sum of t' over any top-left crop.
__global__ void __launch_bounds__(256)
prep_template_kernel(const void *__restrict__ t, int dtype, int th, int tw, int64_t pitch, int oh, int ow,
                     const Stats *__restrict__ st, float *__restrict__ tp)
{
    const double mean = st->mean;
    const int64_t total = (int64_t)oh * ow;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = i / ow, x = i - y * ow;
        double v = load_as_double(t, dtype, y * pitch + x);
        tp[i] = isnan(v) ? 0.0f : (float)(v - mean);
    }
}

__global__ void __launch_bounds__(256)
template_sat_kernel(const float *__restrict__ tp, int th, int tw, double *__restrict__ tsat)
{
    // small table: row prefixes by one thread per row, then column prefixes by
    // one thread per column
    const int tpw = tw + 1;
    for (int y = threadIdx.x; y <= th; y += blockDim.x) {
        double run = 0.0;
        tsat[(int64_t)y * tpw] = 0.0;
        for (int x = 0; x < tw; ++x) {
            run += y == 0 ? 0.0 : (double)tp[(int64_t)(y - 1) * tw + x];
            tsat[(int64_t)y * tpw + x + 1] = run;
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x <= tw; x += blockDim.x) {
        double run = 0.0;
        for (int y = 0; y <= th; ++y) {
            run += tsat[(int64_t)y * tpw + x];
            tsat[(int64_t)y * tpw + x] = run;
        }
    }
}

// Extended float32 image.  mode 0: region - c inside, 0 outside (zero padding of
// the centred data).  mode 1: circular wrap of region - c.
__global__ void __launch_bounds__(256)
extend_kernel(const void *__restrict__ src, int dtype, int rh, int rw, int64_t pitch, int mode,
              int exact_mean, const Stats *__restrict__ st, int eh, int ew, int64_t epitch,
              float *__restrict__ ext)
{
    const double c = exact_mean ? st->mean : centre_value(st, dtype);
    const int64_t total = (int64_t)eh * epitch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = i / epitch, x = i - y * epitch;
        float v = 0.0f;
        if (x < ew) {
            if (mode == 0) {
                if (y < rh && x < rw)
                    v = (float)(load_as_double(src, dtype, y * pitch + x) - c);
            } else {
                v = (float)(load_as_double(src, dtype, (y % rh) * pitch + (x % rw)) - c);
            }
        }
        ext[i] = v;
    }
}

// 1D: ext[k] = (v2 zero-padded by n-1 on both sides)[k] - mean(v2)
__global__ void __launch_bounds__(256)
extend_1d_full_kernel(const double *__restrict__ v2, int64_t n, const Stats *__restrict__ st,
                      float *__restrict__ ext)
{
    const double m2 = st->mean;
    const int64_t total = 3 * n - 2;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = k - (n - 1);
        double v = (s >= 0 && s < n) ? v2[s] : 0.0;
        ext[k] = (float)(v - m2);
    }
}

// 1D circular: ext[k] = padded[k mod N], padded = zeros(nt-1) ++ signal (raw)
__global__ void __launch_bounds__(256)
extend_1d_circular_kernel(const double *__restrict__ sig, int64_t nt, int64_t ns, float *__restrict__ ext)
{
    const int64_t N = ns + nt - 1, total = N + nt - 1;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        int64_t idx = k >= N ? k - N : k;
        ext[k] = idx >= nt - 1 ? (float)sig[idx - (nt - 1)] : 0.0f;
    }
}

// ---- the sliding dot product --------------------------------------------------------
// One thread per output; a block covers 64 x 4 outputs so that the lanes of a
// wavefront read consecutive image columns (coalesced when sx == 1).  Template
// values are wave-uniform and come in through the scalar data path.  Each run of
// up to RUN products is accumulated in float32 and the runs are added in
// float64, which keeps the rounding error of long sums near 1e-7 relative.
constexpr int SLIDE_BX = 64, SLIDE_BY = 4, SLIDE_RUN = 128;

__global__ void __launch_bounds__(SLIDE_BX *SLIDE_BY)
slide_kernel(const float *__restrict__ t, int th, int tw, const float *__restrict__ img, int64_t pitch,
             int oh, int ow, int sx, int sy, double *__restrict__ out)
{
    const int j = blockIdx.x * SLIDE_BX + threadIdx.x;
    const int i = blockIdx.y * SLIDE_BY + threadIdx.y;
    if (i >= oh || j >= ow)
        return;
    const float *base = img + (int64_t)i * sy * pitch + (int64_t)j * sx;
    double acc = 0.0;
    for (int m = 0; m < th; ++m) {
        const float *trow = t + (int64_t)m * tw;
        const float *irow = base + (int64_t)m * pitch;
        for (int n0 = 0; n0 < tw; n0 += SLIDE_RUN) {
            const int n1 = min(n0 + SLIDE_RUN, tw);
            float part = 0.0f;
            for (int n = n0; n < n1; ++n)
                part = fmaf(trow[n], irow[n], part);
            acc += (double)part;
        }
    }
    out[(int64_t)i * ow + j] = acc;
}

int slide_launch(const float *t, int th, int tw, const float *img, int64_t img_pitch, int oh, int ow,
                 int sx, int sy, double *out, hipStream_t stream)
{
    dim3 grid((ow + SLIDE_BX - 1) / SLIDE_BX, (oh + SLIDE_BY - 1) / SLIDE_BY);
    hipLaunchKernelGGL(slide_kernel, grid, dim3(SLIDE_BX, SLIDE_BY), 0, stream, t, th, tw, img, img_pitch,
                       oh, ow, sx, sy, out);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ---- epilogues ------------------------------------------------------------------
template <typename TSat>
__device__ __forceinline__ TSat rect_sum(const TSat *__restrict__ s, int64_t tp, int y0, int x0, int y1,
                                         int x1)
{
    return s[(int64_t)y1 * tp + x1] - s[(int64_t)y0 * tp + x1] - s[(int64_t)y1 * tp + x0] +
           s[(int64_t)y0 * tp + x0];
}

// Spatial NCC normalisation.  raw = sum t' * (R - c) over the (cropped) overlap;
// the covariance term of the reference is sum t' * (R - mean(snapshot)) =
// raw + (c - mean(snapshot)) * sum_crop(t').
template <typename TSat>
__global__ void __launch_bounds__(256)
spatial_epilogue_kernel(double *__restrict__ out, int orows, int ocols, int sx, int sy, int th, int tw,
                        int rh, int rw, int region_dtype, const Stats *__restrict__ tst,
                        const Stats *__restrict__ rst, const double *__restrict__ tsat,
                        const TSat *__restrict__ s1, const TSat *__restrict__ s2)
{
    const int64_t total = (int64_t)orows * ocols;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    const int i = (int)(idx / ocols), j = (int)(idx - (int64_t)i * ocols);
    const int y0 = i * sy, x0 = j * sx;
    const int h = min(th, rh - y0), w = min(tw, rw - x0);
    const double n = (double)h * (double)w;
    const int64_t tp = (int64_t)rw + 1;
    const TSat a1 = rect_sum(s1, tp, y0, x0, y0 + h, x0 + w);
    const TSat a2 = rect_sum(s2, tp, y0, x0, y0 + h, x0 + w);
    const double rmean = (double)a1 / n;
    double rstd;
    bool flat;
    if constexpr (std::is_integral<TSat>::value) {
        // exact integer test for a constant snapshot
        const int64_t var_n2 = (int64_t)(h * w) * (int64_t)a2 - (int64_t)a1 * (int64_t)a1;
        flat = var_n2 == 0;
        rstd = sqrt((double)var_n2) / n;
    } else {
        const double m2 = (double)a2 / n;
        const double var = m2 - rmean * rmean;
        // float64 tables cannot resolve a relative variance below ~1e-12
        flat = !(var > 1e-12 * m2);
        rstd = sqrt(var);
    }
    const double tstd = tst->std;
    const double c = centre_value(rst, region_dtype);
    const double tcrop = tsat[(int64_t)h * (tw + 1) + w];
    double v = (out[idx] + (c - rmean) * tcrop) / ((double)th * (double)tw);
    if (tstd == 0.0 || flat)
        v = NAN;
    else
        v /= (tstd * rstd);
    out[idx] = v;
}

// Circular NCC normalisation, in place: either the low-deviation branch's
// constant surface or raw / (std T * T.size * std R).
__global__ void __launch_bounds__(256)
circular_epilogue_kernel(double *__restrict__ out, int64_t total, double tsize, double min_std,
                         double min_mean_diff, const Stats *__restrict__ tst, const Stats *__restrict__ rst)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    if (tst->std < min_std) {
        out[idx] = fabs(tst->mean - rst->mean) < min_mean_diff ? 1.0 : 0.0;
        return;
    }
    out[idx] = out[idx] / (tst->std * tsize) / rst->std;
}

__global__ void __launch_bounds__(256)
scale_1d_kernel(double *__restrict__ out, int64_t total, double divisor, int use_std1, int use_std2,
                const Stats *__restrict__ s1, const Stats *__restrict__ s2)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= total)
        return;
    double v = out[idx] / divisor;
    if (use_std1 && use_std2)
        v /= (s1->std * s2->std);
    else if (use_std1)
        v /= s1->std;
    out[idx] = v;
}

inline int grid_for(int64_t total, int block = 256, int cap = 65536)
{
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace svamd

using namespace svamd;

// ================================ a1 ==========================================
namespace {
struct A1Layout {
    int eh, ew;
    int64_t epitch;
    int orows, ocols;
    size_t n_tsat, n_sat, n_ext, n_tp;
};
A1Layout a1_layout(int th, int tw, int rh, int rw)
{
    A1Layout L;
    L.eh = rh + th - 1;
    L.ew = rw + tw - 1;
    L.epitch = (int64_t)align_up((size_t)L.ew, 4);
    L.n_tp = (size_t)th * tw;
    L.n_tsat = (size_t)(th + 1) * (tw + 1);
    L.n_sat = (size_t)(rh + 1) * (rw + 1);
    L.n_ext = (size_t)L.eh * L.epitch;
    L.orows = L.ocols = 0;
    return L;
}
}  // namespace

extern "C" size_t svamd_cross_correlate_2d_workspace_bytes(int th, int tw, int rh, int rw)
{
    if (th <= 0 || tw <= 0 || rh <= 0 || rw <= 0)
        return 0;
    A1Layout L = a1_layout(th, tw, rh, rw);
    return 2 * 256 + align_up(L.n_tp * 4, 256) + align_up(L.n_tsat * 8, 256) +
           2 * align_up(L.n_sat * 8, 256) + align_up(L.n_ext * 4, 256);
}

extern "C" int svamd_cross_correlate_2d(const double *tmpl, int th, int tw, const void *region,
                                        int region_dtype, int rh, int rw, int step_x, int step_y,
                                        double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && region && out, SVAMD_ERR_INVALID, "cross_correlate_2d: NULL pointer");
    SV_REQUIRE(th > 0 && tw > 0 && rh > 0 && rw > 0 && step_x > 0 && step_y > 0, SVAMD_ERR_INVALID,
               "cross_correlate_2d: non-positive size or step");
    SV_REQUIRE(region_dtype >= SVAMD_U8 && region_dtype <= SVAMD_F64, SVAMD_ERR_INVALID,
               "cross_correlate_2d: unsupported region dtype %d", region_dtype);
    SV_REQUIRE(th <= rh && tw <= rw, SVAMD_ERR_SHAPE,
               "Dimensions of template must not exceed those of region.");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_2d_workspace_bytes(th, tw, rh, rw),
               SVAMD_ERR_WORKSPACE, "cross_correlate_2d: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    A1Layout L = a1_layout(th, tw, rh, rw);
    const int orows = (rh + step_y - 1) / step_y, ocols = (rw + step_x - 1) / step_x;

    Carver ws(workspace);
    Stats *tst = ws.take<Stats>(1);
    Stats *rst = ws.take<Stats>(1);
    float *tp = ws.take<float>(L.n_tp);
    double *tsat = ws.take<double>(L.n_tsat);
    void *s1 = ws.take<double>(L.n_sat);  // int64 or double: both 8 bytes
    void *s2 = ws.take<double>(L.n_sat);
    float *ext = ws.take<float>(L.n_ext);

    int rc;
    if ((rc = stats_launch(tmpl, SVAMD_F64, th, tw, tw, true, tst, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(region, region_dtype, rh, rw, rw, false, rst, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for((int64_t)L.n_tp)), dim3(256), 0, st, tmpl,
                       SVAMD_F64, th, tw, (int64_t)tw, th, tw, tst, tp);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(template_sat_kernel, dim3(1), dim3(256), 0, st, tp, th, tw, tsat);
    SV_LAUNCH_CHECK();
    rc = svamd_sat2d(region, region_dtype, rh, rw, rw, 0.0, s1, s2, stream);
    if (rc != SVAMD_OK) return rc;
    hipLaunchKernelGGL(extend_kernel, dim3(grid_for((int64_t)L.n_ext)), dim3(256), 0, st, region,
                       region_dtype, rh, rw, (int64_t)rw, 0, 0, rst, L.eh, L.ew, L.epitch, ext);
    SV_LAUNCH_CHECK();
    if ((rc = slide_launch(tp, th, tw, ext, L.epitch, orows, ocols, step_x, step_y, out, st)) != SVAMD_OK)
        return rc;
    const int64_t total = (int64_t)orows * ocols;
    dim3 eg((unsigned)((total + 255) / 256));
    if (region_dtype == SVAMD_U8)
        hipLaunchKernelGGL((spatial_epilogue_kernel<int64_t>), eg, dim3(256), 0, st, out, orows, ocols,
                           step_x, step_y, th, tw, rh, rw, region_dtype, tst, rst, tsat,
                           static_cast<const int64_t *>(s1), static_cast<const int64_t *>(s2));
    else
        hipLaunchKernelGGL((spatial_epilogue_kernel<double>), eg, dim3(256), 0, st, out, orows, ocols,
                           step_x, step_y, th, tw, rh, rw, region_dtype, tst, rst, tsat,
                           static_cast<const double *>(s1), static_cast<const double *>(s2));
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ================================ a2 ==========================================
extern "C" size_t svamd_cross_correlate_2d_spectral_workspace_bytes(int th, int tw, int rh, int rw)
{
    if (th <= 0 || tw <= 0 || rh <= 0 || rw <= 0)
        return 0;
    const int mh = th < rh ? th : rh, mw = tw < rw ? tw : rw;
    const size_t epitch = align_up((size_t)(rw + mw - 1), 4);
    return 2 * 256 + align_up((size_t)mh * mw * 4, 256) + align_up((size_t)(rh + mh - 1) * epitch * 4, 256);
}

extern "C" int svamd_cross_correlate_2d_spectral(const void *tmpl, int tmpl_dtype, int th, int tw,
                                                 const void *region, int region_dtype, int rh, int rw,
                                                 double min_std, double min_mean_diff, double *out,
                                                 void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && region && out, SVAMD_ERR_INVALID, "cross_correlate_2d_spectral: NULL pointer");
    SV_REQUIRE(th > 0 && tw > 0 && rh > 0 && rw > 0, SVAMD_ERR_INVALID,
               "cross_correlate_2d_spectral: non-positive size");
    SV_REQUIRE(tmpl_dtype >= SVAMD_U8 && tmpl_dtype <= SVAMD_F64 && region_dtype >= SVAMD_U8 &&
                   region_dtype <= SVAMD_F64,
               SVAMD_ERR_INVALID, "cross_correlate_2d_spectral: unsupported dtype");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_2d_spectral_workspace_bytes(th, tw, rh, rw),
               SVAMD_ERR_WORKSPACE,
               "cross_correlate_2d_spectral: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mh = th < rh ? th : rh, mw = tw < rw ? tw : rw;
    const int eh = rh + mh - 1, ew = rw + mw - 1;
    const int64_t epitch = (int64_t)align_up((size_t)ew, 4);

    Carver ws(workspace);
    Stats *tst = ws.take<Stats>(1);
    Stats *rst = ws.take<Stats>(1);
    float *tp = ws.take<float>((size_t)mh * mw);
    float *ext = ws.take<float>((size_t)eh * epitch);

    int rc;
    if ((rc = stats_launch(tmpl, tmpl_dtype, th, tw, tw, false, tst, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(region, region_dtype, rh, rw, rw, false, rst, st)) != SVAMD_OK) return rc;
    // template minus its mean, cropped top-left to the region's shape
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for((int64_t)mh * mw)), dim3(256), 0, st, tmpl,
                       tmpl_dtype, th, tw, (int64_t)tw, mh, mw, tst, tp);
    SV_LAUNCH_CHECK();
    // region minus its exact mean, wrapped circularly
    hipLaunchKernelGGL(extend_kernel, dim3(grid_for((int64_t)eh * epitch)), dim3(256), 0, st, region,
                       region_dtype, rh, rw, (int64_t)rw, 1, 1, rst, eh, ew, epitch, ext);
    SV_LAUNCH_CHECK();
    if ((rc = slide_launch(tp, mh, mw, ext, epitch, rh, rw, 1, 1, out, st)) != SVAMD_OK) return rc;
    const int64_t total = (int64_t)rh * rw;
    hipLaunchKernelGGL(circular_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       out, total, (double)th * (double)tw, min_std, min_mean_diff, tst, rst);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ================================ a3 ==========================================
extern "C" size_t svamd_cross_correlate_1d_workspace_bytes(int64_t n)
{
    if (n <= 0)
        return 0;
    return 2 * 256 + align_up((size_t)n * 4, 256) + align_up((size_t)(3 * n - 2) * 4, 256);
}

extern "C" int svamd_cross_correlate_1d(const double *v1, const double *v2, int64_t n, int normalised,
                                        double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(v1 && v2 && out, SVAMD_ERR_INVALID, "cross_correlate_1d: NULL pointer");
    SV_REQUIRE(n > 0 && n < (1LL << 30), SVAMD_ERR_INVALID, "cross_correlate_1d: bad length %lld",
               (long long)n);
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_1d_workspace_bytes(n),
               SVAMD_ERR_WORKSPACE, "cross_correlate_1d: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver ws(workspace);
    Stats *s1 = ws.take<Stats>(1);
    Stats *s2 = ws.take<Stats>(1);
    float *tp = ws.take<float>((size_t)n);
    float *ext = ws.take<float>((size_t)(3 * n - 2));
    int rc;
    if ((rc = stats_launch(v1, SVAMD_F64, 1, (int)n, n, false, s1, st)) != SVAMD_OK) return rc;
    if ((rc = stats_launch(v2, SVAMD_F64, 1, (int)n, n, false, s2, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for(n)), dim3(256), 0, st, v1, SVAMD_F64, 1, (int)n,
                       n, 1, (int)n, s1, tp);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(extend_1d_full_kernel, dim3(grid_for(3 * n - 2)), dim3(256), 0, st, v2, n, s2, ext);
    SV_LAUNCH_CHECK();
    const int64_t total = 2 * n - 1;
    if ((rc = slide_launch(tp, 1, (int)n, ext, 3 * n - 2, 1, (int)total, 1, 1, out, st)) != SVAMD_OK)
        return rc;
    hipLaunchKernelGGL(scale_1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total,
                       (double)n, normalised ? 1 : 0, normalised ? 1 : 0, s1, s2);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

// ================================ a5 ==========================================
extern "C" size_t svamd_cross_correlate_1d_spectral_workspace_bytes(int64_t nt, int64_t ns)
{
    if (nt <= 0 || ns <= 0)
        return 0;
    return 256 + align_up((size_t)nt * 4, 256) + align_up((size_t)(ns + 2 * nt - 2) * 4, 256);
}

extern "C" int svamd_cross_correlate_1d_spectral(const double *tmpl, int64_t nt, const double *signal,
                                                 int64_t ns, double *out, void *workspace,
                                                 size_t workspace_bytes, void *stream)
{
    SV_REQUIRE(tmpl && signal && out, SVAMD_ERR_INVALID, "cross_correlate_1d_spectral: NULL pointer");
    SV_REQUIRE(nt > 0 && ns > 0 && nt + ns < (1LL << 30), SVAMD_ERR_INVALID,
               "cross_correlate_1d_spectral: bad lengths");
    SV_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                   workspace_bytes >= svamd_cross_correlate_1d_spectral_workspace_bytes(nt, ns),
               SVAMD_ERR_WORKSPACE,
               "cross_correlate_1d_spectral: workspace missing, misaligned or too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver ws(workspace);
    Stats *s1 = ws.take<Stats>(1);
    float *tp = ws.take<float>((size_t)nt);
    float *ext = ws.take<float>((size_t)(ns + 2 * nt - 2));
    const int64_t N = ns + nt - 1;
    int rc;
    if ((rc = stats_launch(tmpl, SVAMD_F64, 1, (int)nt, nt, false, s1, st)) != SVAMD_OK) return rc;
    hipLaunchKernelGGL(prep_template_kernel, dim3(grid_for(nt)), dim3(256), 0, st, tmpl, SVAMD_F64, 1,
                       (int)nt, nt, 1, (int)nt, s1, tp);
    SV_LAUNCH_CHECK();
    hipLaunchKernelGGL(extend_1d_circular_kernel, dim3(grid_for(N + nt - 1)), dim3(256), 0, st, signal, nt,
                       ns, ext);
    SV_LAUNCH_CHECK();
    if ((rc = slide_launch(tp, 1, (int)nt, ext, N + nt - 1, 1, (int)N, 1, 1, out, st)) != SVAMD_OK)
        return rc;
    hipLaunchKernelGGL(scale_1d_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, out, N,
                       (double)nt, 1, 0, s1, s1);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
