// The inner loop of the stereo scan: for many (template window, search region)
// pairs at once, the globally normalised circular cross-correlation surface of
// the pair and its maximum, followed by the per-window selection rule.
//
// One 256-thread workgroup per pair.  The workgroup reduces the float64 sums of
// both boxes (two passes: mean, then squared deviations), stages the
// mean-removed template and a horizontally wrapped copy of the mean-removed
// region in LDS, and evaluates the surface directly,
//     c[k][l] = sum_{m<mh, n<mw} t'[m][n] * r'[(m + k) mod rh][(n + l) mod rw],
// which is what conj(FFT(t')) * FFT(r') transforms back to.  The multiply-
// accumulate runs in float32 (rows summed in float32, rows added in float64) or
// entirely in float64; statistics, scaling and the max-reduction are float64.
#include "common.h"
#include "kernels.h"

namespace svamd {

struct PairDesc {
    int64_t t_off, t_pitch, t_h, t_w, r_off, r_pitch, r_h, r_w;
};

struct BestNan {
    double best;   // maximum over non-NaN values (-inf if none)
    int arg;       // its first index (INT_MAX if none)
    int has_nan;
};

__device__ __forceinline__ BestNan combine(BestNan a, BestNan b)
{
    BestNan r;
    r.has_nan = a.has_nan | b.has_nan;
    bool take_b = b.best > a.best || (b.best == a.best && b.arg < a.arg);
    r.best = take_b ? b.best : a.best;
    r.arg = take_b ? b.arg : a.arg;
    return r;
}

__device__ __forceinline__ BestNan block_best(BestNan v, double *red_d, int *red_i)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        BestNan n;
        n.best = __shfl_xor(v.best, o, WAVE);
        n.arg = __shfl_xor(v.arg, o, WAVE);
        n.has_nan = __shfl_xor(v.has_nan, o, WAVE);
        v = combine(v, n);
    }
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int nw = blockDim.x / WAVE;
    __syncthreads();
    if (lane == 0) {
        red_d[wid] = v.best;
        red_i[2 * wid] = v.arg;
        red_i[2 * wid + 1] = v.has_nan;
    }
    __syncthreads();
    BestNan r;
    r.best = red_d[0];
    r.arg = red_i[0];
    r.has_nan = red_i[1];
    for (int w = 1; w < nw; ++w) {
        BestNan n;
        n.best = red_d[w];
        n.arg = red_i[2 * w];
        n.has_nan = red_i[2 * w + 1];
        r = combine(r, n);
    }
    return r;
}

template <typename CT>
__global__ void __launch_bounds__(256)
scan_pairs_kernel(const double *__restrict__ tbase, const double *__restrict__ rbase,
                  const PairDesc *__restrict__ desc, int64_t npairs, int max_side, double min_std,
                  double min_mean_diff, double *__restrict__ out_max, int64_t *__restrict__ out_argmax,
                  double *__restrict__ surfaces, const int64_t *__restrict__ surf_off)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double red[17];
    __shared__ double red_d[4];
    __shared__ int red_i[8];   // up to 4 wavefronts per workgroup
    CT *Tb = reinterpret_cast<CT *>(smem_raw);

    for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
        const PairDesc d = desc[p];
        const int th = (int)d.t_h, tw = (int)d.t_w, rh = (int)d.r_h, rw = (int)d.r_w;
        __syncthreads();  // LDS reuse across loop iterations
        if (rh <= 0 || rw <= 0 || th < 0 || tw < 0 || th > max_side || tw > max_side || rh > max_side ||
            rw > max_side) {
            if (threadIdx.x == 0) {
                out_max[p] = NAN;
                if (out_argmax)
                    out_argmax[p] = -1;
            }
            continue;
        }
        const double *T = tbase + d.t_off;
        const double *R = rbase + d.r_off;
        const int tn = th * tw, rn = rh * rw;

        // pass 1: means
        double ts = 0.0, rs = 0.0;
        for (int i = threadIdx.x; i < tn; i += blockDim.x) {
            int y = i / tw, x = i - y * tw;
            ts += T[y * d.t_pitch + x];
        }
        for (int i = threadIdx.x; i < rn; i += blockDim.x) {
            int y = i / rw, x = i - y * rw;
            rs += R[y * d.r_pitch + x];
        }
        const double tmean = block_sum(ts, red) / (double)tn;
        const double rmean = block_sum(rs, red) / (double)rn;
        // pass 2: squared deviations
        double tq = 0.0, rq = 0.0;
        for (int i = threadIdx.x; i < tn; i += blockDim.x) {
            int y = i / tw, x = i - y * tw;
            double v = T[y * d.t_pitch + x] - tmean;
            tq += v * v;
        }
        for (int i = threadIdx.x; i < rn; i += blockDim.x) {
            int y = i / rw, x = i - y * rw;
            double v = R[y * d.r_pitch + x] - rmean;
            rq += v * v;
        }
        const double tstd = sqrt(block_sum(tq, red) / (double)tn);
        const double rstd = sqrt(block_sum(rq, red) / (double)rn);
        double *surf = surfaces ? surfaces + surf_off[p] : nullptr;

        if (tstd < min_std) {  // homogeneous template: constant surface
            const double fill = fabs(tmean - rmean) < min_mean_diff ? 1.0 : 0.0;
            if (surf)
                for (int i = threadIdx.x; i < rn; i += blockDim.x)
                    surf[i] = fill;
            if (threadIdx.x == 0) {
                out_max[p] = fill;
                if (out_argmax)
                    out_argmax[p] = 0;
            }
            continue;
        }

        // LDS images: template rows padded with zeros to a multiple of 4 columns,
        // region rows wrapped horizontally and long enough for every 4-wide read
        const int mh = min(th, rh), mw = min(tw, rw);
        const int mwp = (mw + 3) & ~3;
        const int rxp = (rw + mwp + 4 + 3) & ~3;
        CT *Rx = Tb + mh * mwp;
        for (int i = threadIdx.x; i < mh * mwp; i += blockDim.x) {
            int y = i / mwp, x = i - y * mwp;
            Tb[i] = x < mw ? (CT)(T[y * d.t_pitch + x] - tmean) : CT(0);
        }
        for (int i = threadIdx.x; i < rh * rxp; i += blockDim.x) {
            int y = i / rxp, x = i - y * rxp;
            CT v = CT(0);
            if (x < rw + mw - 1) {
                int xs = x >= rw ? x - rw : x;
                v = (CT)(R[y * d.r_pitch + xs] - rmean);
            }
            Rx[i] = v;
        }
        __syncthreads();

        // An empty template transforms to an all-zero spectrum: the surface is 0
        // wherever the normalised region is finite.
        const double scale = tn == 0 ? ((rstd > 0.0) ? 0.0 : NAN) : 1.0 / (tstd * (double)tn) / rstd;

        // One task = 4 horizontally adjacent surface elements (k, l0 .. l0+3): the
        // region row is walked with a sliding 8-value register window, so every
        // 4-wide LDS read of template and region feeds 16 FMAs.
        BestNan mine;
        mine.best = -INFINITY;
        mine.arg = 0x7fffffff;
        mine.has_nan = 0;
        const int lgroups = (rw + 3) / 4;
        for (int task = threadIdx.x; task < rh * lgroups; task += blockDim.x) {
            const int k = task / lgroups, l0 = (task - k * lgroups) * 4;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int m = 0; m < mh; ++m) {
                int rr = m + k;
                if (rr >= rh)
                    rr -= rh;
                const CT *trow = Tb + m * mwp;
                const CT *rrow = Rx + rr * rxp + l0;
                CT part[4] = {CT(0), CT(0), CT(0), CT(0)};
                CT win[8];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    win[j] = rrow[j];
                for (int n0 = 0; n0 < mwp; n0 += 4) {
                    CT tv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tv[j] = trow[n0 + j];
                        win[4 + j] = rrow[n0 + 4 + j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            part[j] = fma(tv[i], win[i + j], part[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        win[j] = win[4 + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] += (double)part[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (l0 + j >= rw)
                    continue;
                const int idx = k * rw + l0 + j;
                const double v = tn == 0 ? scale : acc[j] * scale;
                if (surf)
                    surf[idx] = v;
                if (isnan(v)) {
                    mine.has_nan = 1;
                } else if (v > mine.best) {      // ascending idx within a thread: first maximum kept
                    mine.best = v;
                    mine.arg = idx;
                }
            }
        }
        const BestNan all = block_best(mine, red_d, red_i);
        if (threadIdx.x == 0) {
            out_max[p] = all.has_nan ? NAN : all.best;
            if (out_argmax)
                out_argmax[p] = all.arg == 0x7fffffff ? -1 : all.arg;
        }
    }
}

// image_scan's sequential selection rule, one thread per window.
__global__ void __launch_bounds__(256)
scan_fold_kernel(const double *__restrict__ corr_max, const uint8_t *__restrict__ is_centre,
                 const uint8_t *__restrict__ skip, const int64_t *__restrict__ win_off, int64_t nwin,
                 double threshold, double centre_tol, int64_t *__restrict__ best_pair,
                 double *__restrict__ best_val, double *__restrict__ min_margin)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= nwin)
        return;
    double best = 0.0, margin = INFINITY;
    int64_t arg = -1;
    for (int64_t p = win_off[w]; p < win_off[w + 1]; ++p) {
        if (skip && skip[p])
            continue;
        const double c = corr_max[p];
        if (isnan(c))
            continue;
        const bool centre = is_centre[p] != 0;
        const double bar = centre ? best : best + centre_tol;
        margin = fmin(margin, fmin(fabs(c - threshold), fabs(c - bar)));
        const bool upd = c > threshold && (centre ? c >= bar : c > bar);
        if (upd) {
            best = c;
            arg = p;
        }
    }
    best_pair[w] = arg;
    best_val[w] = best;
    if (min_margin)
        min_margin[w] = margin;
}

}  // namespace svamd

using namespace svamd;

extern "C" int svamd_scan_pairs(const double *tbase, const double *rbase, const int64_t *desc,
                                int64_t npairs, int max_side, double min_std, double min_mean_diff,
                                int acc_precision, double *out_max, int64_t *out_argmax, double *surfaces,
                                const int64_t *surf_off, void *stream)
{
    SV_REQUIRE(npairs >= 0, SVAMD_ERR_INVALID, "scan_pairs: negative pair count");
    if (npairs == 0)
        return SVAMD_OK;
    SV_REQUIRE(tbase && rbase && desc && out_max, SVAMD_ERR_INVALID, "scan_pairs: NULL pointer");
    SV_REQUIRE(!surfaces || surf_off, SVAMD_ERR_INVALID, "scan_pairs: surfaces given without offsets");
    SV_REQUIRE(max_side > 0 && max_side <= SVAMD_PAIR_MAX_SIDE, SVAMD_ERR_LIMIT,
               "scan_pairs: max_side %d outside 1..%d", max_side, SVAMD_PAIR_MAX_SIDE);
    SV_REQUIRE(acc_precision == SVAMD_ACC_F32 || acc_precision == SVAMD_ACC_F64, SVAMD_ERR_INVALID,
               "scan_pairs: unknown accumulation precision %d", acc_precision);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t side4 = ((size_t)max_side + 3) & ~(size_t)3;
    const size_t elems = (size_t)max_side * side4 + (size_t)max_side * (((size_t)max_side + side4 + 4 + 3) & ~(size_t)3);
    // one task per 4 surface elements of a row: small windows get small workgroups
    const int tasks = max_side * ((max_side + 3) / 4);
    const int threads = tasks <= 64 ? 64 : tasks <= 128 ? 128 : 256;
    const unsigned grid = (unsigned)(npairs < (1 << 20) ? npairs : (1 << 20));
    const PairDesc *pd = reinterpret_cast<const PairDesc *>(desc);
    if (acc_precision == SVAMD_ACC_F32) {
        const size_t lds = elems * sizeof(float);
        hipLaunchKernelGGL((scan_pairs_kernel<float>), dim3(grid), dim3(threads), lds, st, tbase, rbase, pd,
                           npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces,
                           surf_off);
    } else {
        const size_t lds = elems * sizeof(double);
        if (lds > 64 * 1024) {
            static bool raised = false;  // attribute is per function, sticky
            if (!raised) {
                SV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_pairs_kernel<double>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                raised = true;
            }
        }
        hipLaunchKernelGGL((scan_pairs_kernel<double>), dim3(grid), dim3(threads), lds, st, tbase, rbase, pd,
                           npairs, max_side, min_std, min_mean_diff, out_max, out_argmax, surfaces,
                           surf_off);
    }
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}

extern "C" int svamd_scan_fold(const double *corr_max, const uint8_t *is_centre, const uint8_t *skip,
                               const int64_t *win_off, int64_t nwin, double threshold, double centre_tol,
                               int64_t *best_pair, double *best_val, double *min_margin, void *stream)
{
    SV_REQUIRE(nwin >= 0, SVAMD_ERR_INVALID, "scan_fold: negative window count");
    if (nwin == 0)
        return SVAMD_OK;
    SV_REQUIRE(corr_max && is_centre && win_off && best_pair && best_val, SVAMD_ERR_INVALID,
               "scan_fold: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(scan_fold_kernel, dim3((unsigned)((nwin + 255) / 256)), dim3(256), 0, st, corr_max,
                       is_centre, skip, win_off, nwin, threshold, centre_tol, best_pair, best_val,
                       min_margin);
    SV_LAUNCH_CHECK();
    return SVAMD_OK;
}
