'''Globally normalised circular ("spectral") cross-correlation on MI355X.

Drop-in for the reference module of the same name (reference
correlation_spectral.py:1-55).  The reference reaches these surfaces through
FFTs; libsvamd evaluates the same circular correlation with hand-written HIP
kernels (direct evaluation in LDS for scan-sized windows, a tiled sliding dot
product for large shapes).
'''
import numpy as np

import _native as nv

# Minimum template standard deviation below which the surface degenerates to a
# constant (reference correlation_spectral.py:6)
MIN_ST_DEV = 1e-4

# Mean difference under which a homogeneous template counts as matching
# (reference correlation_spectral.py:10)
MIN_MEAN_DIFF = 1e-9

# The float32 FFT route's rounding error grows with sqrt(region size / template
# size); it is taken only when the (cropped) template covers at least 1/FFT_MIN_COVER
# of the region, smaller templates go through the direct sliding kernel.
FFT_MIN_COVER = 4


def cross_correlate_1d_spectral(template: np.ndarray, signal: np.ndarray):
    '''Circular cross-correlation, length len(signal) + len(template) - 1, of the
    normalised template with the raw, left-zero-padded signal (reference
    correlation_spectral.py:12-30).'''
    t = np.ascontiguousarray(template, dtype=np.float64).ravel()
    s = np.ascontiguousarray(signal, dtype=np.float64).ravel()
    nt, ns = t.size, s.size
    torch = nv.torch_cuda()
    d_t, d_s = nv.to_device(t), nv.to_device(s)
    out = nv.empty(ns + nt - 1, torch.float64)
    nbytes = nv.load_library().svamd_cross_correlate_1d_spectral_workspace_bytes(nt, ns)
    ws = nv.workspace(nbytes)
    nv.call('svamd_cross_correlate_1d_spectral', nv.ptr(d_t), nt, nv.ptr(d_s), ns, nv.ptr(out),
            nv.ptr(ws), ws.numel(), nv.stream_ptr())
    return out.cpu().numpy()


def _small_pair_surface(template, region):
    '''One pair through the scan's workgroup-per-pair kernel, float64
    accumulation, full surface returned.'''
    torch = nv.torch_cuda()
    t = np.ascontiguousarray(template, dtype=np.float64)
    r = np.ascontiguousarray(region, dtype=np.float64)
    th, tw = t.shape
    rh, rw = r.shape
    buf = nv.to_device(np.concatenate([t.ravel(), r.ravel()]))
    desc = nv.to_device(np.array([0, tw, th, tw, t.size, rw, rh, rw], dtype=np.int64))
    out_max = nv.empty(1, torch.float64)
    surf = nv.empty((rh, rw), torch.float64)
    off = nv.to_device(np.zeros(1, dtype=np.int64))
    nv.call('svamd_scan_pairs', nv.ptr(buf), nv.ptr(buf), nv.ptr(desc), 1, max(th, tw, rh, rw),
            float(MIN_ST_DEV), float(MIN_MEAN_DIFF), nv.ACC_F64, nv.ptr(out_max), None,
            nv.ptr(surf), nv.ptr(off), nv.stream_ptr())
    return surf.cpu().numpy()


def cross_correlate_2d_spectral_batched(templates, regions, out_dtype=np.float32):
    '''`cross_correlate_2d_spectral` for a stack of pairs at once: templates
    [B, th, tw] and regions [B, H, W] with H and W powers of two (4..4096),
    through the batched LDS FFT kernels.  Returns [B, H, W] of out_dtype.  Inputs
    are used as float32.'''
    torch = nv.torch_cuda()
    lib = nv.load_library()
    t = np.ascontiguousarray(templates, dtype=np.float32)
    r = np.ascontiguousarray(regions, dtype=np.float32)
    batch, th, tw = t.shape
    _, H, W = r.shape
    if r.shape[0] != batch:
        raise Exception('Numbers of templates and regions must match')
    d_t, d_r = nv.to_device(t), nv.to_device(r)
    f64 = np.dtype(out_dtype) == np.float64
    out = nv.empty((batch, H, W), torch.float64 if f64 else torch.float32)
    ws = nv.workspace(lib.svamd_xcorr2d_fft_workspace_bytes(batch, H, W))
    nv.call('svamd_xcorr2d_fft', nv.ptr(d_t), th, tw, nv.ptr(d_r), H, W, batch, float(MIN_ST_DEV),
            float(MIN_MEAN_DIFF), nv.ptr(out), nv.F64 if f64 else nv.F32, nv.ptr(ws), ws.numel(),
            nv.stream_ptr())
    return out.cpu().numpy()


def cross_correlate_2d_spectral(orig_template: np.ndarray, orig_region: np.ndarray):
    '''Normalised circular cross-correlation surface of a template and a region,
    shape = region.shape; the template is zero-padded or cropped (top-left) to
    the region's shape, statistics come from the full arrays, and a template
    whose deviation is below MIN_ST_DEV yields an all-ones / all-zeros surface
    (reference correlation_spectral.py:32-55).'''
    template = np.asarray(orig_template)
    region = np.asarray(orig_region)
    th, tw = template.shape
    rh, rw = region.shape
    if template.size == 0 and region.size > 0:
        # An empty template (a box cut off at the image edge) has nothing to transform:
        # the reference's zero-padded transform of it is all zeros, so the surface is 0
        # wherever the normalised region is finite - and NaN everywhere for a constant
        # region, whose normalisation is 0 / 0 (correlation_spectral.py:37-54).
        flat = float(np.std(region)) == 0.0 or bool(np.isnan(region).any())
        return np.full(region.shape, np.nan if flat else 0.0)
    if max(th, tw, rh, rw) <= nv.PAIR_MAX_SIDE and min(th, tw, rh, rw) > 0:
        return _small_pair_surface(template, region)
    overlap = min(th, rh) * min(tw, rw)
    if overlap * FFT_MIN_COVER >= rh * rw and nv.load_library().svamd_fft_supported(rh, rw):
        # power-of-two region: batched LDS FFT route with a batch of one
        return cross_correlate_2d_spectral_batched(template[None], region[None], np.float64)[0]
    return spectral_surface_on_device(template, region).cpu().numpy()


def spectral_surface_on_device(template, region):
    '''cross_correlate_2d_spectral's surface through the sliding kernel, left on
    the device as a float64 tensor (for callers that keep working on it there).'''
    template = np.asarray(template)
    region = np.asarray(region)
    th, tw = template.shape
    rh, rw = region.shape
    torch = nv.torch_cuda()
    t = nv.as_image(template)
    r = nv.as_image(region)
    d_t, d_r = nv.to_device(t), nv.to_device(r)
    out = nv.empty((rh, rw), torch.float64)
    nbytes = nv.load_library().svamd_cross_correlate_2d_spectral_workspace_bytes(th, tw, rh, rw)
    ws = nv.workspace(nbytes)
    nv.call('svamd_cross_correlate_2d_spectral', nv.ptr(d_t), nv.dtype_code(t), th, tw,
            nv.ptr(d_r), nv.dtype_code(r), rh, rw, float(MIN_ST_DEV), float(MIN_MEAN_DIFF),
            nv.ptr(out), nv.ptr(ws), ws.numel(), nv.stream_ptr())
    return out
