'''Spatial cross-correlation on MI355X.

Drop-in for the reference module of the same name: identical function names,
arguments, return types, exceptions and progress prints (reference
correlation.py:4-117); the arithmetic runs in libsvamd's HIP kernels.
'''
import math

import numpy as np

import _native as nv

# The reference prints a progress line every 1000 lags (1D) / every 10 output
# rows (2D) while it grinds through its Python loops.  The same lines are
# printed here (after the kernels have finished) so driver output is unchanged;
# set to False to silence them.
PRINT_PROGRESS = True


def _progress(total, every):
    if PRINT_PROGRESS:
        for i in range(0, total, every):
            print(f'{i} / {total}')


def _xcorr_1d_device(v1, v2, normalised):
    n = len(v1)
    d1 = nv.to_device(np.asarray(v1, dtype=np.float64))
    d2 = nv.to_device(np.asarray(v2, dtype=np.float64))
    torch = nv.torch_cuda()
    out = nv.empty(2 * n - 1, torch.float64)
    nbytes = nv.load_library().svamd_cross_correlate_1d_workspace_bytes(n)
    ws = nv.workspace(nbytes)
    nv.call('svamd_cross_correlate_1d', nv.ptr(d1), nv.ptr(d2), n, int(bool(normalised)),
            nv.ptr(out), nv.ptr(ws), ws.numel(), nv.stream_ptr())
    return out.cpu().numpy()


def cross_correlate_1d_raw(vec_1, vec_2, normalised=False):
    '''Full-lag cross-correlation of two equal-length sequences, returned as a
    Python list (reference correlation.py:4-42).'''
    n = len(vec_1)
    if n != len(vec_2):
        raise Exception(f'Vectors {vec_1} ({n}) and'
            f' {vec_2} ({len(vec_2)}) are of unequal length.')
    out = _xcorr_1d_device(vec_1, vec_2, normalised)
    _progress(2 * n - 1, 1000)
    return out.tolist()


def cross_correlate_1d(vec_1: np.ndarray, vec_2: np.ndarray, normalised=False):
    '''Full-lag (2n-1) mean-removed cross-correlation of two equal-length
    vectors, optionally divided by both standard deviations (reference
    correlation.py:44-75).'''
    n = len(vec_1)
    if n != len(vec_2):
        raise Exception(f'Vectors {vec_1} ({n}) and'
            f' {vec_2} ({len(vec_2)}) are of unequal length.')
    out = _xcorr_1d_device(vec_1, vec_2, normalised)
    _progress(2 * n - 1, 1000)
    return out


def cross_correlate_2d(template: np.ndarray, region: np.ndarray, step_x=1, step_y=1):
    '''Strided sliding normalised cross-correlation of a (possibly NaN-masked)
    template over a region, including the truncated placements at the bottom
    and right edges (reference correlation.py:77-117).'''
    template = np.asarray(template)
    region = np.asarray(region)
    if template.shape[0] > region.shape[0] or template.shape[1] > region.shape[1]:
        raise Exception('Dimensions of template must not exceed those of region.')
    th, tw = template.shape
    rh, rw = region.shape
    output_rows = math.ceil(rh / step_y)
    output_columns = math.ceil(rw / step_x)

    torch = nv.torch_cuda()
    reg = nv.as_image(region)
    # A NaN in the region makes every placement whose snapshot touches it NaN in the
    # reference (the snapshot's mean and deviation are NaN): the kernel sees zeros
    # there and the touched placements are marked afterwards.
    holes = np.isnan(reg) if reg.dtype.kind == 'f' else None
    if holes is not None and holes.any():
        reg = np.where(holes, 0, reg)
    else:
        holes = None
    d_t = nv.to_device(template.astype(np.float64, copy=False))
    d_r = nv.to_device(reg)
    out = nv.empty((output_rows, output_columns), torch.float64)
    nbytes = nv.load_library().svamd_cross_correlate_2d_workspace_bytes(th, tw, rh, rw)
    ws = nv.workspace(nbytes)
    nv.call('svamd_cross_correlate_2d', nv.ptr(d_t), th, tw, nv.ptr(d_r), nv.dtype_code(reg), rh, rw,
            int(step_x), int(step_y), nv.ptr(out), nv.ptr(ws), ws.numel(), nv.stream_ptr())
    result = out.cpu().numpy()
    if holes is not None:
        # NaNs per snapshot from the mask's summed-area table (snapshots are cut off at
        # the region's bottom and right edges)
        table = np.zeros((rh + 1, rw + 1), dtype=np.int64)
        table[1:, 1:] = holes.cumsum(axis=0).cumsum(axis=1)
        y0 = np.arange(output_rows) * step_y
        x0 = np.arange(output_columns) * step_x
        y1, x1 = np.minimum(y0 + th, rh), np.minimum(x0 + tw, rw)
        count = (table[np.ix_(y1, x1)] - table[np.ix_(y0, x1)] - table[np.ix_(y1, x0)] + table[np.ix_(y0, x0)])
        result[count > 0] = np.nan
    _progress(output_rows, 10)
    return result
