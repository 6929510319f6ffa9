'''CPU oracle for the normalised cross-correlation hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and the
benchmark's CPU-baseline leg, never by the product modules.  It restates the
reference's algorithms in float64 on the CPU: the inner loops live in
ncc_oracle.c (built by oracle/Makefile into oracle/_build/), the window/region
enumeration of the stereo scan is restated here in Python.

Pinned against the reference's own outputs: tests/test_oracle_golden.py checks
every function below against the golden vectors in tests/golden/ (produced by
tests/golden/make_golden.py from the reference itself).
'''
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libncc_oracle.so')

MIN_ST_DEV = 1e-4        # correlation_spectral.py:6
MIN_MEAN_DIFF = 1e-9     # correlation_spectral.py:10
CENTRE_TOL = 0.01        # sv_image_comparison.py:11

_c_double_p = ctypes.POINTER(ctypes.c_double)
_lib = None


def build(force=False):
    '''Compiles ncc_oracle.c with gcc (no-op when the library is up to date).'''
    src = os.path.join(_HERE, 'ncc_oracle.c')
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(['make', '-s', '-C', _HERE, '-B'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_c_double_p)


# -- a1: correlation.py:77-117 ------------------------------------------------
def cross_correlate_2d(template, region, step_x=1, step_y=1):
    t = _f64(template)
    r = _f64(region)
    if t.shape[0] > r.shape[0] or t.shape[1] > r.shape[1]:
        raise Exception('Dimensions of template must not exceed those of region.')
    rows = math.ceil(r.shape[0] / step_y)
    cols = math.ceil(r.shape[1] / step_x)
    out = np.empty((rows, cols))
    rc = lib().orc_ncc2d_spatial(_p(t), t.shape[0], t.shape[1], _p(r), r.shape[0], r.shape[1],
                                 int(step_x), int(step_y), _p(out))
    assert rc == 0
    return out


# -- a2: correlation_spectral.py:32-55 ----------------------------------------
def cross_correlate_2d_spectral(template, region, want='surface'):
    '''Direct evaluation of the circular correlation the reference computes
    with FFTs.  want: 'surface' | 'max' | 'both'.'''
    t = _f64(template)
    r = _f64(region)
    surf = np.empty(r.shape) if want in ('surface', 'both') else None
    mx = ctypes.c_double()
    arg = ctypes.c_long()
    lib().orc_xcorr2d_circular(
        _p(t), t.shape[0], t.shape[1], ctypes.c_long(t.shape[1]),
        _p(r), r.shape[0], r.shape[1], ctypes.c_long(r.shape[1]),
        ctypes.c_double(MIN_ST_DEV), ctypes.c_double(MIN_MEAN_DIFF),
        _p(surf) if surf is not None else None, ctypes.byref(mx), ctypes.byref(arg))
    if want == 'surface':
        return surf
    if want == 'max':
        return mx.value
    return surf, mx.value, arg.value


def cross_correlate_2d_spectral_fft(template, region):
    '''The same function through numpy's FFT (correlation_spectral.py:37-54),
    for sizes where the direct O(N^2) evaluation is too slow on a CPU.'''
    template = np.asarray(template)
    region = np.asarray(region)
    t = template - np.mean(template)
    r = region - np.mean(region)
    if np.std(template) < MIN_ST_DEV:
        same = abs(np.mean(template) - np.mean(region)) < MIN_MEAN_DIFF
        return np.ones(region.shape) if same else np.zeros(region.shape)
    with np.errstate(all='ignore'):
        r = r / np.std(region)
        t = t / (np.std(template) * template.size)
        ft = np.fft.fft2(t, s=region.shape)
        fr = np.fft.fft2(r, s=region.shape)
        return np.real(np.fft.ifft2(np.conj(ft) * fr))


# -- a3 / a4: correlation.py:4-75 ---------------------------------------------
def cross_correlate_1d(vec_1, vec_2, normalised=False):
    v1, v2 = _f64(vec_1), _f64(vec_2)
    if len(v1) != len(v2):
        raise Exception('Vectors are of unequal length.')
    n = len(v1)
    out = np.empty(2 * n - 1)
    lib().orc_xcorr1d(_p(v1), _p(v2), ctypes.c_long(n), int(bool(normalised)), _p(out))
    return out


# -- a5: correlation_spectral.py:12-30 ----------------------------------------
def cross_correlate_1d_spectral(template, signal):
    t, s = _f64(template), _f64(signal)
    out = np.empty(len(t) + len(s) - 1)
    lib().orc_xcorr1d_circular(_p(t), ctypes.c_long(len(t)), _p(s), ctypes.c_long(len(s)), _p(out))
    return out


# -- a11: utils.py:17-26 --------------------------------------------------------
def max_pos(array, step_x=1, step_y=1):
    idx = int(np.nanargmax(array))
    ncols = array.shape[1]
    return (step_y * (idx % ncols), step_x * (idx // ncols))


# -- dense band scan (spatial NCC at full overlap, correlation.py:104-116) -----
def band_scan(left, right, win_w, win_h, max_disp, stride_x=1, stride_y=1, mode='argmax',
              threshold=0.0, centre_tol=CENTRE_TOL, want_scores=False):
    l, r = _f64(left), _f64(right)
    assert l.shape == r.shape
    H, W = l.shape
    nwx = (W - win_w) // stride_x + 1
    nwy = (H - win_h) // stride_y + 1
    disp = np.empty((nwy, nwx), dtype=np.int32)
    val = np.empty((nwy, nwx))
    scores = np.empty((nwy, nwx, 2 * max_disp + 1)) if want_scores else None
    lib().orc_band_scan(_p(l), _p(r), H, W, int(win_w), int(win_h), int(stride_x), int(stride_y),
                        int(max_disp), 0 if mode == 'argmax' else 1,
                        ctypes.c_double(threshold), ctypes.c_double(centre_tol),
                        disp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _p(val),
                        _p(scores) if scores is not None else None)
    return (disp, val, scores) if want_scores else (disp, val)


# -- a9: sv_image_comparison.py:283-298 ---------------------------------------
def window_box(cx, cy, w, h, img_w, img_h):
    if cx + w / 2 <= 0 or cy + h / 2 <= 0 or cx - w / 2 > img_w or cy - h / 2 > img_h:
        return None
    return (int(max(cx - w / 2, 0)), int(min(cx + w / 2, img_w)),
            int(max(cy - h / 2, 0)), int(min(cy + h / 2, img_h)))


# -- a8: region_pairs, sv_image_comparison.py:119-141 -------------------------
def candidate_centres(cx, w, cy, h, scheme, shift):
    if scheme[0] % 2 != 1 or scheme[1] % 2 != 1:
        raise Exception('Scheme dimensions must be odd.')
    half_x = (shift[0] if shift[0] else w) * (scheme[0] - 1) / 2
    half_y = (shift[1] if shift[1] else h) * (scheme[1] - 1) / 2
    xs = np.linspace(cx - half_x, cx + half_x, scheme[0])
    ys = np.linspace(cy - half_y, cy + half_y, scheme[1])
    out = [(cx, cy)]
    out += [(x, y) for y in ys for x in xs if not (x == cx and y == cy)]
    return out


# -- a10: region_partition_pairs, sv_image_comparison.py:193-204 --------------
def partition_centres(cx, w, cy, h, factor):
    xs = np.linspace(cx - w / 2, cx + w / 2, 2 * factor + 1)[1::2]
    ys = np.linspace(cy - h / 2, cy + h / 2, 2 * factor + 1)[1::2]
    return [(x, y) for y in ys for x in xs]


def _fold(template, centre, regions, threshold, centre_tol):
    '''image_scan's selection rule, sv_image_comparison.py:236-255.  regions:
    list of (centre, slice).'''
    best, pos = 0, centre
    for rc, rs in regions:
        if rs.size == 0:
            continue
        c = cross_correlate_2d_spectral(template, rs, want='max')
        if c > threshold and ((rc == centre and c >= best) or (rc != centre and c > best + centre_tol)):
            best, pos = c, rc
    return pos, best


def sequence_scan(left, right, scan_config, centre_tol=CENTRE_TOL):
    '''Window-by-window restatement of sv_image_comparison.py:21-117.  Returns,
    per stage, a dict id -> {row, column, centre, dp_x, dp_y, stage_centres,
    stage_sizes, corr_max} (no image slices).'''
    left, right = _f64(left), _f64(right)
    if left.shape != right.shape:
        raise Exception('Dimensions of template and region images must match')
    H, W = left.shape
    stages = []
    for stage, cfg in enumerate(scan_config):
        cur = {}
        if stage == 0:
            w, h = cfg['window_width'], cfg['window_height']
            sx = int(w * (1 - cfg['window_overlap']))
            sy = int(h * (1 - cfg['window_overlap']))
            wid = 0
            y = row = 0
            while y < H:
                x = col = 0
                while x < W:
                    centre = (x + w / 2, y + h / 2)
                    regs = []
                    for c in candidate_centres(centre[0], w, centre[1], h,
                                               tuple(cfg['scheme']), tuple(cfg['scheme_shift_size'])):
                        b = window_box(c[0], c[1], w, h, W, H)
                        if b is not None:
                            regs.append((c, right[b[2]:b[3], b[0]:b[1]]))
                    pos, best = _fold(left[y:y + h, x:x + w], centre, regs,
                                      cfg['correlation_threshold'], centre_tol)
                    cur[wid] = dict(row=row, column=col, centre=centre, size=(w, h),
                                    dp_x=pos[0] - centre[0], dp_y=pos[1] - centre[1],
                                    stage_centres=[centre], stage_sizes=[(w, h)], corr_max=best)
                    wid += 1
                    x += sx
                    col += 1
                y += sy
                row += 1
        else:
            f = cfg['factor']
            for wid, prev in stages[stage - 1].items():
                if (prev['dp_x'] == 0 and prev['dp_y'] == 0) or len(prev['stage_sizes']) < stage:
                    continue
                pw, ph = prev['stage_sizes'][stage - 1]
                nw, nh = int(pw / f), int(ph / f)
                prev['stage_sizes'].append((nw, nh))           # list shared between stages
                pcx, pcy = prev['stage_centres'][stage - 1]
                regs = []
                for c in partition_centres(pcx + prev['dp_x'], f * nw, pcy + prev['dp_y'], f * nh, f):
                    b = window_box(c[0], c[1], nw, nh, W, H)
                    if b is not None:
                        regs.append((c, right[b[2]:b[3], b[0]:b[1]]))
                w = dict(prev)                                   # shallow: lists stay shared
                wbest = 0
                for c in partition_centres(pcx, pw, pcy, ph, f):
                    b = window_box(c[0], c[1], pw, ph, W, H)
                    if b is None:
                        continue
                    pos, best = _fold(left[b[2]:b[3], b[0]:b[1]], c, regs, 0, centre_tol)
                    if best > wbest:
                        wbest = best
                        w['centre'] = c
                        w['dp_x'], w['dp_y'] = pos[0] - c[0], pos[1] - c[1]
                        if len(w['stage_centres']) < stage + 1:
                            w['stage_centres'].append(c)
                        else:
                            w['stage_centres'][stage] = c
                w['corr_max'] = wbest
                cur[wid] = w
        stages.append(cur)
    return stages


# ---- displacement grids (reference test_image_comparison.py:49-98) -------------------
def accumulate_sequence(seq_results, shape):
    '''The four grids one sequence paints: dp_x, dp_y, |dp| and a contribution
    count over each window's box; only the deepest stage a window reached
    contributes (reference test_image_comparison.py:61-88).'''
    image_height, image_width = shape
    dp_x_grid, dp_y_grid = np.zeros(shape), np.zeros(shape)
    depth_grid, contributions = np.zeros(shape), np.zeros(shape)
    for stage, windows in enumerate(seq_results):
        for window_info in windows.values():
            reached = len(window_info['stage_centres']) - 1
            if reached > stage:
                continue
            x, y = window_info['stage_centres'][stage - 1 if reached < stage else stage]
            magnitude = np.sqrt(window_info['dp_x'] ** 2 + window_info['dp_y'] ** 2)
            x_window, y_window = window_info['stage_sizes'][stage]
            box = (slice(int(max(y - y_window / 2, 0)), int(min(y + y_window / 2, image_height))),
                   slice(int(max(x - x_window / 2, 0)), int(min(x + x_window / 2, image_width))))
            dp_x_grid[box] += window_info['dp_x']
            dp_y_grid[box] += window_info['dp_y']
            depth_grid[box] += magnitude
            if magnitude > 0:
                contributions[box] += 1
    return depth_grid, dp_x_grid, dp_y_grid, contributions


def depth_grids(all_seq_results, shape):
    '''Sum over sequences, then the average by contributions with 0 -> infinity
    (reference test_image_comparison.py:89-98).  Returns (depth, dp_x, dp_y,
    contributions).'''
    totals = [np.zeros(shape) for _ in range(4)]
    for seq_results in all_seq_results:
        for total, grid in zip(totals, accumulate_sequence(seq_results, shape)):
            total += grid
    depth, dp_x, dp_y, contributions = totals
    contributions[contributions == 0] = np.inf
    return depth / contributions, dp_x / contributions, dp_y / contributions, contributions


def band_depth_grids(disparity, shape, win_w, win_h, stride_y=1):
    '''The same grids for a dense band scan's disparity map (one window per
    origin, dp_y = 0), before the averaging.'''
    depth, dp_x, contributions = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for wy in range(disparity.shape[0]):
        for wx in range(disparity.shape[1]):
            d = int(disparity[wy, wx])
            if d == -2 ** 31:
                d = 0
            box = (slice(wy * stride_y, wy * stride_y + win_h), slice(wx, wx + win_w))
            dp_x[box] += d
            depth[box] += abs(d)
            if d != 0:
                contributions[box] += 1
    return depth, dp_x, contributions


# ---- calibration peak finder (reference sv_calibration.py:115-141) ----------------------
def _neighbourhood_extreme(data, size, fn):
    '''scipy.ndimage maximum_filter / minimum_filter with a scalar size: window
    offsets -(size // 2) .. size - size // 2 - 1, borders reflected about the edge.'''
    lo = size // 2
    hi = size - lo - 1
    padded = np.pad(data, ((lo, hi), (lo, hi)), mode='symmetric')
    out = None
    for dy in range(size):
        for dx in range(size):
            part = padded[dy:dy + data.shape[0], dx:dx + data.shape[1]]
            out = part.copy() if out is None else fn(out, part)
    return out


def peak_mask(data, size, threshold):
    data = _f64(data)
    data_max = _neighbourhood_extreme(data, size, np.maximum)
    data_min = _neighbourhood_extreme(data, size, np.minimum)
    return (data == data_max) & ((data_max - data_min) > threshold)


def _label_boxes(mask):
    '''Bounding boxes (y0, y1, x0, x1; inclusive) of the 4-connected groups of a
    mask, ordered by each group's first pixel in raster order (the order of
    scipy.ndimage.label + find_objects).'''
    h, w = mask.shape
    seen = np.zeros_like(mask, dtype=bool)
    boxes = []
    for y, x in zip(*np.nonzero(mask)):
        if seen[y, x]:
            continue
        stack = [(y, x)]
        seen[y, x] = True
        y0 = y1 = y
        x0 = x1 = x
        while stack:
            cy, cx = stack.pop()
            y0, y1, x0, x1 = min(y0, cy), max(y1, cy), min(x0, cx), max(x1, cx)
            for ny, nx in ((cy - 1, cx), (cy + 1, cx), (cy, cx - 1), (cy, cx + 1)):
                if 0 <= ny < h and 0 <= nx < w and mask[ny, nx] and not seen[ny, nx]:
                    seen[ny, nx] = True
                    stack.append((ny, nx))
        boxes.append((y0, y1, x0, x1))
    return boxes


def find_peaks(region, neighborhood_size, threshold):
    grid = np.linspace(-5, 5, 101)
    x_grid, y_grid = np.meshgrid(grid, grid)
    template = np.exp(-(x_grid ** 2 + y_grid ** 2))
    data = cross_correlate_2d_spectral(template, region)
    boxes = _label_boxes(peak_mask(data, neighborhood_size, threshold))
    x = np.array([(b[2] + b[3]) / 2 for b in boxes]) + template.shape[1] / 2
    y = np.array([(b[0] + b[1]) / 2 for b in boxes]) + template.shape[0] / 2
    return x, y


# ---- image preparation (reference utils.py:6-15, 28-35) -------------------------------
def greyscale(pixels, nan_alpha=False):
    '''PIL's "L" conversion of 8-bit pixels (H x W, H x W x 2 grey + alpha, RGB or
    RGBA): (19595 R + 38470 G + 7471 B + 32768) >> 16.  With nan_alpha the result
    is float64 with NaN where alpha is 0 (reference utils.py:11-15), else uint8.'''
    pixels = np.asarray(pixels)
    if pixels.ndim == 2:
        grey, alpha = pixels.astype(np.uint32), None
    elif pixels.shape[2] == 2:
        grey, alpha = pixels[..., 0].astype(np.uint32), pixels[..., 1]
    else:
        r, g, b = (pixels[..., c].astype(np.uint32) for c in range(3))
        grey = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
        alpha = pixels[..., 3] if pixels.shape[2] == 4 else None
    if not nan_alpha:
        return grey.astype(np.uint8)
    out = grey.astype(np.float64)
    if alpha is not None:
        out[alpha == 0] = np.nan
    return out


def _mirror(i, n):
    '''Index folding for borders mirrored about the edge pixel: c b | a b c | b a.'''
    if n == 1:
        return np.zeros_like(i)
    p = 2 * n - 2
    i = np.mod(i, p)
    return np.where(i < n, i, p - i)


def _gaussian_axis(image, sigma, axis):
    '''scipy.ndimage.gaussian_filter1d (truncate 4, mode "mirror"): centre tap
    first, then the tap pairs from the outermost inwards.'''
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    data = np.moveaxis(image, axis, 0)
    n = data.shape[0]
    c = np.arange(n)
    acc = data * phi[radius]
    for k in range(radius, 0, -1):
        acc = acc + (data[_mirror(c - k, n)] + data[_mirror(c + k, n)]) * phi[radius - k]
    return np.moveaxis(acc, 0, axis)


def _zoom_axis_plan(n_in, n_out):
    '''Source rows and weights of scipy.ndimage.zoom(order=1, mode="mirror",
    grid_mode=True) along one axis.'''
    zoom = n_in / n_out
    cc = (np.arange(n_out) + 0.5) * zoom - 0.5
    if n_in > 1:
        sz2 = 2 * n_in - 2
        neg = cc < 0
        folded = sz2 * np.trunc(-cc / sz2) + cc
        folded = np.where(folded <= 1 - n_in, folded + sz2, -folded)
        over = cc > n_in - 1
        wrapped = cc - sz2 * np.trunc(cc / sz2)
        wrapped = np.where(wrapped > n_in - 1, sz2 - wrapped, wrapped)
        cc = np.where(neg, folded, np.where(over, wrapped, cc))
    else:
        cc = np.zeros(n_out)
    start = np.floor(cc)
    w0 = 1.0 - (cc - start)
    w1 = 1.0 - w0
    start = start.astype(np.int64)
    return _mirror(start, n_in), _mirror(start + 1, n_in), w0, w1


def resize(image, out_shape):
    '''scikit-image 0.19 `resize(image, out_shape)` with its defaults, for 2D uint8
    or float64 images: scaling of integer values by 1 / 255, Gaussian anti-aliasing
    of sigma (in / out - 1) / 2 per shrinking axis, linear interpolation on the
    pixel-area grid, clip to the scaled input's range (NaNs ignored).'''
    image = np.asarray(image)
    if image.dtype == np.uint8:
        image = image * (1.0 / 255)
    else:
        image = image.astype(np.float64)
    h, w = image.shape
    oh, ow = out_shape
    filtered = image
    if oh < h or ow < w:
        for axis, (n_in, n_out) in enumerate(((h, oh), (w, ow))):
            sigma = max(0.0, (n_in / n_out - 1) / 2)
            if sigma > 1e-15:
                filtered = _gaussian_axis(filtered, sigma, axis)
    y0, y1, wy0, wy1 = _zoom_axis_plan(h, oh)
    x0, x1, wx0, wx1 = _zoom_axis_plan(w, ow)
    wy0, wy1 = wy0[:, None], wy1[:, None]
    out = np.zeros((oh, ow))
    out = out + filtered[np.ix_(y0, x0)] * wy0 * wx0
    out = out + filtered[np.ix_(y0, x1)] * wy0 * wx1
    out = out + filtered[np.ix_(y1, x0)] * wy1 * wx0
    out = out + filtered[np.ix_(y1, x1)] * wy1 * wx1
    if not np.all(np.isnan(image)):
        lo, hi = np.nanmin(image), np.nanmax(image)
        out = np.where(out < lo, lo, np.where(out > hi, hi, out))
    return out


def downsample(image, factor=1):
    '''reference utils.py:28-35.'''
    image = np.asarray(image)
    return resize(image, (math.floor(image.shape[0] / factor), math.floor(image.shape[1] / factor)))
