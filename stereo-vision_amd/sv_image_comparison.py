'''Stereo image comparison (template-window scan of a left image against search
regions of a right image) on MI355X.

Drop-in for the reference module of the same name (reference
sv_image_comparison.py:1-380): same functions, arguments, module-level knobs,
returned dict/list/tuple structure, exceptions and log lines.  The window and
region enumeration stays on the host; every (template, region) correlation of
a stage, its maximum and the per-window selection rule run on the GPU in two
launches (libsvamd: svamd_scan_pairs, svamd_scan_fold).
'''
import contextlib
import gc
from collections.abc import Sequence

import numpy as np

import correlation_spectral
from correlation_spectral import cross_correlate_2d_spectral as x_corr_2d
from utils import max_pos, resident_float64
import _native as nv
import sharding

# Amount by which the correlation of a non-central region must exceed that of
# the central one to register a displacement (reference sv_image_comparison.py:11)
CENTRE_TOL = 0.01

# Plot every correlation surface (reference :15); forces the pair-at-a-time path
PLOT_XCORR = False

# True / False / 'nonzero': which per-window summaries are printed (reference :19)
LOG_WINDOW_SUMMARY = 'nonzero'

# Arithmetic of the batched correlations: 'mixed' = float32 multiply-accumulate,
# with every window whose selection came within GUARD_BAND of flipping
# re-evaluated in float64; 'fp64' = float64 throughout; 'fp32' = no re-check.
SCAN_PRECISION = 'mixed'
GUARD_BAND = 1e-4

_REFERENCE_CORRELATOR = x_corr_2d


# ---------------------------------------------------------------------------
# geometry (host)
# ---------------------------------------------------------------------------
def get_window_boundaries(x, y, x_window, y_window, image_width, image_height):
    '''Clipped integer box (x_start, x_end, y_start, y_end) of a window given by
    centre and size, or None when it lies outside the image (reference
    sv_image_comparison.py:283-298).'''
    half_w, half_h = x_window / 2, y_window / 2
    if x + half_w <= 0 or y + half_h <= 0 or x - half_w > image_width or y - half_h > image_height:
        return None
    return (int(max(x - half_w, 0)), int(min(x + half_w, image_width)),
            int(max(y - half_h, 0)), int(min(y + half_h, image_height)))


def region_pairs(x_centre, x_window, y_centre, y_window, scheme, shift_size):
    '''Centres of the candidate regions of one window: the window's own centre
    first, then the scheme grid row by row (reference :119-141).'''
    if scheme[0] % 2 != 1 or scheme[1] % 2 != 1:
        raise Exception('Scheme dimensions must be odd.')
    x_span_hw = (shift_size[0] if shift_size[0] else x_window) * (scheme[0] - 1) / 2
    y_span_hw = (shift_size[1] if shift_size[1] else y_window) * (scheme[1] - 1) / 2
    x_vec = np.linspace(x_centre - x_span_hw, x_centre + x_span_hw, scheme[0])
    y_vec = np.linspace(y_centre - y_span_hw, y_centre + y_span_hw, scheme[1])
    pairs = [(x_centre, y_centre)]
    pairs.extend((x, y) for y in y_vec for x in x_vec if not (x == x_centre and y == y_centre))
    return pairs


def region_partition_pairs(x_centre, x_window, y_centre, y_window, factor):
    '''Centres of the factor x factor sub-cells of a box (reference :193-204).'''
    x_vec = np.linspace(x_centre - x_window / 2, x_centre + x_window / 2, 2 * factor + 1)[1::2]
    y_vec = np.linspace(y_centre - y_window / 2, y_centre + y_window / 2, 2 * factor + 1)[1::2]
    return [(x, y) for y in y_vec for x in x_vec]


def _regions_for(centres, x_window, y_window, region_image):
    '''target_regions list and the matching boxes for a list of centres.'''
    image_height, image_width = region_image.shape
    regions, boxes = [], []
    for _x, _y in centres:
        bounds = get_window_boundaries(_x, _y, x_window, y_window, image_width, image_height)
        if bounds is None:
            continue
        x_start, x_end, y_start, y_end = bounds
        regions.append({
            'centre': (_x, _y),
            'size': (x_end - x_start, y_end - y_start),
            'image_slice': region_image[y_start:y_end, x_start:x_end],
        })
        boxes.append(bounds)
    return regions, boxes


def _axis_candidates(starts, window, span_hw, count, limit):
    '''Along one image axis, for every window start: the candidate centres
    (np.linspace, exactly as region_pairs builds them), their clipped integer
    [begin, end) and whether get_window_boundaries would reject them.'''
    centres = starts + window / 2
    grid = np.linspace(centres - span_hw, centres + span_hw, count, axis=-1)        # [n, count]
    half = window / 2
    rejected = (grid + half <= 0) | (grid - half > limit)
    begin = np.maximum(grid - half, 0).astype(np.int64)      # int() truncation of a value >= 0
    end = np.minimum(grid + half, limit).astype(np.int64)
    own_begin = np.maximum(centres - half, 0).astype(np.int64)
    own_end = np.minimum(centres + half, limit).astype(np.int64)
    own_rejected = (centres + half <= 0) | (centres - half > limit)
    return dict(centres=centres.tolist(), grid=grid.tolist(), rejected=rejected.tolist(),
                begin=begin.tolist(), end=end.tolist(), own_begin=own_begin.tolist(),
                own_end=own_end.tolist(), own_rejected=own_rejected.tolist(),
                centre_array=centres, grid_array=grid,
                arrays=dict(rejected=rejected, begin=begin, end=end, own_begin=own_begin, own_end=own_end,
                            own_rejected=own_rejected, at_centre=grid == centres[:, None]))


def _stage0_boxes(ax, ay, xs, ys, x_window, y_window, image_width, image_height):
    '''The boxes of _stage0_windows as arrays, without a Python loop over regions:
    template boxes [windows, 4], region boxes [pairs, 4] (x_start, x_end, y_start,
    y_end) in the order _stage0_windows lists the regions, the number of regions per
    window, which pair is the window's own position, and the regions' centres
    [pairs, 2].'''
    X, Y = ax['arrays'], ay['arrays']
    ncol, nx = X['begin'].shape
    nrow, ny = Y['begin'].shape
    slots = 1 + ny * nx
    boxes = np.empty((nrow, ncol, slots, 4), dtype=np.int64)
    valid = np.empty((nrow, ncol, slots), dtype=bool)
    # slot 0: the window's own position
    boxes[:, :, 0, 0] = X['own_begin'][None, :]
    boxes[:, :, 0, 1] = X['own_end'][None, :]
    boxes[:, :, 0, 2] = Y['own_begin'][:, None]
    boxes[:, :, 0, 3] = Y['own_end'][:, None]
    valid[:, :, 0] = ~(X['own_rejected'][None, :] | Y['own_rejected'][:, None])
    # slots 1 + j * nx + i: the scheme grid, row by row
    grid = boxes[:, :, 1:, :].reshape(nrow, ncol, ny, nx, 4)
    grid[..., 0] = X['begin'][None, :, None, :]
    grid[..., 1] = X['end'][None, :, None, :]
    grid[..., 2] = Y['begin'][:, None, :, None]
    grid[..., 3] = Y['end'][:, None, :, None]
    valid[:, :, 1:] = ~(X['rejected'][None, :, None, :] | Y['rejected'][:, None, :, None] |
                        (X['at_centre'][None, :, None, :] & Y['at_centre'][:, None, :, None])
                        ).reshape(nrow, ncol, ny * nx)
    counts = valid.sum(axis=2).ravel()
    region_boxes = boxes[valid]
    # the regions' centres, same order (what image_scan reports as the match position)
    cx = np.empty((nrow, ncol, slots))
    cy = np.empty((nrow, ncol, slots))
    cx[:, :, 0] = ax['centre_array'][None, :]
    cy[:, :, 0] = ay['centre_array'][:, None]
    cx[:, :, 1:].reshape(nrow, ncol, ny, nx)[...] = ax['grid_array'][None, :, None, :]
    cy[:, :, 1:].reshape(nrow, ncol, ny, nx)[...] = ay['grid_array'][:, None, :, None]
    region_centres = np.stack([cx[valid], cy[valid]], axis=1)
    centre = np.zeros(len(region_boxes), dtype=bool)
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    own = valid[:, :, 0].ravel()
    centre[first[own & (counts > 0)]] = True
    template_boxes = np.empty((nrow, ncol, 4), dtype=np.int64)
    template_boxes[:, :, 0] = xs[None, :]
    template_boxes[:, :, 1] = np.minimum(xs + x_window, image_width)[None, :]
    template_boxes[:, :, 2] = ys[:, None]
    template_boxes[:, :, 3] = np.minimum(ys + y_window, image_height)[:, None]
    return template_boxes.reshape(-1, 4), region_boxes, counts, centre, region_centres


@contextlib.contextmanager
def _gc_paused():
    '''Enumerating a stage creates hundreds of thousands of small dicts and tuples,
    none of them cyclic; the cycle collector's generational passes over them cost
    as much again as building them.'''
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def _stage0_windows(template_image, region_image, x_window, y_window, scheme, scheme_shift_size,
                    window_overlap):
    '''Windows of the first stage plus the arrays of _stage0_boxes (template boxes,
    region boxes, regions per window, own-position flags).  Same enumeration as the
    reference's double loop with region_pairs / get_window_boundaries per window,
    but the candidate grids and their clipping are computed once per window
    column and once per window row.'''
    image_height, image_width = region_image.shape
    if template_image.shape[0] != image_height or template_image.shape[1] != image_width:
        raise Exception('Dimensions of template and region images must match')
    if scheme[0] % 2 != 1 or scheme[1] % 2 != 1:
        raise Exception('Scheme dimensions must be odd.')
    x_stride = int(x_window * (1 - window_overlap))
    y_stride = int(y_window * (1 - window_overlap))
    if x_stride <= 0 or y_stride <= 0:
        raise ValueError('window_overlap leaves a window stride of zero pixels')
    x_span_hw = (scheme_shift_size[0] if scheme_shift_size[0] else x_window) * (scheme[0] - 1) / 2
    y_span_hw = (scheme_shift_size[1] if scheme_shift_size[1] else y_window) * (scheme[1] - 1) / 2
    xs = np.arange(0, image_width, x_stride)
    ys = np.arange(0, image_height, y_stride)
    ax = _axis_candidates(xs, x_window, x_span_hw, scheme[0], image_width)
    ay = _axis_candidates(ys, y_window, y_span_hw, scheme[1], image_height)
    size = (x_window, y_window)

    # per window column / row: the candidates that survive clipping, as tuples
    # (centre, begin, end, is the window's own centre)
    x_cands = [[(g, b, e, g == c) for g, b, e, rej in zip(grid, begin, end, rejected) if not rej]
               for c, grid, begin, end, rejected in zip(ax['centres'], ax['grid'], ax['begin'], ax['end'],
                                                        ax['rejected'])]
    y_cands = [[(g, b, e, g == c) for g, b, e, rej in zip(grid, begin, end, rejected) if not rej]
               for c, grid, begin, end, rejected in zip(ay['centres'], ay['grid'], ay['begin'], ay['end'],
                                                        ay['rejected'])]
    x_own = list(zip(ax['own_rejected'], ax['own_begin'], ax['own_end'], ax['centres']))
    y_own = list(zip(ay['own_rejected'], ay['own_begin'], ay['own_end'], ay['centres']))
    stage_boxes = _stage0_boxes(ax, ay, xs, ys, x_window, y_window, image_width, image_height)
    counts = stage_boxes[2].tolist()
    windows = {}
    window_id = 0
    for row, y in enumerate(ys.tolist()):
        own_y = y_own[row]
        y_cand = y_cands[row]
        for column, x in enumerate(xs.tolist()):
            own_x = x_own[column]
            windows[window_id] = {
                'row': row,
                'column': column,
                'centre': (own_x[3], own_y[3]),
                'size': size,
                'image_slice': template_image[y:y + y_window, x:x + x_window],
                'target_regions': _Stage0Regions(region_image, own_x, own_y, x_cands[column], y_cand,
                                                 counts[window_id]),
            }
            window_id += 1
    return windows, stage_boxes


class _Stage0Regions(Sequence):
    '''The target_regions list of a first-stage window, built when it is first looked
    at.  A scan holds hundreds of thousands of regions, each a dict with a view of
    the region image; sequence_scan itself works from the box arrays and never
    needs them, so they cost nothing unless the caller (plots, image_scan on a
    returned window) reads them.  Behaves as the list it stands for: length,
    indexing, slicing, iteration, comparison with a list.'''
    __slots__ = ('_image', '_own_x', '_own_y', '_x_cand', '_y_cand', '_count', '_items')

    def __init__(self, region_image, own_x, own_y, x_cand, y_cand, count):
        self._image, self._own_x, self._own_y = region_image, own_x, own_y
        self._x_cand, self._y_cand, self._count, self._items = x_cand, y_cand, count, None

    def _list(self):
        if self._items is None:
            image = self._image
            x_rej, ox0, ox1, x_centre = self._own_x
            y_rej, oy0, oy1, y_centre = self._own_y
            # the window's own position first, then the scheme grid row by row
            items = [] if (x_rej or y_rej) else [{'centre': (x_centre, y_centre), 'size': (ox1 - ox0, oy1 - oy0),
                                                  'image_slice': image[oy0:oy1, ox0:ox1]}]
            items += [{'centre': (_x, _y), 'size': (x1 - x0, y1 - y0), 'image_slice': image[y0:y1, x0:x1]}
                      for _y, y0, y1, at_row in self._y_cand
                      for _x, x0, x1, at_column in self._x_cand
                      if not (at_row and at_column)]
            assert len(items) == self._count
            self._items = items
        return self._items

    def __len__(self):
        return self._count

    def __getitem__(self, index):
        return self._list()[index]

    def __iter__(self):
        return iter(self._list())

    def __eq__(self, other):
        return self._list() == (other._list() if isinstance(other, _Stage0Regions) else other)

    def __repr__(self):
        return repr(self._list())


def whole_image_search_regions(template_image, region_image, x_window, y_window,
        scheme=(1, 3), scheme_shift_size=(0, 0), window_overlap=0):
    '''Tiles the template image into windows and lists, for each, the search
    regions of the region image given by the scheme (reference :143-191).'''
    return _stage0_windows(template_image, region_image, x_window, y_window, scheme,
                           scheme_shift_size, window_overlap)[0]


def _multi_pass_regions(template_window_info, x_centre, y_centre, size, region_image, factor):
    x_window, y_window = size
    regions, boxes = _regions_for(
        region_partition_pairs(x_centre, factor * x_window, y_centre, factor * y_window, factor),
        x_window, y_window, region_image)
    return {**template_window_info, 'target_regions': regions}, boxes


def multi_pass_search_regions(template_window_info, x_centre, y_centre, size,
        region_image, factor):
    '''Search regions of a later stage: the factor x factor partition of a box of
    factor * size centred on the previous stage's match (reference :206-227).'''
    return _multi_pass_regions(template_window_info, x_centre, y_centre, size, region_image, factor)[0]


# ---------------------------------------------------------------------------
# correlation + selection (device)
# ---------------------------------------------------------------------------
def _batched_path_available():
    return x_corr_2d is _REFERENCE_CORRELATOR and not PLOT_XCORR


def _select(tbase, rbase, desc, is_centre, skip, win_off, threshold, max_side):
    '''Pairs + selection rule for all windows of a stage.  With an initialised
    torch.distributed process group of N ranks (one per GPU, both images resident
    on each) every rank evaluates a contiguous 1/N of the windows and the
    per-window results are all-gathered; otherwise everything runs locally.'''
    win_off = np.asarray(win_off, dtype=np.int64)
    nwin = len(win_off) - 1
    rank, size = sharding.world()
    if size == 1 or nwin < size:
        return _select_local(tbase, rbase, desc, is_centre, skip, win_off, threshold, max_side)
    begin, end = sharding.shard_range(nwin, rank, size)
    p0, p1 = int(win_off[begin]), int(win_off[end])
    best_pair, best_val = _select_local(tbase, rbase, desc[p0:p1], is_centre[p0:p1], skip[p0:p1],
                                        win_off[begin:end + 1] - p0, threshold, max_side)
    best_pair = np.where(best_pair >= 0, best_pair + p0, -1)
    return sharding.all_gather_concat(best_pair, nwin), sharding.all_gather_concat(best_val, nwin)


def _select_local(tbase, rbase, desc, is_centre, skip, win_off, threshold, max_side):
    '''Runs every pair and the selection rule on the GPU.  desc: P x 8 int64,
    win_off: nwin + 1 offsets.  Returns (best_pair, best_val) as numpy arrays.'''
    torch = nv.torch_cuda()
    npairs, nwin = len(desc), len(win_off) - 1
    if nwin == 0:
        return np.zeros(0, np.int64), np.zeros(0)
    min_std = float(correlation_spectral.MIN_ST_DEV)
    min_diff = float(correlation_spectral.MIN_MEAN_DIFF)
    d_desc = nv.to_device(desc)
    d_centre = nv.to_device(is_centre.astype(np.uint8))
    d_skip = nv.to_device(skip.astype(np.uint8))
    d_off = nv.to_device(win_off.astype(np.int64))
    corr = nv.empty(max(npairs, 1), torch.float64)
    best_pair = nv.empty(nwin, torch.int64)
    best_val = nv.empty(nwin, torch.float64)
    margin = nv.empty(nwin, torch.float64)
    stream = nv.stream_ptr()

    def pairs(desc_dev, count, precision, out):
        nv.call('svamd_scan_pairs', nv.ptr(tbase), nv.ptr(rbase), nv.ptr(desc_dev), count, max_side,
                min_std, min_diff, precision, nv.ptr(out), None, None, None, stream)

    def fold():
        nv.call('svamd_scan_fold', nv.ptr(corr), nv.ptr(d_centre), nv.ptr(d_skip), nv.ptr(d_off), nwin,
                float(threshold), float(CENTRE_TOL), nv.ptr(best_pair), nv.ptr(best_val),
                nv.ptr(margin), stream)

    if SCAN_PRECISION not in ('mixed', 'fp64', 'fp32'):
        raise ValueError(f'unknown SCAN_PRECISION {SCAN_PRECISION!r}')
    pairs(d_desc, npairs, nv.ACC_F64 if SCAN_PRECISION == 'fp64' else nv.ACC_F32, corr)
    fold()
    if SCAN_PRECISION == 'mixed' and npairs:
        close = (margin < GUARD_BAND).cpu().numpy()
        if close.any():
            # pairs of the windows whose selection was within the guard band
            counts = np.diff(win_off)
            redo = np.repeat(close, counts).nonzero()[0]
            if redo.size:
                d_redo = nv.to_device(redo.astype(np.int64))
                sub = d_desc.view(-1, 8).index_select(0, d_redo).contiguous()
                exact = nv.empty(redo.size, torch.float64)
                pairs(sub, int(redo.size), nv.ACC_F64, exact)
                corr.index_copy_(0, d_redo, exact)
                fold()
    if SCAN_PRECISION == 'mixed' and npairs:
        # The value reported for a window (printed, and compared between the partitions
        # of a later stage) is its selected pair's maximum: re-evaluated in float64, so
        # that it does not carry the float32 accumulation's last digits.
        matched = (best_pair >= 0).nonzero().squeeze(1)
        if matched.numel():
            chosen = best_pair.index_select(0, matched)
            sub = d_desc.view(-1, 8).index_select(0, chosen).contiguous()
            exact = nv.empty(int(chosen.numel()), torch.float64)
            pairs(sub, int(chosen.numel()), nv.ACC_F64, exact)
            best_val.index_copy_(0, matched, exact)
    return best_pair.cpu().numpy(), best_val.cpu().numpy()


def _log_summary(window_id, centre, corr_max, corr_max_pos):
    dp_x = corr_max_pos[0] - centre[0]
    dp_y = corr_max_pos[1] - centre[1]
    if LOG_WINDOW_SUMMARY == True or \
            (LOG_WINDOW_SUMMARY == 'nonzero' and (dp_x != 0 or dp_y != 0)):
        print(f'Window {window_id}: '
              f'centre at {centre}, '
              f'max correlation of {round(corr_max, 5)} at {corr_max_pos}, '
              f'shift of ({dp_x}, {dp_y})')


def _image_scan_pairwise(window_info, window_id, corr_threshold):
    '''One correlation at a time through the module-level `x_corr_2d` hook: used
    when that hook has been replaced, when surfaces are to be plotted, or when a
    window is larger than the batched kernel's tile.'''
    template = window_info['image_slice']
    centre = window_info['centre']
    window_corr_max = 0
    window_corr_max_pos = centre
    for region_info in window_info['target_regions']:
        target_centre = region_info['centre']
        if region_info['image_slice'].size == 0:
            continue
        x_corr = x_corr_2d(template, region_info['image_slice'])
        corr_max = np.max(x_corr)
        is_centre = target_centre == centre
        bar = window_corr_max if is_centre else window_corr_max + CENTRE_TOL
        if corr_max > corr_threshold and (corr_max >= bar if is_centre else corr_max > bar):
            window_corr_max = corr_max
            window_corr_max_pos = target_centre
        if PLOT_XCORR:
            _plot_pair(template, region_info, centre, x_corr, corr_max)
    _log_summary(window_id, centre, window_corr_max, window_corr_max_pos)
    return window_corr_max_pos, window_corr_max


def _plot_pair(template, region_info, centre, x_corr, corr_max):
    from matplotlib import pyplot as plt
    from matplotlib.patches import Circle
    local = max_pos(x_corr)
    figure = plt.figure()
    for k, (title, data) in enumerate((
            (f'Template; centre at {centre}', template),
            (f'Target region; centre at {region_info["centre"]}', region_info['image_slice']),
            (f'Max XC of {round(corr_max, 4)} at local pos {tuple(reversed(local))}', x_corr))):
        ax = figure.add_subplot(1, 3, k + 1)
        ax.set_title(title)
        if k == 2:
            ax.add_patch(Circle(local, radius=1, color='red'))
        plt.imshow(data)
    plt.show()


def _max_side(*shapes):
    return max((max(s) for s in shapes if len(s)), default=0)


def image_scan(window_info, window_id, corr_threshold=0):
    '''Correlates one window with each of its target regions and returns the
    centre of the selected region and its maximum correlation; (window centre,
    0) when no region qualifies (reference sv_image_comparison.py:229-281).'''
    template = np.asarray(window_info['image_slice'])
    regions = window_info['target_regions']
    centre = window_info['centre']
    sides = _max_side(template.shape, *(np.shape(r['image_slice']) for r in regions))
    if not _batched_path_available() or sides > nv.PAIR_MAX_SIDE or template.size == 0:
        return _image_scan_pairwise(window_info, window_id, corr_threshold)

    # pack the template and the regions into one float64 buffer
    chunks = [np.ascontiguousarray(template, dtype=np.float64).ravel()]
    offset = template.size
    th, tw = template.shape
    desc = np.zeros((len(regions), 8), dtype=np.int64)
    skip = np.zeros(len(regions), dtype=bool)
    is_centre = np.zeros(len(regions), dtype=bool)
    for i, region_info in enumerate(regions):
        sl = np.asarray(region_info['image_slice'])
        skip[i] = sl.size == 0
        is_centre[i] = region_info['centre'] == centre
        rh, rw = sl.shape if sl.ndim == 2 else (0, 0)
        desc[i] = (0, tw, th, tw, offset, rw, rh, rw)
        chunks.append(np.ascontiguousarray(sl, dtype=np.float64).ravel())
        offset += sl.size
    if len(regions) == 0:
        _log_summary(window_id, centre, 0, centre)
        return centre, 0
    buf = nv.to_device(np.concatenate(chunks))
    best_pair, best_val = _select(buf, buf, desc, is_centre, skip, np.array([0, len(regions)]),
                                  corr_threshold, max(sides, 1))
    if best_pair[0] < 0:
        pos, val = centre, 0
    else:
        pos, val = regions[int(best_pair[0])]['centre'], float(best_val[0])
    _log_summary(window_id, centre, val, pos)
    return pos, val


class _StageBatch:
    '''Accumulates the (template, regions) groups of one stage for a single
    pair of launches; descriptor rows {t_off, t_pitch, t_h, t_w, r_off, r_pitch,
    r_h, r_w} are built in one vectorised step.'''

    def __init__(self, width_t, width_r):
        self.width_t, self.width_r = width_t, width_r
        self.template_boxes, self.region_boxes, self.centre, self.offsets = [], [], [], [0]
        self._desc = None

    def add_arrays(self, template_boxes, region_boxes, counts, centre):
        '''A whole stage at once, from the arrays of _stage0_boxes.'''
        assert not self.template_boxes
        self.template_boxes = template_boxes
        self.region_boxes = region_boxes
        self.centre = centre
        self.offsets = np.concatenate([[0], np.cumsum(counts)]).tolist()
        self._desc = None

    def add(self, template_box, centre, regions, region_boxes):
        self.template_boxes.append(template_box)
        self.region_boxes.extend(region_boxes)
        self.centre.extend(r['centre'] == centre for r in regions)
        self.offsets.append(self.offsets[-1] + len(regions))
        self._desc = None

    def desc(self):
        if self._desc is None:
            tb = np.asarray(self.template_boxes, dtype=np.int64).reshape(-1, 4)
            rb = np.asarray(self.region_boxes, dtype=np.int64).reshape(-1, 4)
            tb = np.repeat(tb, np.diff(self.offsets), axis=0)
            d = np.empty((len(rb), 8), dtype=np.int64)
            d[:, 0] = tb[:, 2] * self.width_t + tb[:, 0]
            d[:, 1] = self.width_t
            d[:, 2] = tb[:, 3] - tb[:, 2]
            d[:, 3] = tb[:, 1] - tb[:, 0]
            d[:, 4] = rb[:, 2] * self.width_r + rb[:, 0]
            d[:, 5] = self.width_r
            d[:, 6] = rb[:, 3] - rb[:, 2]
            d[:, 7] = rb[:, 1] - rb[:, 0]
            self._desc = d
        return self._desc

    def max_side(self):
        d = self.desc()
        return int(d[:, [2, 3, 6, 7]].max()) if len(d) else 0

    def run(self, left_dev, right_dev, threshold):
        desc = self.desc()
        centre = np.asarray(self.centre, dtype=bool)
        skip = (desc[:, 6] <= 0) | (desc[:, 7] <= 0)
        return _select(left_dev, right_dev, desc, centre, skip, np.asarray(self.offsets),
                       threshold, max(self.max_side(), 1))


def sequence_scan(left_image, right_image, scan_config):
    '''Compares a left and a right image by window searching over the stages of
    one sequence and returns one dict of windows per stage (reference
    sv_image_comparison.py:21-117).'''
    with _gc_paused():
        return _sequence_scan(left_image, right_image, scan_config)


def _sequence_scan(left_image, right_image, scan_config):
    seq_results = []
    image_width = left_image.shape[1]
    image_height = left_image.shape[0]
    print(f'Image width: {image_width}, height: {image_height}')
    batched = _batched_path_available()
    left_dev = right_dev = None

    def resident():
        nonlocal left_dev, right_dev
        if left_dev is None:
            left_dev = resident_float64(left_image)
            right_dev = resident_float64(right_image)
        return left_dev, right_dev

    for stage, stage_config in enumerate(scan_config):
        windows = {}
        if stage == 0:
            windows, stage_boxes = _stage0_windows(
                template_image=left_image,
                region_image=right_image,
                x_window=stage_config['window_width'],
                y_window=stage_config['window_height'],
                scheme=tuple(stage_config['scheme']),
                scheme_shift_size=tuple(stage_config['scheme_shift_size']),
                window_overlap=stage_config['window_overlap'],
            )
            threshold = stage_config['correlation_threshold']
            for window_info in windows.values():
                window_info['stage_centres'] = [window_info['centre']]
                window_info['stage_sizes'] = [window_info['size']]
            batch = _StageBatch(image_width, right_image.shape[1])
            batch.add_arrays(*stage_boxes[:4])
            if batched and batch.max_side() <= nv.PAIR_MAX_SIDE:
                best_pair, best_val = batch.run(*resident(), threshold)
                # the selected region's centre straight from the centre array: the region
                # dicts themselves are built only if somebody reads them
                region_centres = stage_boxes[4]
                matched = best_pair >= 0
                chosen = region_centres[np.where(matched, best_pair, 0)].tolist()
                values = best_val.tolist()
                log_all = LOG_WINDOW_SUMMARY == True          # noqa: E712 (the reference compares this way)
                for i, (window_id, window_info) in enumerate(windows.items()):
                    centre = window_info['centre']
                    if matched[i]:
                        pos, val = tuple(chosen[i]), values[i]
                    else:
                        pos, val = centre, 0
                    dp_x, dp_y = pos[0] - centre[0], pos[1] - centre[1]
                    if log_all or dp_x != 0 or dp_y != 0:
                        _log_summary(window_id, centre, val, pos)
                    window_info['dp_x'] = dp_x
                    window_info['dp_y'] = dp_y
            else:
                for window_id, window_info in windows.items():
                    pos, _ = _image_scan_pairwise(window_info, window_id, threshold)
                    window_info['dp_x'] = pos[0] - window_info['centre'][0]
                    window_info['dp_y'] = pos[1] - window_info['centre'][1]
        else:
            factor = stage_config['factor']
            region_boxes = {}
            for window_id, window_info in seq_results[stage - 1].items():
                # windows without a displacement in the previous stage drop out
                if (window_info['dp_x'] == 0 and window_info['dp_y'] == 0) or \
                        len(window_info['stage_sizes']) < stage:
                    continue
                prev_w, prev_h = window_info['stage_sizes'][stage - 1]
                window_info['stage_sizes'].append((int(prev_w / factor), int(prev_h / factor)))
                prev_centre = window_info['stage_centres'][stage - 1]
                windows[window_id], region_boxes[window_id] = _multi_pass_regions(
                    template_window_info=window_info,
                    x_centre=prev_centre[0] + window_info['dp_x'],
                    y_centre=prev_centre[1] + window_info['dp_y'],
                    size=window_info['stage_sizes'][stage],
                    region_image=right_image,
                    factor=factor,
                )
            # template partitions: sub-cells of the previous window, each still of
            # the previous window's size
            partitions = {}
            batch = _StageBatch(image_width, right_image.shape[1])
            for window_id, window_info in windows.items():
                x_centre, y_centre = window_info['stage_centres'][stage - 1]
                x_window, y_window = window_info['stage_sizes'][stage - 1]
                parts = []
                for _x, _y in region_partition_pairs(x_centre, x_window, y_centre, y_window, factor):
                    bounds = get_window_boundaries(_x, _y, x_window, y_window, image_width, image_height)
                    if bounds is None:
                        continue
                    x_start, x_end, y_start, y_end = bounds
                    parts.append({
                        'centre': (_x, _y),
                        'size': (x_end - x_start, y_end - y_start),
                        'image_slice': left_image[y_start:y_end, x_start:x_end],
                    })
                    batch.add(bounds, (_x, _y), window_info['target_regions'], region_boxes[window_id])
                partitions[window_id] = parts
            use_batch = batched and batch.max_side() <= nv.PAIR_MAX_SIDE and \
                all(p['image_slice'].size for parts in partitions.values() for p in parts)
            if use_batch:
                best_pair, best_val = batch.run(*resident(), 0)
            group = 0
            for window_id, window_info in windows.items():
                window_corr_max = 0
                for wp in partitions[window_id]:
                    if use_batch:
                        if best_pair[group] < 0:
                            pos, corr_max = wp['centre'], 0
                        else:
                            local = int(best_pair[group]) - batch.offsets[group]
                            pos = window_info['target_regions'][local]['centre']
                            corr_max = float(best_val[group])
                        _log_summary(window_id, wp['centre'], corr_max, pos)
                    else:
                        pos, corr_max = _image_scan_pairwise(
                            {**wp, 'target_regions': window_info['target_regions']}, window_id, 0)
                    group += 1
                    if corr_max > window_corr_max:
                        window_corr_max = corr_max
                        window_info['centre'] = wp['centre']
                        window_info['dp_x'] = pos[0] - wp['centre'][0]
                        window_info['dp_y'] = pos[1] - wp['centre'][1]
                        if len(window_info['stage_centres']) < stage + 1:
                            window_info['stage_centres'].append(wp['centre'])
                        else:
                            window_info['stage_centres'][stage] = wp['centre']
        seq_results.append(windows)
    return seq_results


# ---------------------------------------------------------------------------
# plotting (host, matplotlib)
# ---------------------------------------------------------------------------
def plot_sequence_output(left_image, right_image, sequence_results,
        max_shift_magnitude, shift_plot_type):
    '''Overlays the window grid, the later-stage templates / target regions and
    the displacements of one sequence on the two images (counterpart of
    reference sv_image_comparison.py:326-380).'''
    from matplotlib import pyplot as plt
    from matplotlib.patches import Rectangle

    def box(centre, size, **kw):
        corner = (centre[0] - int(size[0] / 2), centre[1] - int(size[1] / 2))
        return Rectangle(corner, *size, **kw)

    figure = plt.figure(figsize=(1, 2))
    left_plot = figure.add_subplot(1, 2, 1)
    left_plot.imshow(left_image)
    right_plot = figure.add_subplot(1, 2, 2)
    right_plot.imshow(right_image)
    last = len(sequence_results) - 1
    for stage, windows in enumerate(sequence_results):
        for window_info in windows.values():
            if len(window_info['stage_centres']) < stage + 1:
                continue
            if stage == 0:
                x, y = window_info['centre']
                for plot in (left_plot, right_plot):
                    plot.add_patch(box((x, y), window_info['size'], linewidth=1, edgecolor='blue',
                                       facecolor='none'))
            else:
                for region_info in window_info['target_regions']:
                    for plot in (left_plot, right_plot):
                        plot.add_patch(box(region_info['centre'], region_info['size'], linewidth=1,
                                           edgecolor='green', facecolor='none'))
                x, y = window_info['stage_centres'][stage]
                left_plot.add_patch(box((x, y), window_info['stage_sizes'][stage], linewidth=1,
                                        edgecolor='cyan', facecolor='none'))
            if stage == last and (window_info['dp_x'] or window_info['dp_y']):
                if shift_plot_type == 'arrows':
                    left_plot.arrow(x, y, window_info['dp_x'], window_info['dp_y'], head_width=5,
                                    head_length=5, edgecolor='red', facecolor='red')
                elif shift_plot_type == 'boxes':
                    magnitude = np.sqrt(window_info['dp_x'] ** 2 + window_info['dp_y'] ** 2)
                    colour = (1, 0, 0) if stage == 0 else (0, 1, 1)
                    left_plot.add_patch(box((x, y), window_info['stage_sizes'][stage], linewidth=1,
                                            edgecolor='none',
                                            facecolor=(*colour, min(magnitude / max_shift_magnitude, 1))))
    plt.show()
