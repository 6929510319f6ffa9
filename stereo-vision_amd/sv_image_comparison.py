'''Stereo image comparison (template-window scan of a left image against search
regions of a right image) on MI355X.

Drop-in for the reference module of the same name (reference
sv_image_comparison.py:1-380): same functions, arguments, module-level knobs,
returned dict/list/tuple structure, exceptions and log lines.  The first stage
tiles the image on a regular grid, so the host only works out the candidate
positions along each axis; the (window, region) pairs are derived on the GPU
from those two small tables, correlated, reduced and put through the
per-window selection rule there, a band of window rows at a time (libsvamd:
svamd_scan_grid), and the windows the caller gets back are built when they
are looked at.  Later stages, which follow the few windows that moved, list
their pairs on the host (svamd_scan_pairs, svamd_scan_fold).
'''
import contextlib
import gc
from collections.abc import Mapping, Sequence

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
    '''The windows of a first stage as a dict, every window built (what
    whole_image_search_regions returns; sequence_scan itself works from the grid's
    arrays and builds windows only when they are read).  Same enumeration as the
    reference's double loop with region_pairs / get_window_boundaries per window, but the
    candidate grids and their clipping are computed once per window column and once per
    window row.'''
    grid = _Stage0Grid(template_image, region_image, x_window, y_window, scheme, scheme_shift_size,
                       window_overlap)
    windows = {}
    for window_id, window_info in _Stage0Windows(grid, None, None).items():
        del window_info['stage_centres'], window_info['stage_sizes'], window_info['dp_x'], window_info['dp_y']
        windows[window_id] = window_info
    return windows, grid


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


# pairs per svamd_scan_grid call: bounds the device workspace (about 90 bytes per pair)
GRID_CHUNK_PAIRS = 1 << 22


class _Axis:
    '''Candidate positions along one image axis for every window start on it: the
    window's own position (entry 0) and the scheme's np.linspace grid, exactly as
    region_pairs / get_window_boundaries compute them, as arrays.'''

    def __init__(self, starts, window, span_hw, count, limit):
        half = window / 2
        self.starts, self.window, self.limit, self.count = starts, window, limit, count
        self.centres = centres = starts + half
        self.grid = grid = np.linspace(centres - span_hw, centres + span_hw, count, axis=-1)      # [n, count]
        self.rejected = (grid + half <= 0) | (grid - half > limit)
        self.begin = np.maximum(grid - half, 0).astype(np.int64)     # int() truncation of a value >= 0
        self.end = np.minimum(grid + half, limit).astype(np.int64)
        self.own_begin = np.maximum(centres - half, 0).astype(np.int64)
        self.own_end = np.minimum(centres + half, limit).astype(np.int64)
        self.own_rejected = (centres + half <= 0) | (centres - half > limit)
        self.at_centre = grid == centres[:, None]
        self.t_end = np.minimum(starts + window, limit)
        kept = ~self.rejected
        self.valid = kept.sum(axis=1)
        self.centred = (kept & self.at_centre).sum(axis=1)
        self._cands = {}

    def tables(self, fields):
        '''(cand, win) int32 tables of svamd_scan_grid for this axis.'''
        n, count = self.grid.shape
        kept = ~self.rejected
        centred = kept & self.at_centre
        cand = np.zeros((n, 1 + count, 5), dtype=np.int32)
        cand[:, 0, 0], cand[:, 0, 1], cand[:, 0, 2] = self.own_begin, self.own_end, self.own_rejected
        cand[:, 1:, 0], cand[:, 1:, 1] = self.begin, self.end
        cand[:, 1:, 2] = self.rejected + 2 * self.at_centre
        cand[:, 1:, 3] = np.cumsum(kept, axis=1) - kept
        cand[:, 1:, 4] = np.cumsum(centred, axis=1) - centred
        win = np.zeros((n, fields), dtype=np.int32)
        own_ok = ~self.own_rejected
        win[:, 0], win[:, 1], win[:, 2], win[:, 3], win[:, 4] = self.starts, self.t_end, own_ok, self.valid, self.centred
        if fields == 8:
            for k in (2, 3, 4):
                win[:, k + 3] = np.cumsum(win[:, k]) - win[:, k]
        return cand, win

    def own(self, i):
        '''(rejected, begin, end, centre) of window i's own position.'''
        return (bool(self.own_rejected[i]), int(self.own_begin[i]), int(self.own_end[i]), float(self.centres[i]))

    def candidates(self, i):
        '''Window i's surviving grid candidates as (centre, begin, end, is the own centre).'''
        got = self._cands.get(i)
        if got is None:
            c = float(self.centres[i])
            got = self._cands[i] = [
                (g, b, e, g == c) for g, b, e, rej in zip(self.grid[i].tolist(), self.begin[i].tolist(),
                                                          self.end[i].tolist(), self.rejected[i].tolist()) if not rej]
        return got


class _Stage0Grid:
    '''Geometry of a first stage: the per-axis candidates and what follows from them.'''

    def __init__(self, template_image, region_image, x_window, y_window, scheme, scheme_shift_size, window_overlap):
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
        self.template_image, self.region_image = template_image, region_image
        self.size = (x_window, y_window)
        self.ax = _Axis(np.arange(0, image_width, x_stride), x_window, x_span_hw, scheme[0], image_width)
        self.ay = _Axis(np.arange(0, image_height, y_stride), y_window, y_span_hw, scheme[1], image_height)
        self.ncol, self.nrow = len(self.ax.starts), len(self.ay.starts)
        self.nwin = self.ncol * self.nrow
        ax, ay = self.ax, self.ay
        own_x, own_y = ~ax.own_rejected, ~ay.own_rejected
        # pairs per window row, and each row's first pair in the whole stage
        per_row = (own_y * int(own_x.sum()) + ay.valid * int(ax.valid.sum()) - ay.centred * int(ax.centred.sum()))
        self.row_first = np.concatenate([[0], np.cumsum(per_row.astype(np.int64))])
        self.max_side = int(max(x_window, y_window, 1))

    def pair_count(self, row, column):
        ax, ay = self.ax, self.ay
        return int((not ax.own_rejected[column] and not ay.own_rejected[row]) +
                   ay.valid[row] * ax.valid[column] - ay.centred[row] * ax.centred[column])

    def row_chunks(self, row_begin, row_end, limit):
        '''[begin, end) runs of window rows whose pairs stay below `limit` (single rows may exceed it).'''
        first = self.row_first
        chunks, begin = [], row_begin
        while begin < row_end:
            end = int(np.searchsorted(first, first[begin] + limit, side='right')) - 1
            end = min(max(end, begin + 1), row_end)
            chunks.append((begin, end))
            begin = end
        return chunks

    def tables(self):
        '''The per-axis tables of svamd_scan_grid as numpy arrays.'''
        xcand, xwin = self.ax.tables(8)
        ycand, ywin = self.ay.tables(5)
        return dict(row_first=self.row_first, xcand=xcand, xwin=xwin, ycand=ycand, ywin=ywin)

    def run(self, left_dev, right_dev, threshold):
        '''Every window of the stage through svamd_scan_grid.  Returns per window the selected
        candidate slot (-1: none) and its correlation, as numpy arrays.  With an initialised
        torch.distributed process group every rank takes a band of window rows and the
        per-window results are all-gathered.'''
        rank, size = sharding.world()
        sharded = size > 1 and self.nrow >= size
        row_begin, row_end = sharding.shard_range(self.nrow, rank, size) if sharded else (0, self.nrow)
        slot, val = _scan_grid_rows(left_dev, right_dev, self, self.tables(), row_begin, row_end, threshold)
        if sharded:
            # shares are whole window rows: gathered as rows of ncol windows
            slot = sharding.all_gather_concat(slot.reshape(-1, self.ncol), self.nrow).ravel()
            val = sharding.all_gather_concat(val.reshape(-1, self.ncol), self.nrow).ravel()
        return slot, val

    def positions(self, slot):
        '''Per window: the centre of the selected candidate (the window's own centre where
        nothing was selected), as two float64 arrays.'''
        nx = self.ax.count
        column = np.tile(np.arange(self.ncol), self.nrow)
        row = np.repeat(np.arange(self.nrow), self.ncol)
        on_grid = slot > 0
        g = np.where(on_grid, slot - 1, 0)
        pos_x = np.where(on_grid, self.ax.grid[column, g % nx], self.ax.centres[column])
        pos_y = np.where(on_grid, self.ay.grid[row, g // nx], self.ay.centres[row])
        return pos_x, pos_y


class _Stage0Windows(Mapping):
    '''The dict of first-stage windows that sequence_scan returns, built window by window
    when it is read.  A dense scan has millions of windows; the scan itself works from
    arrays, and a caller that walks the result pays for the dicts it looks at.  Behaves as
    the dict it stands for (ids 0 .. n - 1 in order); a window, once built, is kept, so
    that later stages and callers see each other's changes to it as they would in a dict.'''

    def __init__(self, grid, dp_x, dp_y):
        self._grid, self._dp_x, self._dp_y, self._built = grid, dp_x, dp_y, {}

    def __len__(self):
        return self._grid.nwin

    def __iter__(self):
        return iter(range(self._grid.nwin))

    def __contains__(self, window_id):
        return isinstance(window_id, (int, np.integer)) and 0 <= window_id < self._grid.nwin

    def __getitem__(self, window_id):
        if window_id not in self:
            raise KeyError(window_id)
        window_id = int(window_id)
        info = self._built.get(window_id)
        if info is None:
            g = self._grid
            row, column = divmod(window_id, g.ncol)
            x, y = int(g.ax.starts[column]), int(g.ay.starts[row])
            own_x, own_y = g.ax.own(column), g.ay.own(row)
            centre = (own_x[3], own_y[3])
            info = self._built[window_id] = {
                'row': row,
                'column': column,
                'centre': centre,
                'size': g.size,
                'image_slice': g.template_image[y:y + g.size[1], x:x + g.size[0]],
                'target_regions': _Stage0Regions(g.region_image, own_x, own_y, g.ax.candidates(column),
                                                 g.ay.candidates(row), g.pair_count(row, column)),
                'stage_centres': [centre],
                'stage_sizes': [g.size],
                'dp_x': float(self._dp_x[window_id]) if self._dp_x is not None else None,
                'dp_y': float(self._dp_y[window_id]) if self._dp_y is not None else None,
            }
        return info

    def moved(self):
        '''ids of the windows with a displacement, ascending.'''
        return np.flatnonzero((self._dp_x != 0) | (self._dp_y != 0)).tolist()

    def __repr__(self):
        return repr(dict(self.items()))


def _scan_grid_rows(left_dev, right_dev, grid, tables, row_begin, row_end, threshold):
    '''Window rows [row_begin, row_end) of a first stage on the GPU: the tables go up in one
    copy, the rows run through svamd_scan_grid in chunks of at most GRID_CHUNK_PAIRS pairs
    (one workspace, reused in stream order) and two small arrays come back: per window of
    those rows the selected candidate slot (-1: none) and its correlation.'''
    torch = nv.torch_cuda()
    if SCAN_PRECISION not in ('mixed', 'fp64', 'fp32'):
        raise ValueError(f'unknown SCAN_PRECISION {SCAN_PRECISION!r}')
    precision = {'mixed': nv.ACC_MIXED, 'fp64': nv.ACC_F64, 'fp32': nv.ACC_F32}[SCAN_PRECISION]
    row_first = tables['row_first']
    # one upload: the int64 row offsets first, then the four int32 tables, each 256-byte aligned
    parts = [row_first.view(np.int32)] + [tables[k].ravel() for k in ('xcand', 'xwin', 'ycand', 'ywin')]
    offsets = np.cumsum([0] + [(p.size + 63) // 64 * 64 for p in parts])
    packed = np.zeros(int(offsets[-1]), dtype=np.int32)
    for part, at in zip(parts, offsets):
        packed[at:at + part.size] = part
    d_tab = nv.to_device(packed)
    t_row, t_xc, t_xw, t_yc, t_yw = (nv._vp(d_tab.data_ptr() + 4 * int(at)) for at in offsets[:-1])
    chunks = grid.row_chunks(row_begin, row_end, GRID_CHUNK_PAIRS)
    lib = nv.load_library()
    nbytes = max((lib.svamd_scan_grid_workspace_bytes((e - b) * grid.ncol, int(row_first[e] - row_first[b]))
                  for b, e in chunks), default=0)
    ws = nv.workspace(nbytes)
    best_slot = nv.empty(grid.nwin, torch.int32)
    best_val = nv.empty(grid.nwin, torch.float64)
    stream = nv.stream_ptr()
    min_std = float(correlation_spectral.MIN_ST_DEV)
    min_diff = float(correlation_spectral.MIN_MEAN_DIFF)
    for begin, end in chunks:
        nv.call('svamd_scan_grid', nv.ptr(left_dev), left_dev.shape[1], nv.ptr(right_dev), right_dev.shape[1],
                t_xc, t_xw, grid.ncol, grid.ax.count, t_yc, t_yw, grid.nrow, grid.ay.count, t_row, begin, end,
                int(row_first[end] - row_first[begin]), grid.max_side, min_std, min_diff, float(threshold),
                float(CENTRE_TOL), precision, float(GUARD_BAND), nv.ptr(best_slot), nv.ptr(best_val), nv.ptr(ws),
                nbytes, stream)
    w0, w1 = row_begin * grid.ncol, row_end * grid.ncol
    return best_slot[w0:w1].cpu().numpy(), best_val[w0:w1].cpu().numpy()


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
            grid = _Stage0Grid(
                template_image=left_image,
                region_image=right_image,
                x_window=stage_config['window_width'],
                y_window=stage_config['window_height'],
                scheme=tuple(stage_config['scheme']),
                scheme_shift_size=tuple(stage_config['scheme_shift_size']),
                window_overlap=stage_config['window_overlap'],
            )
            threshold = stage_config['correlation_threshold']
            if batched and grid.max_side <= nv.PAIR_MAX_SIDE:
                slot, best_val = grid.run(*resident(), threshold)
                pos_x, pos_y = grid.positions(slot)
                dp_x = pos_x - np.tile(grid.ax.centres, grid.nrow)
                dp_y = pos_y - np.repeat(grid.ay.centres, grid.ncol)
                windows = _Stage0Windows(grid, dp_x, dp_y)
                log_all = LOG_WINDOW_SUMMARY == True          # noqa: E712 (the reference compares this way)
                if LOG_WINDOW_SUMMARY:
                    for window_id in (range(grid.nwin) if log_all else windows.moved()):
                        row, column = divmod(window_id, grid.ncol)
                        centre = (float(grid.ax.centres[column]), float(grid.ay.centres[row]))
                        if slot[window_id] >= 0:
                            pos, val = (float(pos_x[window_id]), float(pos_y[window_id])), float(best_val[window_id])
                        else:
                            pos, val = centre, 0
                        _log_summary(window_id, centre, val, pos)
            else:
                windows = _stage0_windows(left_image, right_image, *grid.size, tuple(stage_config['scheme']),
                                          tuple(stage_config['scheme_shift_size']),
                                          stage_config['window_overlap'])[0]
                for window_id, window_info in windows.items():
                    window_info['stage_centres'] = [window_info['centre']]
                    window_info['stage_sizes'] = [window_info['size']]
                    pos, _ = _image_scan_pairwise(window_info, window_id, threshold)
                    window_info['dp_x'] = pos[0] - window_info['centre'][0]
                    window_info['dp_y'] = pos[1] - window_info['centre'][1]
        else:
            factor = stage_config['factor']
            region_boxes = {}
            previous = seq_results[stage - 1]
            # windows without a displacement in the previous stage drop out (a first stage
            # knows them from its arrays, without building the others)
            candidates = previous.moved() if isinstance(previous, _Stage0Windows) else previous
            for window_id in candidates:
                window_info = previous[window_id]
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
