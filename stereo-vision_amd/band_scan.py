'''Dense horizontal-band stereo scan on MI355X.

The reference scans a stereo pair by calling its correlator once per (window,
candidate region) pair from Python.  For the dense case - a window at every
pixel and a band of candidate offsets on the same image rows - libsvamd
evaluates all of it in one kernel (svamd_band_scan_u8): the spatial NCC of
reference correlation.py:104-116 at full overlap, then per window either the
arg-max of utils.max_pos (utils.py:17-26) or image_scan's selection rule
(sv_image_comparison.py:249-255).  8-bit images (uint8, or 8-bit levels scaled
to 0..1) take the integer-exact kernel; any other floating-point image - what
the reference's downsample() returns for factors above 1 - takes
svamd_band_scan_f64 (float32 products of centred pixels, every near-tie
re-evaluated in float64).
'''
import numpy as np

import _native as nv
import sharding
import sv_image_comparison as _svic

# Floating-point images: decisions (comparisons of a score with the threshold or with
# the best score so far) closer than this are repeated in float64.  The float32 scores
# of the band kernel are products of centred pixels and stay within a few 1e-6 of the
# float64 ones; evaluations that cannot promise that are sent to float64 regardless.
BAND_GUARD_BAND = 1e-5

WINDOW_WIDTHS = tuple(range(3, 33))   # widths with a fully unrolled kernel; others take the run-time-width one


def as_uint8_image(image):
    '''uint8 view of an image given either as uint8 or as uint8 / 255 floats (what
    the stereo driver feeds sequence_scan); anything else is rejected, because
    the band kernel's exactness rests on 8-bit integer pixels.'''
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return np.ascontiguousarray(image)
    if image.dtype.kind == 'f':
        q = np.rint(image * 255.0)
        # 8-bit levels scaled either way: value / 255 or value * (1 / 255), which is
        # what `utils.downsample(image, 1)` (scikit-image's float conversion) yields
        if q.min() >= 0 and q.max() <= 255 and (np.array_equal(q / 255.0, image) or
                                                 np.array_equal(q * (1.0 / 255), image)):
            return np.ascontiguousarray(q.astype(np.uint8))
    raise TypeError('not an 8-bit image (uint8, or uint8 / 255 as float)')


def _as_float_image(image):
    image = np.asarray(image)
    if image.dtype.kind not in 'fiub':
        raise TypeError(f'band_scan needs real-valued images, got {image.dtype}')
    return np.ascontiguousarray(image, dtype=np.float64)


class BandScanner:
    '''Device-resident state for repeated scans of same-sized pairs: images,
    workspace and outputs are allocated once (per pixel type: a scanner that
    is handed floating-point images after 8-bit ones holds both sets).'''

    def __init__(self, height, width, window_width, window_height, max_disparity, row_stride=1,
                 want_scores=False):
        torch = nv.torch_cuda()
        self.shape = (height, width)
        self.win = (window_width, window_height)
        self.max_disp = int(max_disparity)
        self.row_stride = int(row_stride)
        self.nwx = width - window_width + 1
        self.nwy = (height - window_height) // self.row_stride + 1
        if self.nwx <= 0 or self.nwy <= 0:
            raise Exception('Dimensions of template must not exceed those of region.')
        self.kind = 'u8'
        self._sets = {}
        self._buffers('u8')
        self.disp = nv.empty((self.nwy, self.nwx), torch.int32)
        self.val = nv.empty((self.nwy, self.nwx), torch.float64)
        self.scores = (nv.empty((self.nwy, self.nwx, 2 * self.max_disp + 1), torch.float64)
                       if want_scores else None)

    def _buffers(self, kind):
        '''(left, right, workspace) on the device for 'u8' or 'f64' pixels.'''
        if kind not in self._sets:
            torch = nv.torch_cuda()
            lib = nv.load_library()
            sizer = lib.svamd_band_scan_u8_workspace_bytes if kind == 'u8' else lib.svamd_band_scan_f64_workspace_bytes
            dtype = torch.uint8 if kind == 'u8' else torch.float64
            self._sets[kind] = (nv.empty(self.shape, dtype), nv.empty(self.shape, dtype),
                                nv.workspace(sizer(*self.shape, *self.win)))
        self.left, self.right, self.ws = self._sets[kind]
        self.kind = kind

    def upload(self, left, right):
        torch = nv.torch_cuda()
        try:
            left, right, kind = as_uint8_image(left), as_uint8_image(right), 'u8'
        except TypeError:
            left, right, kind = _as_float_image(left), _as_float_image(right), 'f64'
        self._buffers(kind)
        self.left.copy_(torch.from_numpy(left))
        self.right.copy_(torch.from_numpy(right))

    def prepare(self):
        '''Enqueues the summed-area tables / window statistics of the uploaded
        pair (kernel K1); run(prepared=True) then skips them.'''
        h, w = self.shape
        nv.call('svamd_band_prepare_u8' if self.kind == 'u8' else 'svamd_band_prepare_f64', nv.ptr(self.left),
                nv.ptr(self.right), h, w, self.win[0], self.win[1], nv.ptr(self.ws), self.ws.numel(),
                nv.stream_ptr())

    def rechecked(self):
        '''Floating-point images: how many windows of the last run() were evaluated again in
        float64 (0 for 8-bit images, whose first evaluation is exact).  Synchronises.'''
        if self.kind == 'u8':
            return 0
        torch = nv.torch_cuda()
        at = nv.load_library().svamd_band_scan_f64_recheck_count_offset(*self.shape, *self.win)
        return int(self.ws[at:at + 4].view(torch.int32).item())

    def box_sums_supported(self):
        '''Whether 8-bit scans of this geometry can take the box-sum kernel.'''
        return bool(nv.load_library().svamd_band_box_sums_supported(self.win[0], self.win[1], self.row_stride,
                                                                    self.max_disp))

    def run(self, correlation_threshold=None, rows=None, prepared=False, method='products'):
        '''Enqueues one scan on the current stream.  correlation_threshold None:
        plain arg-max per window; a number: image_scan's selection rule with that
        threshold and the module's CENTRE_TOL.  rows: (begin, end) window rows.
        method (8-bit images): 'products' - one multiply-add per window pixel and offset, the
        kernel the benchmark times; 'sums' - the same cross terms from per-offset summed-area
        tables of the product image (identical results, several times faster).'''
        h, w = self.shape
        mode = nv.BAND_ARGMAX if correlation_threshold is None else nv.BAND_FOLD
        begin, end = rows if rows is not None else (0, self.nwy)
        if method not in ('products', 'sums'):
            raise ValueError(f'unknown method {method!r}')
        flags = (nv.BAND_PREPARED if prepared else 0) | (nv.BAND_BOX_SUMS if method == 'sums' and self.kind == 'u8' else 0)
        head = (nv.ptr(self.left), nv.ptr(self.right), h, w, self.win[0], self.win[1], 1, self.row_stride,
                self.max_disp, mode, float(correlation_threshold or 0.0), float(_svic.CENTRE_TOL))
        tail = (int(begin), int(end), flags, nv.ptr(self.disp), nv.ptr(self.val),
                nv.ptr(self.scores), nv.ptr(self.ws), self.ws.numel(), nv.stream_ptr())
        if self.kind == 'u8':
            nv.call('svamd_band_scan_u8', *head, *tail)
        else:
            # float32 products; decisions within the scan's guard band are repeated in float64
            guard = 0.0 if _svic.SCAN_PRECISION == 'fp32' else float(BAND_GUARD_BAND)
            nv.call('svamd_band_scan_f64', *head, guard, *tail)


def band_scan(left_image, right_image, window_width, window_height, max_disparity,
              correlation_threshold=None, row_stride=1, want_scores=False, method='products', column_stride=1):
    '''Scans every window_width x window_height window of the left image against
    the right image at horizontal offsets -max_disparity..+max_disparity.
    Returns (disparity int32 [rows, cols], correlation float64 [rows, cols]) and,
    if asked, the full score volume [rows, cols, 2*max_disparity+1] (NaN where a
    candidate box leaves the image or a window is constant).  column_stride > 1 keeps the
    windows at every column_stride-th column: the kernels scan every column (their
    neighbouring windows share their work) and the others are dropped here.'''
    if int(column_stride) < 1:
        raise ValueError('column_stride must be a positive integer')
    if int(column_stride) > 1:
        out = band_scan(left_image, right_image, window_width, window_height, max_disparity,
                        correlation_threshold, row_stride, want_scores, method)
        return tuple(np.ascontiguousarray(a[:, ::int(column_stride)]) for a in out)
    left, right = np.asarray(left_image), np.asarray(right_image)
    if left.shape != right.shape:
        raise Exception('Dimensions of template and region images must match')
    scanner = BandScanner(left.shape[0], left.shape[1], window_width, window_height, max_disparity,
                          row_stride, want_scores)
    scanner.upload(left, right)
    rank, size = sharding.world()
    if size == 1:
        scanner.run(correlation_threshold, method=method)
        out = (scanner.disp.cpu().numpy(), scanner.val.cpu().numpy())
        if want_scores:
            out += (scanner.scores.cpu().numpy(),)
        return out
    # one band of window rows per rank, then an all-gather of the results
    begin, end = sharding.shard_range(scanner.nwy, rank, size)
    scanner.run(correlation_threshold, rows=(begin, end), method=method)
    out = (sharding.all_gather_concat(scanner.disp[begin:end].cpu().numpy(), scanner.nwy),
           sharding.all_gather_concat(scanner.val[begin:end].cpu().numpy(), scanner.nwy))
    if want_scores:
        out += (sharding.all_gather_concat(scanner.scores[begin:end].cpu().numpy(), scanner.nwy),)
    return out
