'''Displacement / depth grids on MI355X: the accumulation the reference's stereo
driver performs after each sequence_scan (reference
test_image_comparison.py:49-98).  Every window paints its dp_x, dp_y, |dp| and a
contribution count over its box of four full-resolution grids; the sequences'
grids are summed and finally divided by the contribution counts.

The painting runs in libsvamd (svamd_accumulate_boxes for sequence_scan results,
svamd_band_accumulate for the dense band scan's disparity map,
svamd_grid_average for the final division); the host only lists the boxes.
'''
import numpy as np

import _native as nv


def sequence_boxes(seq_results, shape):
    '''The boxes and values one sequence paints, in the reference's visiting
    order (reference test_image_comparison.py:65-88): only the deepest stage a
    window reached contributes.  Returns (boxes int32 [n, 4] = x_start, x_end,
    y_start, y_end; values float64 [n, 3] = dp_x, dp_y, |dp|).'''
    image_height, image_width = shape
    boxes, values = [], []
    for stage, windows in enumerate(seq_results):
        for window_info in windows.values():
            reached = len(window_info['stage_centres']) - 1
            if reached > stage:
                continue
            x, y = window_info['stage_centres'][stage - 1 if reached < stage else stage]
            dp_x, dp_y = window_info['dp_x'], window_info['dp_y']
            x_window, y_window = window_info['stage_sizes'][stage]
            boxes.append((int(max(x - x_window / 2, 0)), int(min(x + x_window / 2, image_width)),
                          int(max(y - y_window / 2, 0)), int(min(y + y_window / 2, image_height))))
            values.append((dp_x, dp_y, np.sqrt(dp_x ** 2 + dp_y ** 2)))
    return (np.array(boxes, dtype=np.int32).reshape(-1, 4),
            np.array(values, dtype=np.float64).reshape(-1, 3))


class DepthGrids:
    '''Running totals of the four grids, resident on the device.'''

    def __init__(self, shape):
        torch = nv.torch_cuda()
        self.shape = (int(shape[0]), int(shape[1]))
        self.depth, self.dp_x, self.dp_y, self.contributions = (
            torch.zeros(self.shape, dtype=torch.float64, device='cuda') for _ in range(4))

    def add_boxes(self, boxes, values):
        '''Paints a box list (see sequence_boxes) and adds it to the totals.'''
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, 3)
        n = boxes.shape[0]
        if n != values.shape[0]:
            raise Exception('Numbers of boxes and values must match')
        if n == 0:
            return
        h, w = self.shape
        max_w = max(int((boxes[:, 1] - boxes[:, 0]).max()), 1)
        max_h = max(int((boxes[:, 3] - boxes[:, 2]).max()), 1)
        lib = nv.load_library()
        ws = nv.workspace(lib.svamd_accumulate_boxes_workspace_bytes(n, h, w, max_w, max_h))
        d_boxes, d_values = nv.to_device(boxes), nv.to_device(values)
        nv.call('svamd_accumulate_boxes', nv.ptr(d_boxes), nv.ptr(d_values), n, h, w, max_w, max_h,
                nv.ptr(self.depth), nv.ptr(self.dp_x), nv.ptr(self.dp_y), nv.ptr(self.contributions),
                nv.ptr(ws), ws.numel(), nv.stream_ptr())

    def add_sequence(self, seq_results):
        '''Adds the grids of one sequence_scan result (reference
        test_image_comparison.py:61-92).'''
        self.add_boxes(*sequence_boxes(seq_results, self.shape))

    def add_band_scan(self, disparity, window_width, window_height, row_stride=1):
        '''Adds the grids of a dense band scan: `disparity` is band_scan's int32
        map (numpy array or device tensor), one window per origin.'''
        torch = nv.torch_cuda()
        h, w = self.shape
        if not isinstance(disparity, torch.Tensor):
            disparity = nv.to_device(np.ascontiguousarray(disparity, dtype=np.int32))
        nwy, nwx = (h - window_height) // row_stride + 1, w - window_width + 1
        if tuple(disparity.shape) != (nwy, nwx) or disparity.dtype != torch.int32:
            raise Exception('Disparity map does not match the image and window sizes')
        lib = nv.load_library()
        ws = nv.workspace(lib.svamd_band_accumulate_workspace_bytes(h, w, window_height, row_stride))
        nv.call('svamd_band_accumulate', nv.ptr(disparity), h, w, window_width, window_height, row_stride,
                nv.ptr(self.depth), nv.ptr(self.dp_x), nv.ptr(self.contributions), nv.ptr(ws), ws.numel(),
                nv.stream_ptr())

    def average(self):
        '''Divides the totals by the contribution counts (0 -> infinity first,
        reference test_image_comparison.py:95-98) and returns the host arrays
        (depth, dp_x, dp_y, contributions).  The totals are consumed.'''
        nv.call('svamd_grid_average', nv.ptr(self.depth), nv.ptr(self.dp_x), nv.ptr(self.dp_y),
                nv.ptr(self.contributions), self.shape[0] * self.shape[1], nv.stream_ptr())
        return tuple(g.cpu().numpy() for g in (self.depth, self.dp_x, self.dp_y, self.contributions))
