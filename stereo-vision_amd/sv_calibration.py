'''Calibration-dot detection on MI355X: the peak finder of the reference's
calibration module (reference sv_calibration.py:115-141, 172-174).  The
correlation of the image with a Gaussian dot template and the neighbourhood
maximum / minimum filters run in libsvamd; only the coordinates of the few
surviving pixels come back to the host, where they are grouped into dots.
'''
import numpy as np

import _native as nv
from correlation_spectral import spectral_surface_on_device


def gaussian(x, y):
    '''The dot model exp(-(x^2 + y^2)) (reference sv_calibration.py:172-174).'''
    return np.exp(-(np.square(x) + np.square(y)))


DOT_TEMPLATE_SIDE = 101         # samples of the dot model over [-5, 5] along each axis


def dot_template():
    axis = np.linspace(-5, 5, DOT_TEMPLATE_SIDE)
    return gaussian(axis[None, :], axis[:, None])


def peak_mask(data, neighborhood_size, threshold, on_device=False):
    '''Boolean mask of the pixels that equal their neighbourhood's maximum where
    that maximum exceeds the neighbourhood's minimum by more than `threshold`
    (reference sv_calibration.py:125-129).  data: host array or device tensor;
    on_device: return the uint8 mask as a device tensor instead of a host array.'''
    torch = nv.torch_cuda()
    if isinstance(data, torch.Tensor):
        d_data = data.to(device='cuda', dtype=torch.float64).contiguous()
    else:
        d_data = nv.to_device(np.ascontiguousarray(data, dtype=np.float64))
    h, w = d_data.shape
    mask = nv.empty((h, w), torch.uint8)
    ws = nv.workspace(nv.load_library().svamd_peak_mask_workspace_bytes(h, w))
    nv.call('svamd_peak_mask', nv.ptr(d_data), h, w, int(neighborhood_size), float(threshold), nv.ptr(mask),
            nv.ptr(ws), ws.numel(), nv.stream_ptr())
    return mask if on_device else mask.cpu().numpy().astype(bool)


def group_boxes(rows, cols):
    '''Bounding boxes of the 4-connected groups of the pixels (rows[i], cols[i]),
    given in raster order: (row_min, row_max, col_min, col_max) per group, groups
    ordered by their first pixel in raster order (the order in which
    scipy.ndimage.label numbers them).  The pixel list is short (a few per dot),
    so a union-find over it beats labelling the whole image.'''
    index = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(rows, cols))}
    parent = list(range(len(index)))

    def root(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for (r, c), i in index.items():
        for neighbour in ((r - 1, c), (r, c - 1)):
            j = index.get(neighbour)
            if j is not None:
                a, b = root(i), root(j)
                if a != b:
                    parent[max(a, b)] = min(a, b)     # the raster-first pixel stays the root
    boxes = {}
    for (r, c), i in index.items():
        k = root(i)
        box = boxes.get(k)
        boxes[k] = (r, r, c, c) if box is None else (min(box[0], r), max(box[1], r), min(box[2], c), max(box[3], c))
    return [boxes[k] for k in sorted(boxes)]


def find_peaks(region, neighborhood_size, threshold):
    '''Positions (x array, y array) of the dots of a calibration image: maxima
    of its circular correlation with a 101 x 101 Gaussian template, one per
    connected group of maximal pixels (the centre of the group's bounding box),
    shifted by half the template's size (reference sv_calibration.py:115-141).
    Correlation and neighbourhood filters run on the device; only the coordinates
    of the surviving pixels come back.'''
    torch = nv.torch_cuda()
    template = dot_template()
    surface = spectral_surface_on_device(template, region)
    keep = torch.nonzero(peak_mask(surface, neighborhood_size, threshold, on_device=True))   # raster order
    rows, cols = keep[:, 0].cpu().numpy(), keep[:, 1].cpu().numpy()
    boxes = group_boxes(rows, cols)
    half_h, half_w = template.shape[0] / 2, template.shape[1] / 2
    x = np.array([(c0 + c1) / 2 for _, _, c0, c1 in boxes], dtype=np.float64) + half_w
    y = np.array([(r0 + r1) / 2 for r0, r1, _, _ in boxes], dtype=np.float64) + half_h
    return x, y
