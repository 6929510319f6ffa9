'''Calibration-dot detection on MI355X: the peak finder of the reference's
calibration module (reference sv_calibration.py:115-141, 172-174).  The
correlation of the image with a Gaussian dot template and the neighbourhood
maximum / minimum filters run in libsvamd; the few surviving pixels are grouped
on the host with scipy.ndimage, as in the reference.
'''
import numpy as np
from scipy import ndimage

import _native as nv
from correlation_spectral import spectral_surface_on_device


def gaussian(x, y):
    '''exp(-(x^2 + y^2)) (reference sv_calibration.py:172-174).'''
    exponent = -(np.power(x, 2) + np.power(y, 2))
    return np.exp(exponent)


def peak_mask(data, neighborhood_size, threshold):
    '''Boolean mask of the pixels that equal their neighbourhood's maximum where
    that maximum exceeds the neighbourhood's minimum by more than `threshold`
    (reference sv_calibration.py:125-129).  data: host array or device tensor.'''
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
    return mask.cpu().numpy().astype(bool)


def find_peaks(region, neighborhood_size, threshold):
    '''Positions (x array, y array) of the dots of a calibration image: maxima
    of its circular correlation with a 101 x 101 Gaussian template, one per
    connected group of maximal pixels, shifted by half the template's size
    (reference sv_calibration.py:115-141).'''
    x_grid, y_grid = np.meshgrid(np.linspace(-5, 5, 101), np.linspace(-5, 5, 101))
    template = gaussian(x_grid, y_grid)
    data = spectral_surface_on_device(template, region)      # stays on the device
    maxima = peak_mask(data, neighborhood_size, threshold)
    labeled, num_objects = ndimage.label(maxima)
    slices = ndimage.find_objects(labeled)
    x, y = [], []
    for dy, dx in slices:
        x.append((dx.start + dx.stop - 1) / 2)
        y.append((dy.start + dy.stop - 1) / 2)
    x = np.add(x, template.shape[1] / 2)
    y = np.add(y, template.shape[0] / 2)
    return x, y
