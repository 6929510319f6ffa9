'''Image helpers, drop-in for the reference module of the same name (reference
utils.py:1-35).  Only `max_pos` belongs to the accelerated path; the two image
loaders run once per image on the host.
'''
import math

import numpy as np

import _native as nv


def greyscale_with_nan(path):
    '''RGBA image -> float64 greyscale with fully transparent pixels set to NaN
    (reference utils.py:6-15).'''
    from PIL import Image
    rgba = np.asarray(Image.open(path))
    grey = np.asarray(Image.open(path).convert('L')).astype('float64')
    alpha = rgba[..., 3]
    grey[alpha == 0] = np.nan
    return grey


def max_pos(array: np.ndarray, step_x=1, step_y=1):
    '''(x, y) position of the first maximum of a 2D array, NaNs ignored, scaled
    by the steps exactly as the reference does (utils.py:17-26: the column is
    multiplied by step_y and the row by step_x).'''
    array = np.asarray(array)
    torch = nv.torch_cuda()
    data = nv.to_device(np.ascontiguousarray(array, dtype=np.float64).ravel())
    idx = nv.empty(1, torch.int64)
    nv.call('svamd_nanargmax', nv.ptr(data), data.numel(), nv.ptr(idx), nv.stream_ptr())
    max_index = int(idx.item())
    if max_index < 0:
        raise ValueError('All-NaN slice encountered')
    return (
        step_y * (max_index % array.shape[1]),
        step_x * (max_index // array.shape[1]),
    )


def downsample(image, factor=1):
    '''Downsamples an image by a factor with scikit-image's `resize` (reference
    utils.py:28-35).  scikit-image is an optional dependency; without it only
    factor 1 is served, as the value rescaling `resize` applies to integer
    images (division by the dtype's maximum).'''
    shape = (math.floor(image.shape[0] / factor), math.floor(image.shape[1] / factor))
    try:
        from skimage.transform import resize
    except ImportError:
        if factor != 1:
            raise
        image = np.asarray(image)
        if image.dtype == np.uint8:
            return image / 255.0
        return image.astype(np.float64)
    return resize(image, shape)
