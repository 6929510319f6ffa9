'''Image helpers, drop-in for the reference module of the same name (reference
utils.py:1-35).  Files are decoded on the host (PIL); the grey-level conversion,
the transparency mask and the resizing run on the device, and the images they
return keep their device copy so the scans that follow do not upload them again.
'''
import math

import numpy as np

import _native as nv


class ResidentImage(np.ndarray):
    '''A host image that remembers its float64 twin in GPU memory (`resident`),
    so that the scans need not upload it again.  It is read-only: edit a copy
    (`np.array(image)`), which is an ordinary array without a twin, or call
    `image.setflags(write=True)`, which drops the twin.  Slices and
    results of arithmetic are ordinary in that sense too.'''

    def __new__(cls, host, resident):
        obj = np.asarray(host).view(cls)
        obj.resident = resident
        obj.flags.writeable = False
        return obj

    def __array_finalize__(self, obj):
        self.resident = None

    def setflags(self, write=None, align=None, uic=None):
        '''Making the image writable (`image.setflags(write=True)`) gives up the
        device twin first: an edited host image must be uploaded again.'''
        if write:
            self.resident = None
        super().setflags(write=write, align=align, uic=uic)


def resident_float64(image):
    '''float64 device copy of a host image: the twin a ResidentImage carries, or
    a fresh upload.'''
    twin = getattr(image, 'resident', None)
    # a twin is trusted only while the host array cannot have been edited: numpy lets a
    # caller flip `image.flags.writeable` without going through setflags()
    if twin is not None and not image.flags.writeable and tuple(twin.shape) == tuple(np.shape(image)):
        return twin
    return nv.to_device(np.ascontiguousarray(image, dtype=np.float64))


def _greyscale_on_device(pixels, nan_alpha, want_u8):
    '''pixels: H x W (x channels) uint8 on the host.  Returns the device grey
    image: uint8 when want_u8, else float64 (NaN where alpha is 0 if asked).'''
    torch = nv.torch_cuda()
    pixels = np.ascontiguousarray(pixels)
    if pixels.dtype != np.uint8:
        raise TypeError(f'8-bit images only, got {pixels.dtype}')
    channels = 1 if pixels.ndim == 2 else pixels.shape[2]
    height, width = pixels.shape[:2]
    d_pix = nv.to_device(pixels)
    out = nv.empty((height, width), torch.uint8 if want_u8 else torch.float64)
    nv.call('svamd_greyscale', nv.ptr(d_pix), channels, height, width, int(nan_alpha),
            nv.ptr(out) if want_u8 else None, None if want_u8 else nv.ptr(out), nv.stream_ptr())
    return out


def greyscale_with_nan(path):
    '''RGBA image -> float64 greyscale with fully transparent pixels set to NaN
    (reference utils.py:6-15).  The file is decoded on the host; the conversion
    runs on the device and the result stays resident there as well.'''
    from PIL import Image
    rgba = np.asarray(Image.open(path))
    if rgba.ndim != 3 or rgba.shape[2] != 4:
        # the reference unpacks four channels (utils.py:13)
        raise ValueError(f'expected an RGBA image, got an array of shape {rgba.shape}')
    grey = _greyscale_on_device(rgba, nan_alpha=True, want_u8=False)
    return ResidentImage(grey.cpu().numpy(), grey)


def greyscale(path):
    '''`np.asarray(Image.open(path).convert('L'))` of the reference's drivers
    (e.g. test_image_comparison.py:36-37) for 8-bit grey, grey + alpha, RGB and
    RGBA files, converted on the device.  Other pixel formats take PIL's own
    conversion on the host.'''
    from PIL import Image
    image = Image.open(path)
    if image.mode not in ('L', 'LA', 'RGB', 'RGBA'):
        return np.asarray(image.convert('L'))
    return _greyscale_on_device(np.asarray(image), nan_alpha=False, want_u8=True).cpu().numpy()


def max_pos(array, step_x=1, step_y=1):
    '''(x, y) position of the first maximum of a 2D array, NaNs ignored, scaled
    by the steps exactly as the reference does (utils.py:17-26: the column is
    multiplied by step_y and the row by step_x).  A surface that is still on the
    device (a torch tensor) is reduced there and one integer comes back; a host
    array is not worth the round trip and is reduced where it is.'''
    torch = nv.torch_cuda()
    if isinstance(array, torch.Tensor):
        data = array.to(device='cuda', dtype=torch.float64).contiguous()
        idx = nv.empty(1, torch.int64)
        nv.call('svamd_nanargmax', nv.ptr(data), data.numel(), nv.ptr(idx), nv.stream_ptr())
        max_index = int(idx.item())
        if max_index < 0:
            raise ValueError('All-NaN slice encountered')
        width = data.shape[1]
    else:
        array = np.asarray(array)
        max_index = int(np.nanargmax(array))          # raises ValueError when every element is NaN
        width = array.shape[1]
    return (
        step_y * (max_index % width),
        step_x * (max_index // width),
    )


def downsample(image, factor=1):
    '''Downsamples an image by a factor the way the reference does with
    scikit-image's `resize` (reference utils.py:28-35), on the device: integer
    images are scaled to 0..1, smoothed by a Gaussian of sigma (factor - 1) / 2
    and sampled linearly.  The float64 result stays resident on the device too.'''
    torch = nv.torch_cuda()
    image = np.asarray(image) if not isinstance(image, np.ndarray) else image
    if image.ndim != 2:
        raise ValueError('downsample serves 2D (single-channel) images')
    height, width = image.shape
    out_h, out_w = math.floor(height / factor), math.floor(width / factor)
    twin = getattr(image, 'resident', None)
    if twin is not None and not image.flags.writeable and tuple(twin.shape) == (height, width):
        d_in, code = twin, nv.F64
    else:
        host = np.asarray(image)
        if host.dtype == np.bool_:
            raise ValueError('boolean images are not resized with anti-aliasing')
        if host.dtype != np.uint8:
            if host.dtype.kind != 'f':
                raise TypeError(f'uint8 or floating-point images only, got {host.dtype}')
            host = host.astype(np.float64, copy=False)
        d_in, code = nv.to_device(host), nv.dtype_code(host)
    out = nv.empty((out_h, out_w), torch.float64)
    nbytes = nv.load_library().svamd_downsample_workspace_bytes(height, width)
    ws = nv.workspace(nbytes)
    nv.call('svamd_downsample', nv.ptr(d_in), code, height, width, out_h, out_w, nv.ptr(out), nv.ptr(ws),
            nbytes, nv.stream_ptr())
    return ResidentImage(out.cpu().numpy(), out)
