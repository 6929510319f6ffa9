'''ctypes binding of libsvamd.so (the HIP kernels behind the C ABI declared in
include/svamd.h) plus the device-memory plumbing the host modules share.

PyTorch is used only as an allocator / stream provider: arrays go to the GPU as
torch tensors and their data pointers are handed to the C entry points.  There
is no CPU fallback: importing this module without the built library, or calling
into it without a GPU, raises.
'''
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVAMD_LIBRARY selects another build of the library (A/B measurements)
LIB_PATH = os.environ.get('SVAMD_LIBRARY') or os.path.join(_HERE, 'libsvamd.so')

OK, ERR_INVALID, ERR_SHAPE, ERR_WORKSPACE, ERR_HIP, ERR_LIMIT = range(6)
U8, F32, F64 = 0, 1, 2
ACC_F32, ACC_F64, ACC_MIXED = 0, 1, 2
BAND_ARGMAX, BAND_FOLD = 0, 1
BAND_NO_MATCH = -2 ** 31
BAND_PREPARED = 1
BAND_BOX_SUMS = 2
PAIR_MAX_SIDE = 64

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_sz = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/svamd.h declares
PROTOTYPES = {
    'svamd_version': (_i, []),
    'svamd_last_error': (ctypes.c_char_p, []),
    'svamd_sat2d': (_i, [_vp, _i, _i, _i, _i64, _d, _vp, _vp, _vp]),
    'svamd_nanargmax': (_i, [_vp, _i64, _vp, _vp]),
    'svamd_cross_correlate_2d_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'svamd_cross_correlate_2d': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'svamd_cross_correlate_2d_spectral_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'svamd_cross_correlate_2d_spectral': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _d, _d, _vp, _vp, _sz, _vp]),
    'svamd_fft_supported': (_i, [_i, _i]),
    'svamd_xcorr2d_fft_workspace_bytes': (_sz, [_i, _i, _i]),
    'svamd_xcorr2d_fft': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _d, _d, _vp, _i, _vp, _sz, _vp]),
    'svamd_cross_correlate_1d_workspace_bytes': (_sz, [_i64]),
    'svamd_cross_correlate_1d': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _sz, _vp]),
    'svamd_cross_correlate_1d_static': (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    'svamd_cross_correlate_1d_spectral_workspace_bytes': (_sz, [_i64, _i64]),
    'svamd_cross_correlate_1d_spectral': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    'svamd_scan_pairs': (_i, [_vp, _vp, _vp, _i64, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp]),
    'svamd_scan_fold': (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _vp, _vp, _vp, _vp]),
    'svamd_scan_grid_workspace_bytes': (_sz, [_i64, _i64]),
    'svamd_scan_grid': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i64,
                             _i, _d, _d, _d, _d, _i, _d, _vp, _vp, _vp, _sz, _vp]),
    'svamd_band_scan_u8_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'svamd_band_box_sums_supported': (_i, [_i, _i, _i, _i]),
    'svamd_band_prepare_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'svamd_band_scan_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _i, _i, _i,
                                _vp, _vp, _vp, _vp, _sz, _vp]),
    'svamd_band_scan_f64_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'svamd_band_scan_f64_recheck_count_offset': (_sz, [_i, _i, _i, _i]),
    'svamd_band_prepare_f64': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'svamd_band_scan_f64': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _d, _i, _i, _i,
                                 _vp, _vp, _vp, _vp, _sz, _vp]),
    'svamd_accumulate_boxes_workspace_bytes': (_sz, [_i64, _i, _i, _i, _i]),
    'svamd_accumulate_boxes': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'svamd_band_accumulate_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'svamd_band_accumulate': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'svamd_grid_average': (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    'svamd_peak_mask_workspace_bytes': (_sz, [_i, _i]),
    'svamd_peak_mask': (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _sz, _vp]),
    'svamd_greyscale': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'svamd_downsample_workspace_bytes': (_sz, [_i, _i]),
    'svamd_downsample': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
}

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def load_library():
    '''Loads libsvamd.so and declares the prototypes.  Needs no GPU.'''
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `make -C {_HERE}` '
                '(hipcc --offload-arch=gfx950); there is no CPU fallback')
        # libsvamd.so asks the loader for "libamdhip64.so.7".  PyTorch-ROCm wheels bundle a
        # copy of the HIP runtime whose file is called libamdhip64.so but whose SONAME is
        # libamdhip64.so.7, so with PyTorch imported first (a plain import: no GPU call is
        # needed) the loader hands the library that already-mapped copy and the process
        # holds ONE runtime - streams, events and pointers pass freely between PyTorch and
        # the library.  Loaded the other way round the library pulls in the system's
        # runtime by its RUNPATH, PyTorch then maps its own file next to it, two HSA
        # runtimes open the device and the library's first launch fails with "no
        # ROCm-capable device is detected".
        import torch        # noqa: F401  (before the library, for the reason above)
        lib = ctypes.CDLL(LIB_PATH)
        _require_single_runtime()
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def hip_runtimes():
    '''Paths of the HIP / HSA runtime files mapped into this process.'''
    found = set()
    try:
        with open('/proc/self/maps') as maps:
            for line in maps:
                path = line.rsplit(None, 1)[-1]
                if 'libamdhip64' in path or 'libhsa-runtime64' in path:
                    found.add(path)
    except OSError:         # pragma: no cover
        pass
    return sorted(found)


def _require_single_runtime():
    hip = [p for p in hip_runtimes() if 'libamdhip64' in p]
    if len(hip) > 1:
        raise RuntimeError(
            'two HIP runtimes are mapped into this process (' + ', '.join(hip) + '): libsvamd.so or another '
            'HIP library was loaded before PyTorch.  Import torch (or this package) first, so that every '
            'library shares the runtime PyTorch ships; handles cannot cross between two runtimes')


def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('no HIP device visible: the NCC kernels need an AMD GPU '
                           '(gfx950); there is no CPU fallback')
    return torch


def call(name, *args):
    lib = load_library()
    rc = getattr(lib, name)(*args)
    if rc != OK:
        raise NativeError(rc, lib.svamd_last_error().decode())


def stream_ptr():
    torch = torch_cuda()
    return _vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def dtype_code(arr):
    if arr.dtype == np.uint8:
        return U8
    if arr.dtype == np.float32:
        return F32
    if arr.dtype == np.float64:
        return F64
    raise TypeError(f'unsupported dtype {arr.dtype}')


def as_image(arr):
    '''numpy array in one of the element types the kernels read (uint8, float32,
    float64), C-contiguous; other real dtypes are widened to float64.'''
    arr = np.asarray(arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype not in (np.uint8, np.float32, np.float64):
        arr = arr.astype(np.float64)
    return np.ascontiguousarray(arr)


def to_device(arr):
    torch = torch_cuda()
    arr = np.ascontiguousarray(arr)
    if not arr.flags.writeable:      # e.g. arrays backed by a PIL image buffer
        arr = arr.copy()
    return torch.from_numpy(arr).cuda()


def empty(shape, dtype):
    torch = torch_cuda()
    return torch.empty(shape, dtype=dtype, device='cuda')


def workspace(nbytes):
    torch = torch_cuda()
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device='cuda')
