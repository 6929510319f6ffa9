'''CPU-only checks of the C-ABI boundary: the shared library loads, exports every
symbol include/svamd.h declares, and the ctypes table covers the same set.'''
import ctypes
import os
import re

import pytest

from helpers import PKG, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'svamd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(svamd_\w+)\s*\(', text)))


def test_header_declares_expected_entry_points():
    names = declared_symbols()
    for want in ('svamd_cross_correlate_2d', 'svamd_cross_correlate_2d_spectral',
                 'svamd_cross_correlate_1d', 'svamd_cross_correlate_1d_spectral',
                 'svamd_scan_pairs', 'svamd_scan_fold', 'svamd_band_scan_u8', 'svamd_sat2d',
                 'svamd_nanargmax'):
        assert want in names


def test_library_exports_every_declared_symbol():
    path = os.path.join(PKG, 'libsvamd.so')
    assert os.path.exists(path), 'libsvamd.so not built (make -C stereo-vision_amd)'
    import torch  # noqa: F401  (first, as a host must: the library then binds to the HIP runtime PyTorch ships)
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in svamd.h but not exported'
    lib.svamd_version.restype = ctypes.c_int
    assert lib.svamd_version() >= 100


def test_ctypes_table_matches_header():
    import _native
    assert sorted(_native.PROTOTYPES) == declared_symbols()
    _native.load_library()


def test_argument_validation_needs_no_gpu():
    '''Shape / argument errors are reported before any device work.'''
    import _native
    lib = _native.load_library()
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    rc = lib.svamd_cross_correlate_2d(one, 5, 5, one, _native.U8, 4, 9, 1, 1, one, one, 1 << 20, None)
    assert rc == _native.ERR_SHAPE
    assert b'must not exceed' in lib.svamd_last_error()
    rc = lib.svamd_band_scan_u8(one, one, 64, 64, 31, 31, 2, 1, 8, 0, 0.0, 0.01, 0, 1, 0, one, one, None,
                                one, 1 << 30, None)
    assert rc == _native.ERR_LIMIT
    rc = lib.svamd_scan_pairs(one, one, one, 4, 65, 1e-4, 1e-9, 0, one, None, None, None, None)
    assert rc == _native.ERR_LIMIT
    assert lib.svamd_cross_correlate_2d_workspace_bytes(31, 31, 64, 64) > 0
    # shape queries are host arithmetic
    assert lib.svamd_band_box_sums_supported(31, 31, 1, 128) == 1
    assert lib.svamd_band_box_sums_supported(5, 200, 1, 6) == 0        # the tile's rows do not fit the LDS
    assert lib.svamd_band_box_sums_supported(31, 31, 1, 40000) == 0    # nor does a band this wide
    assert lib.svamd_band_box_sums_supported(40, 40, 1, 128) == 1      # larger windows than the multiply-add kernel takes
    assert lib.svamd_band_box_sums_supported(80, 80, 1, 8) == 0        # beyond the tables' window area


def test_host_modules_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import numpy as np
    import correlation
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        correlation.cross_correlate_2d(np.zeros((3, 3)), np.zeros((8, 8), np.uint8))
