'''Pins the CPU oracle to the reference: every oracle function is compared with
the golden vectors that tests/golden/make_golden.py produced by running the
reference itself.  CPU only.'''
import json

import numpy as np
import pytest

from helpers import golden, rand_pair

TIGHT = 1e-11   # float64 restatement vs float64 reference (different summation order)


def close(a, b, tol=TIGHT):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    assert np.all(np.abs(a[m] - b[m]) <= tol * np.maximum(1.0, np.abs(b[m])))


def test_g1_spatial_template_region(oracle):
    g = golden('g1_spatial_template_region')
    out = oracle.cross_correlate_2d(g['template'], g['region'], 5, 5)
    close(out, g['surface'])
    assert tuple(oracle.max_pos(out, 5, 5)) == tuple(g['max_pos']) == (290, 150)
    assert np.isnan(out).sum() == 345


def test_g2_spatial_nan_template(oracle):
    g = golden('g2_spatial_nan_template')
    for k in 'abc':
        sx, sy = g[f'step_{k}']
        out = oracle.cross_correlate_2d(g['template'], g[f'region_{k}'], int(sx), int(sy))
        close(out, g[f'surface_{k}'])


def test_g3_spatial_band(oracle):
    g = golden('g3_spatial_band')
    for i in range(len(g['templates'])):
        out = oracle.cross_correlate_2d(g['templates'][i], g['bands'][i])
        close(out, g['surfaces'][i])


def test_g3_band_scan_row0(oracle):
    '''The dense band scan's scores are row 0 of the spatial NCC surface of a
    31-row band.'''
    g = golden('g3_spatial_band')
    t, b = g['templates'][0], g['bands'][0]
    # left image holds the template at x = 128; right image is the band
    left = np.zeros((31, 287), np.uint8)
    left[:, 128:159] = t
    _, _, scores = oracle.band_scan(left, b, 31, 31, 128, want_scores=True)
    row = scores[0, 128]                      # window at x = 128: offsets -128..128 -> cols 0..256
    close(row, g['surfaces'][0][0, :257])


def test_template_larger_than_region_raises(oracle):
    with pytest.raises(Exception, match='must not exceed'):
        oracle.cross_correlate_2d(np.zeros((5, 5)), np.zeros((4, 9)))


def test_g4_spectral_small(oracle):
    g = golden('g4_spectral_small')
    tags = sorted(k[:-4] for k in g.files if k.endswith('_out'))
    assert len(tags) == 12
    for tag in tags:
        surf, mx, arg = oracle.cross_correlate_2d_spectral(g[f'{tag}_t'], g[f'{tag}_r'], want='both')
        ref = g[f'{tag}_out']
        if tag == 'const_r':
            assert np.isnan(ref).all() and np.isnan(surf).all() and np.isnan(mx)
            continue
        close(surf, ref)
        close(mx, np.max(ref))
        close(oracle.cross_correlate_2d_spectral_fft(g[f'{tag}_t'], g[f'{tag}_r']), ref)
    assert np.all(g['const_t_same_mean_out'] == 1.0)
    assert np.all(g['const_t_diff_mean_out'] == 0.0)
    assert np.all(g['lowstd_t_out'] == 0.0)


def test_g4_spectral_images(oracle):
    g = golden('g4_spectral_images')
    out = oracle.cross_correlate_2d_spectral_fft(g['template'], g['region'])
    close(out.ravel()[g['sample_idx']], g['sample_val'])
    close(out[::16], g['rows'])
    assert tuple(oracle.max_pos(out)) == tuple(g['max_pos']) == (288, 150)
    # direct evaluation on a few rows of the same surface
    direct = oracle.cross_correlate_2d_spectral(g['template'], g['region'][:, :])
    close(direct[::16], g['rows'])


def test_g4_spectral_rand256(oracle):
    g = golden('g4_spectral_rand256')
    t, r, idx = rand_pair(256)
    assert t.astype(np.float64).sum() == g['t_checksum']
    out = oracle.cross_correlate_2d_spectral_fft(t.astype(np.float64), r.astype(np.float64))
    close(out.ravel()[idx], g['sample_val'])
    assert int(np.argmax(out)) == int(g['argmax'])


def test_g5_1d(oracle):
    g = golden('g5_1d')
    close(oracle.cross_correlate_1d(g['v1'], g['v2']), g['xc'])
    close(oracle.cross_correlate_1d(g['v1'], g['v2'], True), g['xc_norm'])
    close(oracle.cross_correlate_1d(g['v1'], g['v2'], True), g['xc_raw_norm'], 1e-9)
    close(oracle.cross_correlate_1d_spectral(g['v1'], g['v2']), g['xc_spec'])
    close(oracle.cross_correlate_1d_spectral(g['t_short'], g['s_long']), g['xc_spec_uneq'])
    assert int(np.argmax(g['xc'])) == int(np.argmax(g['xc_spec'])) == 216
    step = np.pi / 2000
    x = np.arange(0, 4 * np.pi, step)
    assert len(x) == int(g['cs_n'])
    full = oracle.cross_correlate_1d(np.cos(x), -np.sin(x), True)
    close(full[::16], g['cs_xc_norm_dec'], 1e-9)
    assert int(np.argmax(full)) == int(g['cs_xc_norm_argmax'])
    fs = oracle.cross_correlate_1d_spectral(np.cos(x), -np.sin(x))
    close(fs[::16], g['cs_spec_dec'], 1e-9)


_CONE_SCANS = {}


def cone_scan(oracle, seq_cfg):
    '''The oracle's scan of the cone pair for one sequence config, computed once
    per test session (several tests look at the same scans).'''
    key = json.dumps(seq_cfg, sort_keys=True)
    if key not in _CONE_SCANS:
        im = golden('g6_cone_images')
        _CONE_SCANS[key] = oracle.sequence_scan(im['left'] / 255.0, im['right'] / 255.0, seq_cfg)
    return _CONE_SCANS[key]


def check_scan(oracle, g, left=None, right=None):
    cfg = json.loads(str(g['config']))
    for s in range(int(g['n_seq'])):
        res = cone_scan(oracle, cfg[s]) if left is None else oracle.sequence_scan(left, right, cfg[s])
        for st, windows in enumerate(res):
            want = g[f'seq{s}_stage{st}_windows']
            assert len(windows) == len(want)
            centres = g[f'seq{s}_stage{st}_centres']
            sizes = g[f'seq{s}_stage{st}_sizes']
            for i, (wid, w) in enumerate(windows.items()):
                row = want[i]
                assert wid == int(row[0])
                assert (w['row'], w['column']) == (int(row[1]), int(row[2]))
                assert w['centre'] == (row[3], row[4])
                assert (w['dp_x'], w['dp_y']) == (row[5], row[6]), (s, st, wid)
                assert len(w['stage_centres']) == int(row[7])
                assert len(w['stage_sizes']) == int(row[8])
                for k, c in enumerate(w['stage_centres']):
                    assert tuple(c) == tuple(centres[i, k])
                for k, z in enumerate(w['stage_sizes']):
                    assert tuple(z) == tuple(sizes[i, k])


@pytest.mark.parametrize('cfg', ['scan_config_portal', 'scan_config', 'scan_config_test_2'])
def test_g6_cone_scans(oracle, cfg):
    check_scan(oracle, golden(f'g6_cone_{cfg}'))


def test_g6_box_test1(oracle):
    g = golden('g6_box256_scan_config_test_1')
    check_scan(oracle, g, g['left'] / 255.0, g['right'] / 255.0)


def test_g9_multipass(oracle):
    check_scan(oracle, golden('g9_cone_multipass'))


def test_g8_dense_u8(oracle):
    g = golden('g8_dense_u8')
    check_scan(oracle, g, g['left'] / 255.0, g['right'] / 255.0)


# ---- displacement grids ----------------------------------------------------------------
@pytest.mark.parametrize('tag', ['scan_config', 'portal', 'multipass'])
def test_g10_depth_grids(oracle, tag):
    '''The grids the reference's stereo driver saves (sum over sequences, average
    by contributions) from the oracle's scan results: identical, bit for bit.'''
    g = golden('g10_cone_depth_grids')
    cfg = json.loads(str(g[f'{tag}_config']))
    results = [cone_scan(oracle, seq) for seq in cfg]
    grids = oracle.depth_grids(results, golden('g6_cone_images')['left'].shape)
    for got, key in zip(grids, ('depth_grid', 'dp_x_grid', 'dp_y_grid', 'contributions_grid')):
        assert np.array_equal(got, g[f'{tag}_{key}']), key


# ---- calibration peak finder -----------------------------------------------------------
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_g11_find_peaks(oracle, tag):
    g = golden('g11_find_peaks')
    x, y = oracle.find_peaks(g[f'{tag}_image'], 10, 0.2)
    assert np.array_equal(x, g[f'{tag}_x']) and np.array_equal(y, g[f'{tag}_y'])


def test_g13_spatial_nan_region(oracle):
    g = golden('g13_spatial_nan_region')
    for k in 'ab':
        sx, sy = (int(v) for v in g[f'step_{k}'])
        close(oracle.cross_correlate_2d(g['template'], g['region'], sx, sy), g[f'surface_{k}'])


# ---- image preparation ---------------------------------------------------------------
def test_g12_greyscale_with_nan(oracle):
    '''The reference's loader on its own two RGBA images: same grey levels, NaN at
    the same pixels, bit for bit.'''
    g = golden('g12_greyscale_with_nan')
    for key in ('template', 'rm_template'):
        got = oracle.greyscale(g[f'{key}_rgba'], nan_alpha=True)
        assert np.array_equal(got, g[f'{key}_grey'], equal_nan=True)
    # the spatial golden's template is the same loader's output
    assert np.array_equal(oracle.greyscale(g['rm_template_rgba'], nan_alpha=True),
                          golden('g2_spatial_nan_template')['template'], equal_nan=True)


@pytest.mark.parametrize('mode,channels', [('RGBA', 4), ('RGB', 3), ('LA', 2), ('L', 1)])
def test_greyscale_matches_pil(oracle, mode, channels):
    from PIL import Image
    rng = np.random.default_rng(channels)
    pixels = rng.integers(0, 256, size=(37, 41, channels), dtype=np.uint8)
    if channels == 1:
        pixels = pixels[..., 0]
    ref = np.asarray(Image.fromarray(pixels, mode=mode).convert('L'))
    assert np.array_equal(oracle.greyscale(pixels), ref)


def ndimage_resize(image, out_shape):
    '''The scipy.ndimage calls scikit-image 0.19's `resize` makes with its defaults
    (scikit-image itself is not installed: parity with it is unpinned, parity with
    the library it delegates to is what this checks).'''
    from scipy import ndimage
    image = np.asarray(image)
    if image.dtype == np.uint8:
        image = np.multiply(image, 1. / 255, dtype=np.float64)
    factors = np.divide(image.shape, out_shape)
    filtered = image
    if any(o < i for o, i in zip(out_shape, image.shape)):
        filtered = ndimage.gaussian_filter(image, np.maximum(0, (factors - 1) / 2), cval=0, mode='mirror')
    out = ndimage.zoom(filtered, [1 / f for f in factors], order=1, mode='mirror', cval=0, grid_mode=True)
    np.clip(out, np.nanmin(image), np.nanmax(image), out=out)
    return out


RESIZE_CASES = [((64, 80), 2, 'u8'), ((101, 77), 3, 'u8'), ((50, 64), 1, 'u8'), ((40, 200), 4, 'f64'),
                ((33, 47), 1.5, 'f64'), ((50, 300), 7, 'u8'), ((64, 64), 2, 'nan'), ((9, 2000), 1, 'f64'),
                ((1, 64), 2, 'f64'), ((30, 30), 0.5, 'f64')]


def resize_case(shape, factor, kind):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    if kind == 'u8':
        image = rng.integers(0, 256, size=shape, dtype=np.uint8)
    else:
        image = rng.normal(size=shape) * 50
        if kind == 'nan':
            image[10:12, 20:23] = np.nan
    return image, (max(1, int(np.floor(shape[0] / factor))), int(np.floor(shape[1] / factor)))


@pytest.mark.parametrize('shape,factor,kind', RESIZE_CASES)
def test_resize_matches_ndimage(oracle, shape, factor, kind):
    image, out_shape = resize_case(shape, factor, kind)
    got, ref = oracle.resize(image, out_shape), ndimage_resize(image, out_shape)
    if factor >= 1:
        assert np.array_equal(got, ref, equal_nan=True)      # same operations in the same order
    else:
        close(got, ref, 1e-13)
    if shape[0] > 1:
        assert np.array_equal(oracle.downsample(image, factor), got, equal_nan=True)


def test_g14_band_fold_matches_reference_rule(oracle):
    '''The oracle's band scan in selection-rule mode against the reference's own
    sequence_scan over its spatial correlator.'''
    from helpers import check_band_fold_against_g14
    check_band_fold_against_g14(
        lambda left, right, win, md, thr: oracle.band_scan(left, right, win, win, md, mode='fold', threshold=thr))
