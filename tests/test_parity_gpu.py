'''Parity of the HIP path (through the host modules and the C ABI) with the golden
vectors taken from the reference and with the CPU oracle.  Needs an MI355X.

Tolerances: NCC surfaces are compared at 1e-5 relative with the scale floored
at 1 (surfaces live on [-1, 1]); positions, disparities and selected regions
must match exactly.
'''
import io
import contextlib
import json

import numpy as np
import pytest

from helpers import assert_surface_close, golden, rand_pair

pytestmark = pytest.mark.gpu

SURFACE_TOL = 1e-5


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope='module')
def mods():
    import correlation
    import correlation_spectral
    import sv_image_comparison
    import utils
    import band_scan
    import _native
    _native.torch_cuda()
    return dict(c=correlation, s=correlation_spectral, v=sv_image_comparison, u=utils, b=band_scan,
                n=_native)


# ---- a1 ------------------------------------------------------------------------
def test_g1_spatial_template_region(mods):
    g = golden('g1_spatial_template_region')
    out = quiet(mods['c'].cross_correlate_2d, g['template'], g['region'], 5, 5)
    assert_surface_close(out, g['surface'], SURFACE_TOL, 'g1')
    assert tuple(mods['u'].max_pos(out, 5, 5)) == tuple(g['max_pos'])


def test_spatial_prints_reference_progress_lines(mods, capsys):
    mods['c'].cross_correlate_2d(np.ones((3, 3)) * np.arange(3), np.arange(25 * 9, dtype=np.uint8).reshape(25, 9))
    lines = capsys.readouterr().out.split('\n')
    assert lines[:3] == ['0 / 25', '10 / 25', '20 / 25']


def test_g13_spatial_nan_region(mods):
    '''NaN pixels in the region: NaN for every placement whose snapshot touches one,
    the reference's values everywhere else.'''
    g = golden('g13_spatial_nan_region')
    for k in 'ab':
        sx, sy = (int(v) for v in g[f'step_{k}'])
        out = quiet(mods['c'].cross_correlate_2d, g['template'], g['region'], sx, sy)
        assert_surface_close(out, g[f'surface_{k}'], SURFACE_TOL, f'g13{k}')
    # float32 regions take the same route
    out = quiet(mods['c'].cross_correlate_2d, g['template'], g['region'].astype(np.float32), 1, 1)
    assert np.array_equal(np.isnan(out), np.isnan(g['surface_a']))


def test_g2_spatial_nan_template(mods):
    g = golden('g2_spatial_nan_template')
    for k in 'abc':
        sx, sy = (int(v) for v in g[f'step_{k}'])
        out = quiet(mods['c'].cross_correlate_2d, g['template'], g[f'region_{k}'], sx, sy)
        assert_surface_close(out, g[f'surface_{k}'], SURFACE_TOL, f'g2{k}')


def test_g3_spatial_band(mods):
    g = golden('g3_spatial_band')
    for i in range(len(g['templates'])):
        out = quiet(mods['c'].cross_correlate_2d, g['templates'][i].astype(np.float64), g['bands'][i])
        assert_surface_close(out, g['surfaces'][i], SURFACE_TOL, f'g3[{i}]')
        assert int(np.nanargmax(out[0, :257])) == int(np.nanargmax(g['surfaces'][i][0, :257]))


def test_spatial_float_region_and_constant_patches(mods, oracle):
    rng = np.random.default_rng(11)
    region = rng.random((40, 56))
    region[5:20, 8:30] = 0.5           # constant snapshot -> NaN outputs
    template = rng.random((9, 13))
    template[2, 3] = np.nan
    out = quiet(mods['c'].cross_correlate_2d, template, region, 2, 3)
    assert_surface_close(out, oracle.cross_correlate_2d(template, region, 2, 3), SURFACE_TOL, 'float region')
    flat_t = np.full((4, 4), 7.0)
    out = quiet(mods['c'].cross_correlate_2d, flat_t, (region * 255).astype(np.uint8))
    assert np.isnan(out).all()


@pytest.mark.parametrize('tshape,rshape,sx,sy', [((9, 13), (40, 56), 1, 1), ((17, 31), (70, 1100), 1, 2),
                                                 ((5, 600), (23, 1300), 1, 1), ((33, 8), (90, 530), 1, 5),
                                                 ((3, 1025), (9, 1030), 1, 3), ((12, 7), (30, 64), 3, 1)])
def test_spatial_unit_column_stride_shapes(mods, oracle, tshape, rshape, sx, sy):
    '''Shapes that walk the tiled kernel through several column tiles, template
    chunks, row strides and ragged edges (uint8 and float regions).'''
    rng = np.random.default_rng(tshape[1] * 7 + rshape[1])
    region = rng.integers(0, 256, size=rshape, dtype=np.uint8)
    template = rng.random(tshape) * 255
    template[rng.random(tshape) < 0.1] = np.nan
    out = quiet(mods['c'].cross_correlate_2d, template, region, sx, sy)
    assert_surface_close(out, oracle.cross_correlate_2d(template, region, sx, sy), SURFACE_TOL, 'u8 region')
    fregion = rng.random(rshape)
    out = quiet(mods['c'].cross_correlate_2d, template, fregion, sx, sy)
    assert_surface_close(out, oracle.cross_correlate_2d(template, fregion, sx, sy), SURFACE_TOL, 'float region')


@pytest.mark.parametrize('tshape,rshape,sx,sy', [
    ((135, 136), (433, 862), 5, 5),     # the step-5 driver case: 8 output rows per lane, columns split 9 ways
    ((9, 7), (60, 300), 2, 9),          # rows of a group share nothing: one output row per lane
    ((40, 210), (90, 700), 3, 4),       # wide template: several pieces per wavefront
    ((17, 5), (200, 130), 7, 1),        # narrow template, unit row step
    ((33, 33), (33, 33), 2, 2),         # template as large as the region: every placement but one overhangs
    ((12, 50), (64, 1500), 11, 3)])
def test_spatial_strided_columns_vs_oracle(mods, oracle, tshape, rshape, sx, sy):
    '''Column steps above 1 run the strided kernel (a lane owns one output column
    and up to 8 output rows; template columns split over wavefronts and grid
    slices): every output, incl. the overhanging bottom/right placements and a
    NaN-masked template, against the oracle.'''
    rng = np.random.default_rng(tshape[0] * 7 + sx)
    template = rng.random(tshape) * 255
    template[rng.random(tshape) < 0.2] = np.nan
    region = rng.integers(0, 256, size=rshape, dtype=np.uint8)
    got = quiet(mods['c'].cross_correlate_2d, template, region, sx, sy)
    want = oracle.cross_correlate_2d(template, region, sx, sy)
    assert got.shape == (-(-rshape[0] // sy), -(-rshape[1] // sx))
    assert_surface_close(got, want, SURFACE_TOL, f'strided {tshape} over {rshape} step ({sx},{sy})')
    # float region, no mask
    regf = rng.random(rshape)
    got = quiet(mods['c'].cross_correlate_2d, np.nan_to_num(template), regf, sx, sy)
    want = oracle.cross_correlate_2d(np.nan_to_num(template), regf, sx, sy)
    assert_surface_close(got, want, SURFACE_TOL, f'strided float {tshape} over {rshape}')


def test_spatial_float_region_with_large_mean(mods, oracle):
    '''A floating-point region whose mean is far larger than its deviation (1000 +-
    0.05): the window statistics come from tables of the pixels minus the region
    mean, so the small variance survives (tables of the plain pixels would round it
    away: 1e6 against 2.5e-3 in sums of squares).'''
    rng = np.random.default_rng(77)
    region = 1000.0 + 0.1 * rng.random((120, 150))
    template = region[40:65, 60:90].copy()
    for reg in (region, region.astype(np.float32)):
        got = quiet(mods['c'].cross_correlate_2d, template, reg, 3, 2)
        want = oracle.cross_correlate_2d(template, reg.astype(np.float64), 3, 2)
        assert np.isfinite(got).all()
        assert_surface_close(got, want, 2e-4 if reg.dtype == np.float32 else SURFACE_TOL, f'large mean {reg.dtype}')
    assert np.unravel_index(np.argmax(got), got.shape) == (20, 20)


def test_spatial_low_contrast_patch_of_a_large_float_region(mods, oracle):
    '''A patch whose deviation is 1e-5 of the image's, far from the origin of a 2048 x
    2048 float region: the variance of a window inside it is below the rounding of the
    summed-area tables there, so it comes from the window's own pixels - finite values
    that match the oracle, not NaN.  An exactly constant patch is NaN (the reference's
    0 / 0).'''
    rng = np.random.default_rng(31)
    region = rng.random((2048, 2048))
    region[1800:1864, 1900:1964] = 0.5 + 1e-5 * region.std() * rng.standard_normal((64, 64))
    region[600:640, 200:240] = 0.25
    template = rng.random((16, 16))
    got = quiet(mods['c'].cross_correlate_2d, template, region, 8, 8)
    want = oracle.cross_correlate_2d(template, region, 8, 8)
    inside = (slice(1800 // 8, (1864 - 16) // 8 + 1), slice(1904 // 8, (1964 - 16) // 8 + 1))
    assert np.isfinite(want[inside]).all() and np.isfinite(got[inside]).all()
    assert np.abs(want[inside]).max() > 0.05
    constant = (slice(600 // 8, (640 - 16) // 8 + 1), slice(200 // 8, (240 - 16) // 8 + 1))
    assert np.isnan(got[constant]).all()
    # the reference's own value for a constant float snapshot is rounding noise over
    # rounding noise (its mean of equal numbers need not equal them): left out of the comparison
    keep = np.ones(got.shape, dtype=bool)
    keep[constant] = False
    assert_surface_close(np.where(keep, got, 0.0), np.where(keep, want, 0.0), SURFACE_TOL, 'low-contrast patch')


def test_spatial_random_shapes_and_steps_vs_oracle(mods, oracle):
    '''Random template / region shapes (1 x 1 up to 300 x 700) and steps 1..8 in each
    direction, uint8 regions with and without a NaN-masked template and float regions:
    every launch geometry of the table, sliding (unit and strided columns, row and
    column slices) and epilogue kernels against the oracle.'''
    rng = np.random.default_rng(5)
    for trial in range(60):
        rh, rw = int(rng.integers(1, 300)), int(rng.integers(1, 700))
        th, tw = int(rng.integers(1, rh + 1)), int(rng.integers(1, rw + 1))
        if trial % 4 == 0:
            th, tw = min(th, 40), min(tw, 40)
        sx, sy = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if trial % 3 == 0:
            sx = 1
        if trial % 5 == 0:
            sy = 1
        kind = trial % 3
        region = (rng.integers(0, 256, size=(rh, rw), dtype=np.uint8) if kind != 2
                  else rng.random((rh, rw)) * 3 + 10)
        template = rng.random((th, tw)) * 255
        if kind == 1 and th * tw > 4:
            template[rng.random((th, tw)) < 0.15] = np.nan
        got = quiet(mods['c'].cross_correlate_2d, template, region, sx, sy)
        want = oracle.cross_correlate_2d(template, region.astype(np.float64), sx, sy)
        assert_surface_close(got, want, 2e-5, f'trial {trial}: {(th, tw)} over {(rh, rw)}, steps {(sx, sy)}, kind {kind}')


def test_spatial_template_exceeds_region(mods):
    with pytest.raises(Exception, match='Dimensions of template must not exceed those of region.'):
        mods['c'].cross_correlate_2d(np.zeros((5, 5)), np.zeros((4, 9), np.uint8))


# ---- a2 ------------------------------------------------------------------------
def test_spectral_random_shapes_vs_oracle(mods, oracle):
    '''Random template / region shapes through cross_correlate_2d_spectral (templates up to
    twice the region: zero-padded, cropped, both), with constant templates (the
    low-deviation branch), constant regions (NaN surface; constants that are exact in
    binary, as the reference's 0/0 depends on it) and uint8 regions: pair kernel and
    sliding route against the oracle's direct circular correlation.'''
    rng = np.random.default_rng(8)
    for trial in range(60):
        big = trial % 3 == 0
        rh, rw = ((int(rng.integers(1, 130)), int(rng.integers(1, 130))) if big
                  else (int(rng.integers(1, 65)), int(rng.integers(1, 65))))
        th, tw = int(rng.integers(1, 2 * rh + 1)), int(rng.integers(1, 2 * rw + 1))
        if not big:
            th, tw = min(th, 64), min(tw, 64)
        t, r = rng.random((th, tw)), rng.random((rh, rw))
        if trial % 7 == 0:
            t[:] = 0.25
        if trial % 11 == 0:
            r[:] = 0.5
        if trial % 13 == 0:
            r = (r * 255).astype(np.uint8)
        got = mods['s'].cross_correlate_2d_spectral(t, r)
        with np.errstate(all='ignore'):
            want = oracle.cross_correlate_2d_spectral(t, np.asarray(r, dtype=np.float64))
        assert_surface_close(got, want, 2e-5, f'trial {trial}: {(th, tw)} against {(rh, rw)}')


def test_g4_spectral_small(mods):
    g = golden('g4_spectral_small')
    tags = sorted(k[:-4] for k in g.files if k.endswith('_out'))
    for tag in tags:
        out = mods['s'].cross_correlate_2d_spectral(g[f'{tag}_t'], g[f'{tag}_r'])
        assert_surface_close(out, g[f'{tag}_out'], SURFACE_TOL, tag)


def test_g4_spectral_images(mods):
    g = golden('g4_spectral_images')
    out = mods['s'].cross_correlate_2d_spectral(g['template'], g['region'])
    assert out.shape == g['region'].shape and out.dtype == np.float64
    assert_surface_close(out.ravel()[g['sample_idx']], g['sample_val'], SURFACE_TOL, 'samples')
    assert_surface_close(out[::16], g['rows'], SURFACE_TOL, 'rows')
    assert tuple(mods['u'].max_pos(out)) == tuple(g['max_pos'])
    assert abs(out.max() - float(g['max_val'])) < SURFACE_TOL


def test_g4_spectral_rand256(mods):
    g = golden('g4_spectral_rand256')
    t, r, idx = rand_pair(256)
    out = mods['s'].cross_correlate_2d_spectral(t, r)
    assert_surface_close(out.ravel()[idx], g['sample_val'], SURFACE_TOL, 'rand256')
    assert int(np.argmax(out)) == int(g['argmax'])


@pytest.mark.parametrize('tshape,rshape', [((70, 70), (80, 90)), ((100, 700), (100, 700)), ((90, 40), (65, 600)),
                                           ((3, 1100), (66, 1200))])
def test_spectral_large_shapes_vs_oracle(mods, oracle, tshape, rshape):
    '''Non-power-of-two shapes beyond the pair kernel: padded, cropped and wide
    templates through the sliding kernel.'''
    rng = np.random.default_rng(tshape[0] + rshape[1])
    t, r = rng.random(tshape), rng.random(rshape)
    out = mods['s'].cross_correlate_2d_spectral(t, r)
    assert_surface_close(out, oracle.cross_correlate_2d_spectral(t, r), SURFACE_TOL, 'large circular')


@pytest.mark.parametrize('tshape,rshape', [
    ((2048, 2048), (2048, 2048)),       # power-of-two sides beyond 1024: the FFT kernels directly
    ((512, 4096), (512, 4096)),
    ((1000, 1200), (1000, 1200)),       # any shape: both operands zero-padded to 2048 x 4096
    ((700, 900), (1000, 1200)),         # template padded to the region (covers half of it)
    ((1300, 1100), (1000, 800)),        # template cropped to the region
    ((333, 1500), (333, 1500))])
def test_spectral_fft_routes_large_and_any_shape(mods, oracle, tshape, rshape):
    '''Large pairs never fall back to the O(N^2) sliding kernel: power-of-two regions
    up to 4096 per side run through the FFT kernels, other shapes through FFTs of
    zero-padded operands that keep the circular semantics of the original shape
    (correlation_spectral.py:51-54).  Against numpy's float64 FFT evaluation.'''
    rng = np.random.default_rng(tshape[0] + rshape[1])
    t, r = rng.random(tshape), rng.random(rshape)
    out = mods['s'].cross_correlate_2d_spectral(t, r)
    want = oracle.cross_correlate_2d_spectral_fft(t, r)
    assert_surface_close(out, want, SURFACE_TOL, f'fft route {tshape} {rshape}')
    # a planted circular shift is found exactly where it is
    if tshape == rshape:
        rolled = np.roll(t, (17, 230), axis=(0, 1))
        out = mods['s'].cross_correlate_2d_spectral(t, rolled)
        assert np.unravel_index(np.argmax(out), out.shape) == (17, 230)
        assert abs(out.max() - 1.0) < 1e-5


def test_spectral_fft_route_low_std_and_constant_region(mods):
    t = np.full((1000, 1200), 3.0)
    rng = np.random.default_rng(2)
    r = rng.random((1000, 1200))
    assert np.all(mods['s'].cross_correlate_2d_spectral(t, r) == 0.0)
    assert np.all(mods['s'].cross_correlate_2d_spectral(t, t.copy()) == 1.0)
    with np.errstate(all='ignore'):
        assert np.isnan(mods['s'].cross_correlate_2d_spectral(r, t)).all()   # 0 / 0 as in the reference


def test_spectral_large_low_std_branches(mods):
    t = np.full((70, 70), 3.0)
    r = np.full((80, 90), 3.0)
    assert np.all(mods['s'].cross_correlate_2d_spectral(t, r) == 1.0)
    assert np.all(mods['s'].cross_correlate_2d_spectral(t, r + 1) == 0.0)


# ---- 1D ------------------------------------------------------------------------
def close_1d(got, want, what):
    scale = np.max(np.abs(want))
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= SURFACE_TOL * scale, what


def test_g5_1d(mods):
    g = golden('g5_1d')
    c, s = mods['c'], mods['s']
    close_1d(quiet(c.cross_correlate_1d, g['v1'], g['v2']), g['xc'], 'xc')
    close_1d(quiet(c.cross_correlate_1d, g['v1'], g['v2'], True), g['xc_norm'], 'xc_norm')
    raw = quiet(c.cross_correlate_1d_raw, list(g['v1']), list(g['v2']), True)
    assert isinstance(raw, list)
    close_1d(np.array(raw), g['xc_raw_norm'], 'raw')
    spec = s.cross_correlate_1d_spectral(g['v1'], g['v2'])
    close_1d(spec, g['xc_spec'], 'spec')
    close_1d(s.cross_correlate_1d_spectral(g['t_short'], g['s_long']), g['xc_spec_uneq'], 'uneq')
    assert int(np.argmax(spec)) == 216 == int(np.argmax(quiet(c.cross_correlate_1d, g['v1'], g['v2'])))
    x = np.arange(0, 4 * np.pi, np.pi / 2000)
    full = quiet(c.cross_correlate_1d, np.cos(x), -np.sin(x), True)
    close_1d(full[::16], g['cs_xc_norm_dec'], 'cos/sin')
    assert abs(int(np.argmax(full)) - int(g['cs_xc_norm_argmax'])) <= 1
    fs = s.cross_correlate_1d_spectral(np.cos(x), -np.sin(x))
    close_1d(fs[::16], g['cs_spec_dec'], 'cos/sin spectral')


@pytest.mark.parametrize('n', [1, 2, 7, 511, 512, 513, 1500])
def test_1d_lengths_vs_oracle(mods, oracle, n):
    rng = np.random.default_rng(n)
    v1, v2 = rng.standard_normal(n) + 3, rng.standard_normal(n) - 1
    close_1d(quiet(mods['c'].cross_correlate_1d, v1, v2), oracle.cross_correlate_1d(v1, v2), f'xc n={n}')
    if n > 1:
        close_1d(mods['s'].cross_correlate_1d_spectral(v1, v2), oracle.cross_correlate_1d_spectral(v1, v2),
                 f'spectral n={n}')
        close_1d(mods['s'].cross_correlate_1d_spectral(v1[:n // 2 + 1], v2),
                 oracle.cross_correlate_1d_spectral(v1[:n // 2 + 1], v2), f'spectral uneven n={n}')
        # a signal shorter than the template: the kernel lets the signal slide instead
        close_1d(mods['s'].cross_correlate_1d_spectral(v1, v2[:n // 2]),
                 oracle.cross_correlate_1d_spectral(v1, v2[:n // 2]), f'spectral short signal n={n}')


def test_1d_spectral_long_template_short_signal(mods, oracle):
    '''The audio-search shape of the reference's music demo: a long recording as
    the template argument, a short snippet as the signal.'''
    rng = np.random.default_rng(99)
    long = rng.standard_normal(30000) * 1000
    snippet = long[12345:12345 + 2100] + 50 * rng.standard_normal(2100)
    got = mods['s'].cross_correlate_1d_spectral(long, snippet)
    want = oracle.cross_correlate_1d_spectral(long, snippet)
    close_1d(got, want, 'long template')
    assert int(np.argmax(got)) == int(np.argmax(want))


def test_1d_long_arrays_take_the_transform_route(mods):
    '''Above 2e9 products the 1D correlations leave the sliding kernel for the 2D
    transforms (arrays cut into rows; raw[aQ + b] = C[a][b] + C[a + 1][b - Q]).  The
    full outputs against the reference's own formulas evaluated with numpy's float64
    FFTs (correlation_spectral.py:17-29; correlation.py:59-73 as a linear correlation
    of the centred, zero-padded-then-centred vectors), within 1e-5 of the scale.'''
    rng = np.random.default_rng(2024)

    def reference_spectral(template, signal):
        padded = np.concatenate([np.zeros(template.size - 1), signal])
        t = (template - template.mean()) / (template.std() * template.size)
        size = signal.size + template.size - 1
        return np.real(np.fft.ifft(np.conj(np.fft.fft(t, n=size)) * np.fft.fft(padded, n=size)))

    def reference_1d(v1, v2):
        n = v1.size
        a = v1 - v1.mean()
        b = np.concatenate([np.zeros(n - 1), v2, np.zeros(n - 1)]) - v2.mean()
        size = 4 * n
        full = np.real(np.fft.ifft(np.conj(np.fft.fft(a, n=size)) * np.fft.fft(b, n=size)))
        return full[:2 * n - 1] / n

    # the template slides (nt * N = 3.8e10 products)
    t, sig = rng.standard_normal(120_000) * 3 + 1, rng.standard_normal(200_000)
    sig[50_000:170_000] += 0.5 * t
    got, want = mods['s'].cross_correlate_1d_spectral(t, sig), reference_spectral(t, sig)
    close_1d(got, want, 'transform route, template slides')
    assert int(np.argmax(got)) == int(np.argmax(want)) == 50_000 + 120_000 - 1
    # the shorter signal slides (ns * N = 4e10)
    long = rng.standard_normal(300_000) * 100
    snippet = long[123_456:223_456] + 5 * rng.standard_normal(100_000)
    got, want = mods['s'].cross_correlate_1d_spectral(long, snippet), reference_spectral(long, snippet)
    close_1d(got, want, 'transform route, signal slides')
    assert int(np.argmax(got)) == int(np.argmax(want))
    # cross_correlate_1d, n (2n - 1) = 2.9e10
    v1 = rng.standard_normal(120_000) + 2
    v2 = np.roll(v1, 777) + 0.1 * rng.standard_normal(120_000)
    got, want = quiet(mods['c'].cross_correlate_1d, v1, v2), reference_1d(v1, v2)
    close_1d(got, want, 'transform route, cross_correlate_1d')
    assert int(np.argmax(got)) == int(np.argmax(want))


def test_1d_unequal_raises(mods):
    with pytest.raises(Exception, match='unequal length'):
        mods['c'].cross_correlate_1d(np.zeros(4), np.zeros(5))


# ---- a11 -----------------------------------------------------------------------
def test_max_pos(mods):
    a = np.array([[1.0, np.nan, 3.0], [3.0, 2.0, np.nan]])
    assert mods['u'].max_pos(a) == (2, 0)
    assert mods['u'].max_pos(a, 5, 7) == (14, 0)
    with pytest.raises(ValueError):
        mods['u'].max_pos(np.full((2, 2), np.nan))
    # a surface that is still on the device is reduced there
    import torch
    rng = np.random.default_rng(3)
    big = rng.random((300, 517))
    big[rng.random(big.shape) < 0.1] = np.nan
    big[7, 11] = big[200, 3] = 2.0                   # tie: the first one wins
    dev = torch.from_numpy(big).cuda()
    assert mods['u'].max_pos(dev, 3, 2) == mods['u'].max_pos(big, 3, 2) == (2 * 11, 3 * 7)
    with pytest.raises(ValueError):
        mods['u'].max_pos(torch.full((4, 4), float('nan'), device='cuda'))


def test_sat2d_tables(mods):
    '''svamd_sat2d through the C ABI: exact int64 tables for uint8 pixels (with an
    offset), float64 tables for float32 / float64 pixels, a pitch wider than the
    row, against numpy's cumulative sums.'''
    import torch
    nv = mods['n']
    rng = np.random.default_rng(9)
    for h, w, pitch in ((1, 1, 1), (37, 200, 200), (130, 65, 80), (600, 1031, 1031)):
        img = rng.integers(0, 256, size=(h, pitch), dtype=np.uint8)
        for dtype, code, offset, acc in ((np.uint8, nv.U8, 128.0, torch.int64), (np.float32, nv.F32, 0.25, torch.float64),
                                         (np.float64, nv.F64, -3.5, torch.float64)):
            src = img.astype(dtype) if dtype == np.uint8 else (img / 7.0).astype(dtype)
            d = nv.to_device(src)
            s1, s2 = nv.empty((h + 1, w + 1), acc), nv.empty((h + 1, w + 1), acc)
            nv.call('svamd_sat2d', nv.ptr(d), code, h, w, pitch, offset, nv.ptr(s1), nv.ptr(s2), nv.stream_ptr())
            v = src[:, :w].astype(np.float64) - offset
            want1 = np.zeros((h + 1, w + 1))
            want2 = np.zeros((h + 1, w + 1))
            want1[1:, 1:] = np.cumsum(np.cumsum(v, axis=0), axis=1)
            want2[1:, 1:] = np.cumsum(np.cumsum(v * v, axis=0), axis=1)
            g1, g2 = s1.cpu().numpy(), s2.cpu().numpy()
            if dtype == np.uint8:
                assert np.array_equal(g1, want1.astype(np.int64)) and np.array_equal(g2, want2.astype(np.int64))
            else:
                assert np.allclose(g1, want1, rtol=1e-12, atol=1e-9) and np.allclose(g2, want2, rtol=1e-12, atol=1e-9)
    with pytest.raises(nv.NativeError):
        nv.call('svamd_sat2d', nv.ptr(d), nv.U8, 4, 4, 4, 0.5, nv.ptr(s1), nv.ptr(s2), nv.stream_ptr())


def test_band_scan_random_geometries_vs_oracle(mods, oracle):
    '''Random window widths (unrolled and run-time instantiations), heights, image sizes,
    offset ranges, row strides and both selection modes of the dense band scan against
    the oracle: disparities and NaN patterns exactly, scores to 1e-11.'''
    import band_scan as bs
    rng = np.random.default_rng(31)
    for trial in range(30):
        ww = int(rng.choice([3, 5, 7, 8, 11, 12, 15, 16, 21, 25, 30, 31, 33, 40]))
        wh = int(rng.integers(1, min(40, 1024 // ww) + 1))
        H, W = wh + int(rng.integers(0, 90)), ww + int(rng.integers(0, 400))
        md = int(rng.integers(0, 70))
        stride = int(rng.choice([1, 1, 2, 3, 5]))
        left = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        shift = int(rng.integers(-min(md, 20), min(md, 20) + 1)) if md else 0
        right = np.clip(np.roll(left, shift, axis=1).astype(int) + rng.integers(-3, 4, size=(H, W)), 0, 255).astype(np.uint8)
        if trial % 5 == 0:
            left[: H // 2, : W // 3] = 77
            right[: H // 2, : W // 3] = 77
        thr = None if trial % 2 else float(rng.choice([0.0, 0.5, 0.9]))
        disp, val, scores = bs.band_scan(left, right, ww, wh, md, correlation_threshold=thr, row_stride=stride,
                                         want_scores=True)
        od, ov, osc = oracle.band_scan(left, right, ww, wh, md, stride_y=stride,
                                       mode='argmax' if thr is None else 'fold',
                                       threshold=0.0 if thr is None else thr, want_scores=True)
        what = f'trial {trial}: window {(ww, wh)} image {(H, W)} offsets {md} stride {stride} threshold {thr}'
        assert disp.shape == od.shape and np.array_equal(disp, od), what
        assert np.array_equal(np.isnan(scores), np.isnan(osc)) and np.array_equal(np.isnan(val), np.isnan(ov)), what
        if np.isfinite(osc).any():
            assert np.nanmax(np.abs(scores - osc)) <= 1e-11, what
        if np.isfinite(ov).any():
            assert np.nanmax(np.abs(val - ov)) <= 1e-11, what


# ---- scans ---------------------------------------------------------------------
def test_spectral_empty_template(mods):
    '''A template box cut off to nothing at the image edge: the reference's zero-padded
    transform of it is zero, so the surface is 0 - NaN for a constant region (0 / 0).'''
    r = np.random.default_rng(0).random((6, 7))
    for t in (np.zeros((0, 4)), np.zeros((3, 0))):
        assert np.array_equal(mods['s'].cross_correlate_2d_spectral(t, r), np.zeros((6, 7)))
        assert np.isnan(mods['s'].cross_correlate_2d_spectral(t, np.full((6, 7), 0.5))).all()


def test_sequence_scan_random_configs_vs_oracle(mods, oracle):
    '''Random small stereo pairs (shifted, noisy, some with a homogeneous patch) under
    random configs - window 5..25, schemes up to 9 x 5, shifts 0..2, four overlaps, three
    thresholds, one or two multi-pass stages of factor 2 or 3: every window's dp, centre,
    stage centres and stage sizes against the oracle's window-by-window scan.'''
    v = mods['v']
    rng = np.random.default_rng(21)
    for trial in range(20):
        H, W = int(rng.integers(40, 130)), int(rng.integers(40, 150))
        base = rng.integers(0, 256, size=(H + 20, W + 20)).astype(np.float64)
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4
        dx, dy = int(rng.integers(-4, 5)), int(rng.integers(-2, 3))
        left = np.round(base[10:10 + H, 10:10 + W]) / 255.0
        right = np.round(base[10 - dy:10 - dy + H, 10 - dx:10 - dx + W] + rng.normal(0, 2, (H, W))).clip(0, 255) / 255.0
        if trial % 6 == 0:
            left[5:25, 5:30] = 0.5
            right[5:25, 5:30] = 0.5
        w, h = int(rng.integers(5, 26)), int(rng.integers(5, 26))
        cfg = [dict(window_width=w, window_height=h,
                    scheme=[int(rng.choice([1, 3, 5, 7, 9])), int(rng.choice([1, 3, 5]))],
                    scheme_shift_size=[int(rng.integers(0, 3)), int(rng.integers(0, 3))],
                    window_overlap=float(rng.choice([0.0, 0.25, 0.5, 0.6])),
                    correlation_threshold=float(rng.choice([0.0, 0.3, 0.8])))]
        if trial % 2 == 0 and min(w, h) >= 8:
            cfg.append(dict(factor=int(rng.choice([2, 3]))))
            if trial % 4 == 0 and min(w, h) >= 16:
                cfg.append(dict(factor=2))
        got = quiet(v.sequence_scan, left, right, cfg)
        want = oracle.sequence_scan(left, right, cfg)
        assert len(got) == len(want)
        for stage, stage_want in enumerate(want):
            assert set(got[stage]) == set(stage_want), f'trial {trial} stage {stage}: window ids'
            for wid, ww in stage_want.items():
                g = got[stage][wid]
                what = f'trial {trial} {cfg} stage {stage} window {wid}'
                assert (g['dp_x'], g['dp_y']) == (ww['dp_x'], ww['dp_y']), what
                assert tuple(g['centre']) == tuple(ww['centre']), what
                assert [tuple(c) for c in g['stage_centres']] == [tuple(c) for c in ww['stage_centres']], what
                assert [tuple(c) for c in g['stage_sizes']] == [tuple(c) for c in ww['stage_sizes']], what


def check_scan(mods, g, left, right, precisions=('mixed',)):
    v = mods['v']
    cfg = json.loads(str(g['config']))
    for precision in precisions:
        v.SCAN_PRECISION = precision
        try:
            for s in range(int(g['n_seq'])):
                res = quiet(v.sequence_scan, left, right, cfg[s])
                assert len(res) == len(cfg[s])
                for st, windows in enumerate(res):
                    want = g[f'seq{s}_stage{st}_windows']
                    centres = g[f'seq{s}_stage{st}_centres']
                    sizes = g[f'seq{s}_stage{st}_sizes']
                    assert list(windows) == [int(i) for i in want[:, 0]]
                    for i, w in enumerate(windows.values()):
                        row = want[i]
                        assert (w['row'], w['column']) == (int(row[1]), int(row[2]))
                        assert w['centre'] == (row[3], row[4])
                        assert (w['dp_x'], w['dp_y']) == (row[5], row[6]), (precision, s, st, i)
                        assert len(w['target_regions']) == int(row[9])
                        assert [tuple(c) for c in w['stage_centres']] == \
                            [tuple(c) for c in centres[i, :int(row[7])]]
                        assert [tuple(z) for z in w['stage_sizes']] == \
                            [tuple(z) for z in sizes[i, :int(row[8])]]
        finally:
            v.SCAN_PRECISION = 'mixed'


@pytest.mark.parametrize('cfg', ['scan_config_portal', 'scan_config', 'scan_config_example',
                                 'scan_config_test_2'])
def test_g6_cone_scans(mods, cfg):
    im = golden('g6_cone_images')
    check_scan(mods, golden(f'g6_cone_{cfg}'), im['left'] / 255.0, im['right'] / 255.0,
               ('mixed', 'fp64') if cfg == 'scan_config_portal' else ('mixed',))


@pytest.mark.parametrize('name', ['box', 'tuscany'])
def test_g6_crops_test1(mods, name):
    g = golden(f'g6_{name}256_scan_config_test_1')
    check_scan(mods, g, g['left'] / 255.0, g['right'] / 255.0)


def test_g9_multipass(mods):
    im = golden('g6_cone_images')
    check_scan(mods, golden('g9_cone_multipass'), im['left'] / 255.0, im['right'] / 255.0)


def test_g7_dense_roll(mods):
    g = golden('g7_dense_roll5')
    check_scan(mods, g, g['left'], g['right'])


def test_g8_dense_u8(mods):
    g = golden('g8_dense_u8')
    check_scan(mods, g, g['left'] / 255.0, g['right'] / 255.0)


def test_image_scan_values_and_log(mods, capsys):
    '''image_scan on reference-shaped window dicts: position, correlation value
    and the summary line.'''
    v = mods['v']
    im = golden('g6_cone_images')
    g = golden('g6_cone_scan_config_portal')
    left, right = im['left'] / 255.0, im['right'] / 255.0
    cfg = json.loads(str(g['config']))[0][0]
    windows = v.whole_image_search_regions(left, right, cfg['window_width'], cfg['window_height'],
                                           tuple(cfg['scheme']), tuple(cfg['scheme_shift_size']),
                                           cfg['window_overlap'])
    calls = g['seq0_calls']
    moved = [int(c[0]) for c in calls if (c[1], c[2]) != windows[int(c[0])]['centre']][:6]
    for wid in moved + [0, 1, 2]:
        pos, val = v.image_scan(windows[wid], wid, cfg['correlation_threshold'])
        want = calls[wid]
        assert (pos[0], pos[1]) == (want[1], want[2])
        assert abs(val - want[3]) < 1e-11           # the selected pair's value is re-evaluated in float64
    out = capsys.readouterr().out
    assert out.count('Window ') == len(moved)
    assert f'Window {moved[0]}: centre at {windows[moved[0]]["centre"]}, max correlation of' in out


def test_scan_shape_errors(mods):
    v = mods['v']
    with pytest.raises(Exception, match='Scheme dimensions must be odd.'):
        v.region_pairs(5, 10, 5, 10, (2, 1), (0, 0))
    with pytest.raises(Exception, match='Dimensions of template and region images must match'):
        v.whole_image_search_regions(np.zeros((4, 4)), np.zeros((4, 5)), 2, 2)


def test_scan_correlator_hook(mods):
    '''Replacing the module-level x_corr_2d hook routes every pair through it.'''
    v = mods['v']
    seen = []
    orig = v.x_corr_2d

    def hook(t, r):
        seen.append(t.shape)
        return orig(t, r)
    rng = np.random.default_rng(1)
    left = rng.random((24, 36))
    cfg = [{'window_width': 12, 'window_height': 12, 'scheme': [3, 1], 'scheme_shift_size': [2, 0],
            'window_overlap': 0, 'correlation_threshold': 0.2}]
    want = quiet(v.sequence_scan, left, np.roll(left, 2, axis=1), cfg)
    v.x_corr_2d = hook
    try:
        got = quiet(v.sequence_scan, left, np.roll(left, 2, axis=1), cfg)
    finally:
        v.x_corr_2d = orig
    assert len(seen) > 0
    assert [(w['dp_x'], w['dp_y']) for w in got[0].values()] == \
        [(w['dp_x'], w['dp_y']) for w in want[0].values()]


# ---- pairs kernel through the C ABI vs the oracle --------------------------------
def test_scan_pairs_vs_oracle(mods, oracle):
    import torch
    nv = mods['n']
    rng = np.random.default_rng(21)
    H, W = 64, 96
    left, right = rng.random((H, W)), rng.random((H, W))
    right[10:22, 30:42] = 0.5                        # constant region -> NaN
    left[40:52, 0:12] = 0.25                         # constant template -> ones/zeros
    right[40:52, 0:12] = 0.25
    boxes = []
    for _ in range(200):
        th, tw, rh, rw = rng.integers(1, 40, size=4)
        ty, tx = rng.integers(0, H - th + 1), rng.integers(0, W - tw + 1)
        ry, rx = rng.integers(0, H - rh + 1), rng.integers(0, W - rw + 1)
        boxes.append((ty * W + tx, W, th, tw, ry * W + rx, W, rh, rw))
    boxes.append((0, W, 12, 12, 10 * W + 30, W, 12, 12))
    boxes.append((40 * W, W, 12, 12, 40 * W, W, 12, 12))
    boxes.append((40 * W, W, 12, 12, 0, W, 12, 12))
    boxes.append((0, W, 12, 12, 0, W, 0, 5))          # empty region
    desc = np.array(boxes, dtype=np.int64)
    want = np.empty(len(desc))
    for p, d in enumerate(desc):
        if d[6] == 0 or d[7] == 0:
            want[p] = np.nan
            continue
        ty, tx = divmod(int(d[0]), W)
        ry, rx = divmod(int(d[4]), W)
        want[p] = oracle.cross_correlate_2d_spectral(left[ty:ty + d[2], tx:tx + d[3]],
                                                     right[ry:ry + d[6], rx:rx + d[7]], want='max')
    dl, dr, dd = nv.to_device(left), nv.to_device(right), nv.to_device(desc)
    for precision, tol in ((nv.ACC_F32, SURFACE_TOL), (nv.ACC_F64, 1e-11)):
        out = nv.empty(len(desc), torch.float64)
        arg = nv.empty(len(desc), torch.int64)
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), len(desc), 40, 1e-4, 1e-9,
                precision, nv.ptr(out), nv.ptr(arg), None, None, nv.stream_ptr())
        assert_surface_close(out.cpu().numpy(), want, tol, f'pairs precision {precision}')


@pytest.mark.parametrize('max_side', [8, 12, 16, 20, 27, 31, 32, 40, 46, 56, 60, 64])
def test_scan_pairs_shared_wavefronts_mixed_sizes(mods, oracle, max_side):
    '''Boxes of every size up to max_side in one call, float32 and float64 accumulation
    against the oracle.  The sides pick the float32 kernel: four pairs per wavefront on the
    matrix cores (up to 16; step counts taken from the largest pair of the wavefront), the
    vector kernel (20, 40), one pair per wavefront (27, 31, 32) or per workgroup (46: three
    strips of 16 columns, 56 - 64: four) on the matrix cores.'''
    import torch
    nv = mods['n']
    rng = np.random.default_rng(max_side)
    H, W = (48, 80) if max_side <= 32 else (80, 112)
    left, right = rng.random((H, W)), rng.random((H, W))
    boxes = []
    for _ in range(400 if max_side <= 32 else 120):
        th, tw, rh, rw = rng.integers(1, max_side + 1, size=4)
        ty, tx = rng.integers(0, H - th + 1), rng.integers(0, W - tw + 1)
        ry, rx = rng.integers(0, H - rh + 1), rng.integers(0, W - rw + 1)
        boxes.append((ty * W + tx, W, th, tw, ry * W + rx, W, rh, rw))
    desc = np.array(boxes, dtype=np.int64)
    want = np.array([oracle.cross_correlate_2d_spectral(
        left[d[0] // W:d[0] // W + d[2], d[0] % W:d[0] % W + d[3]],
        right[d[4] // W:d[4] // W + d[6], d[4] % W:d[4] % W + d[7]], want='max') for d in desc])
    dl, dr, dd = nv.to_device(left), nv.to_device(right), nv.to_device(desc)
    for precision, tol in ((nv.ACC_F32, SURFACE_TOL), (nv.ACC_F64, 1e-11)):
        out = nv.empty(len(desc), torch.float64)
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), len(desc), max_side, 1e-4, 1e-9,
                precision, nv.ptr(out), None, None, None, nv.stream_ptr())
        assert_surface_close(out.cpu().numpy(), want, tol, f'pairs up to {max_side}, precision {precision}')


def test_scan_pairs_random_boxes_f32_against_f64(mods):
    '''Random batches of boxes (sides 1..64: every float32 kernel and staging mode;
    templates larger than regions, empty templates, NaN pixels, constant patches): the float32 kernel's surfaces against the
    float64 kernel's, same NaN pattern, values within 2e-5.'''
    import torch
    nv = mods['n']
    rng = np.random.default_rng(11)
    H, W = 96, 128
    for trial in range(48):
        left, right = rng.random((H, W)), rng.random((H, W))
        if trial % 3 == 0:
            left[rng.random((H, W)) < 0.01] = np.nan
            right[rng.random((H, W)) < 0.01] = np.nan
        if trial % 5 == 0:
            right[10:40, 20:70] = 0.25
            left[50:70, 30:60] = 0.5
        # the largest side picks the kernel (launch_pairs): matrix cores up to 16, 21..32, 45..48, 55..64
        max_side = int(rng.integers(2, 17)) if trial % 2 == 0 else int(rng.integers(17, 65))
        n = int(rng.integers(1, 400 if max_side <= 16 else 60))
        th = rng.integers(0 if trial % 7 == 0 else 1, max_side + 1, size=n)
        tw, rh, rw = (rng.integers(1, max_side + 1, size=n) for _ in range(3))
        ty, ry = (rng.integers(0, H - max_side, size=n) for _ in range(2))
        tx, rx = (rng.integers(0, W - max_side, size=n) for _ in range(2))
        desc = np.stack([ty * W + tx, np.full(n, W), th, tw, ry * W + rx, np.full(n, W), rh, rw], axis=1).astype(np.int64)
        offsets = np.concatenate([[0], np.cumsum(rh * rw)]).astype(np.int64)
        dl, dr, dd, do = nv.to_device(left), nv.to_device(right), nv.to_device(desc), nv.to_device(offsets[:-1].copy())
        got = {}
        for precision in (nv.ACC_F32, nv.ACC_F64):
            out, arg = nv.empty(n, torch.float64), nv.empty(n, torch.int64)
            surf = nv.empty(int(offsets[-1]), torch.float64)
            nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, max_side, 1e-4, 1e-9, precision,
                    nv.ptr(out), nv.ptr(arg), nv.ptr(surf), nv.ptr(do), nv.stream_ptr())
            got[precision] = (out.cpu().numpy(), surf.cpu().numpy())
        (m32, s32), (m64, s64) = got[nv.ACC_F32], got[nv.ACC_F64]
        assert np.array_equal(np.isnan(s32), np.isnan(s64)), f'trial {trial}: NaN pattern of the surfaces'
        assert np.array_equal(np.isnan(m32), np.isnan(m64)), f'trial {trial}: NaN pattern of the maxima'
        if np.isfinite(s64).any():
            assert np.nanmax(np.abs(s32 - s64)) < 2e-5, f'trial {trial}, sides up to {max_side}'
            assert np.nanmax(np.abs(m32 - m64)) < 2e-5


@pytest.mark.parametrize('max_side', [8, 16, 20, 24, 32, 40, 48, 64])
def test_scan_pairs_empty_boxes_in_every_kernel(mods, max_side):
    """All-zero descriptors (what svamd_scan_grid leaves for pairs its tables never name)
    and empty regions between ordinary pairs: NaN and index -1 for them in every float32
    kernel and in the float64 kernel, the neighbours' results untouched."""
    import torch
    nv = mods['n']
    rng = np.random.default_rng(100 + max_side)
    H, W = 80, 96
    left, right = rng.random((H, W)), rng.random((H, W))
    m = max_side
    boxes = []
    for i in range(600):
        y, x = int(rng.integers(0, H - m + 1)), int(rng.integers(0, W - m + 1))
        if i % 5 == 1:
            boxes.append((0,) * 8)
        elif i % 5 == 3:
            boxes.append((y * W + x, W, m, m, y * W + x, W, 0, m) if i % 2 else (y * W + x, W, m, m, y * W + x, W, m, 0))
        else:
            boxes.append((y * W + x, W, m, m, y * W + x, W, m, m))
    desc = np.array(boxes, dtype=np.int64)
    empty = (desc[:, 6] == 0) | (desc[:, 7] == 0)
    dl, dr, dd = nv.to_device(left), nv.to_device(right), nv.to_device(desc)
    outs = {}
    for precision in (nv.ACC_F32, nv.ACC_F64):
        out, arg = nv.empty(len(desc), torch.float64), nv.empty(len(desc), torch.int64)
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), len(desc), max_side, 1e-4, 1e-9,
                precision, nv.ptr(out), nv.ptr(arg), None, None, nv.stream_ptr())
        outs[precision] = (out.cpu().numpy(), arg.cpu().numpy())
    for precision, (val, arg) in outs.items():
        assert np.isnan(val[empty]).all() and np.all(arg[empty] == -1), precision
        assert np.isfinite(val[~empty]).all(), precision
    assert np.array_equal(outs[nv.ACC_F32][1], outs[nv.ACC_F64][1])
    assert_surface_close(outs[nv.ACC_F32][0], outs[nv.ACC_F64][0], SURFACE_TOL, 'pairs next to empty boxes')


@pytest.mark.parametrize('max_side', [12, 16, 24, 32, 48, 64])
def test_scan_pairs_narrow_pair_after_nan_pairs(mods, max_side):
    '''A pair clipped to a few columns shares its wavefront with full-size pairs and
    walks their (wider) column range (sides up to 16), or walks whole tiles of the matrix
    cores (above): what it reads beyond its own staging must be
    zeros, not what an earlier pair left in the LDS.  A first call fills every slot
    with NaN pixels; the second call's results must be finite and equal to the
    float64 kernel's (which gives every pair a workgroup of its own).'''
    import torch
    nv = mods['n']
    rng = np.random.default_rng(5 + max_side)
    H, W = 64, 64
    poison = np.full((H, W), np.nan)
    img_l, img_r = rng.random((H, W)), rng.random((H, W))
    m = max_side
    full = (0, W, m, m, 0, W, m, m)
    ndesc = np.array([full] * 8192, dtype=np.int64)
    boxes = []
    for i in range(8192):
        y, x = int(rng.integers(0, H - m + 1)), int(rng.integers(0, W - m + 1))
        if i % 4 == 1:      # region clipped to a few columns / rows
            boxes.append((y * W + x, W, m, m, y * W + W - 4, W, m, 4))
        elif i % 4 == 2:
            boxes.append((y * W + x, W, m, m, (H - 3) * W + x, W, 3, m))
        else:
            boxes.append((y * W + x, W, m, m, y * W + x, W, m, m))
    desc = np.array(boxes, dtype=np.int64)
    dp, dl, dr = nv.to_device(poison), nv.to_device(img_l), nv.to_device(img_r)
    dn, dd = nv.to_device(ndesc), nv.to_device(desc)
    outs = {}
    for precision in (nv.ACC_F32, nv.ACC_F64):
        junk = nv.empty(len(ndesc), torch.float64)
        nv.call('svamd_scan_pairs', nv.ptr(dp), nv.ptr(dp), nv.ptr(dn), len(ndesc), max_side, 1e-4, 1e-9,
                nv.ACC_F32, nv.ptr(junk), None, None, None, nv.stream_ptr())
        out = nv.empty(len(desc), torch.float64)
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), len(desc), max_side, 1e-4, 1e-9,
                precision, nv.ptr(out), None, None, None, nv.stream_ptr())
        outs[precision] = out.cpu().numpy()
    assert np.isfinite(outs[nv.ACC_F64]).all()
    assert np.isfinite(outs[nv.ACC_F32]).all(), 'stale LDS reached a narrow pair'
    assert_surface_close(outs[nv.ACC_F32], outs[nv.ACC_F64], SURFACE_TOL, 'narrow pairs after NaN pairs')


# ---- dense band scan ---------------------------------------------------------------
def synthetic_pair(h, w, seed, max_shift):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h, w + 2 * max_shift)).astype(np.float64)
    k = np.array([1, 2, 1]) / 4
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), 1, base)
    base = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    left = base[:, max_shift:max_shift + w].copy()
    right = np.empty_like(left)
    for y0 in range(0, h, 16):
        d = int(rng.integers(-max_shift, max_shift + 1))
        right[y0:y0 + 16] = base[y0:y0 + 16, max_shift - d:max_shift - d + w]
    right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, size=right.shape), 0, 255).astype(np.uint8)
    return left, right


@pytest.mark.parametrize('win,h,w,md', [(31, 70, 300, 24), (31, 47, 140, 128), (15, 40, 90, 10),
                                        (7, 33, 64, 5), (11, 30, 530, 3), (21, 50, 77, 9), (5, 20, 70, 4),
                                        (10, 31, 150, 12), (12, 40, 90, 7), (20, 45, 200, 30), (24, 50, 310, 16)])
def test_band_scan_vs_oracle(mods, oracle, win, h, w, md):
    left, right = synthetic_pair(h, w, win * 1000 + md, min(md, 12))
    left[5:5 + win, 3:3 + win] = 200                     # a constant window
    b = mods['b']
    disp, val, scores = b.band_scan(left, right, win, win, md, want_scores=True)
    odisp, oval, oscores = oracle.band_scan(left, right, win, win, md, want_scores=True)
    assert_surface_close(scores, oscores, 1e-12, 'band scores')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'band values')
    # image_scan's rule
    for thr in (0.0, 0.7):
        disp, val = b.band_scan(left / 255.0, right / 255.0, win, win, md, correlation_threshold=thr)
        odisp, oval = oracle.band_scan(left, right, win, win, md, mode='fold', threshold=thr)
        assert np.array_equal(disp, odisp)
        assert_surface_close(val, oval, 1e-12, 'band fold values')


@pytest.mark.parametrize('win_w', list(range(3, 33)))
def test_band_scan_every_unrolled_width(mods, oracle, win_w):
    """Each window width from 3 to 32 has its own fully unrolled kernel for 8-bit images:
    score volume, disparities and the selection rule against the oracle, with a window
    height that is not the width."""
    win_h = 9 + win_w % 5
    left, right = synthetic_pair(36, 170, 500 + win_w, 7)
    b = mods['b']
    disp, val, scores = b.band_scan(left, right, win_w, win_h, 11, want_scores=True)
    odisp, oval, oscores = oracle.band_scan(left, right, win_w, win_h, 11, want_scores=True)
    assert_surface_close(scores, oscores, 1e-12, f'band scores, width {win_w}')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'band values')
    disp, val = b.band_scan(left, right, win_w, win_h, 11, correlation_threshold=0.6)
    odisp, oval = oracle.band_scan(left, right, win_w, win_h, 11, mode='fold', threshold=0.6)
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'band fold values')


@pytest.mark.parametrize('ww,wh,h,w,md,stride', [(31, 31, 70, 300, 24, 1), (31, 31, 47, 140, 128, 1),
                                                 (15, 15, 40, 90, 10, 1), (7, 9, 33, 64, 5, 1), (12, 12, 90, 530, 17, 1),
                                                 (25, 25, 120, 600, 9, 2), (20, 24, 75, 270, 40, 3), (5, 60, 90, 280, 6, 1),
                                                 (32, 32, 64, 64, 8, 1), (1, 1, 9, 40, 3, 1)])
def test_band_scan_box_sums_vs_oracle_and_products(mods, oracle, ww, wh, h, w, md, stride):
    """The box-sum kernel (per-offset summed-area tables of the product image): score volume,
    disparities and selection against the oracle, and every output equal bit for bit to
    the multiply-add kernel's, in both selection modes."""
    left, right = synthetic_pair(h, w, ww * 100 + wh + md, min(md, 12))
    left[5:5 + wh, 3:3 + ww] = 200                       # a constant window
    b = mods['b']
    disp, val, scores = b.band_scan(left, right, ww, wh, md, row_stride=stride, want_scores=True, method='sums')
    odisp, oval, oscores = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, want_scores=True)
    assert_surface_close(scores, oscores, 1e-12, 'box-sum band scores')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'box-sum band values')
    pdisp, pval, pscores = b.band_scan(left, right, ww, wh, md, row_stride=stride, want_scores=True)
    assert np.array_equal(disp, pdisp) and np.array_equal(val, pval, equal_nan=True)
    assert np.array_equal(scores, pscores, equal_nan=True)
    for thr in (0.0, 0.6):
        disp, val = b.band_scan(left, right, ww, wh, md, correlation_threshold=thr, row_stride=stride, method='sums')
        odisp, oval = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, mode='fold', threshold=thr)
        assert np.array_equal(disp, odisp)
        assert_surface_close(val, oval, 1e-12, 'box-sum band fold values')
        pdisp, pval = b.band_scan(left, right, ww, wh, md, correlation_threshold=thr, row_stride=stride)
        assert np.array_equal(disp, pdisp) and np.array_equal(val, pval, equal_nan=True)


@pytest.mark.parametrize('ww,wh,h,w,md,stride', [(40, 40, 70, 330, 20, 1), (45, 44, 60, 300, 128, 1),
                                                 (60, 40, 80, 280, 12, 2), (50, 50, 75, 310, 9, 1), (128, 32, 40, 400, 7, 1)])
def test_band_scan_box_sums_large_windows(mods, oracle, ww, wh, h, w, md, stride):
    """Windows beyond the multiply-add kernel's 1024 pixels (its float32 sums would not be
    exact) run by box sums, whose integer sums are: the shipped configurations' 40 x 40 and
    50 x 50 windows among them.  Score volume, disparities and selection against the oracle."""
    left, right = synthetic_pair(h, w, ww + wh + md, min(md, 12))
    b = mods['b']
    with pytest.raises(mods['n'].NativeError, match='exceeds 1024'):
        b.band_scan(left, right, ww, wh, md, row_stride=stride)
    disp, val, scores = b.band_scan(left, right, ww, wh, md, row_stride=stride, want_scores=True, method='sums')
    odisp, oval, oscores = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, want_scores=True)
    assert_surface_close(scores, oscores, 1e-12, 'box-sum band scores, large window')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'box-sum band values, large window')
    disp, val = b.band_scan(left, right, ww, wh, md, correlation_threshold=0.5, row_stride=stride, method='sums')
    odisp, oval = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, mode='fold', threshold=0.5)
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'box-sum band fold values, large window')


@pytest.mark.parametrize('method', ['products', 'sums'])
def test_band_scan_column_stride(mods, oracle, method):
    """Windows at every third column and every second row (a window_overlap below the dense
    0.96): the same windows, offsets and selections as the oracle's strided scan."""
    left, right = synthetic_pair(60, 200, 33, 6)
    b = mods['b']
    disp, val, scores = b.band_scan(left, right, 15, 15, 9, row_stride=2, column_stride=3, want_scores=True,
                                    method=method)
    odisp, oval, oscores = oracle.band_scan(left, right, 15, 15, 9, stride_x=3, stride_y=2, want_scores=True)
    assert disp.shape == odisp.shape and np.array_equal(disp, odisp)
    assert_surface_close(scores, oscores, 1e-12, 'strided band scores')
    assert_surface_close(val, oval, 1e-12, 'strided band values')
    disp, val = b.band_scan(left, right, 15, 15, 9, correlation_threshold=0.5, row_stride=2, column_stride=3,
                            method=method)
    odisp, oval = oracle.band_scan(left, right, 15, 15, 9, stride_x=3, stride_y=2, mode='fold', threshold=0.5)
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'strided band fold values')


def test_band_scan_takes_the_loaders_output(mods):
    '''`downsample(image, 1)` scales 8-bit levels by the product with 1 / 255 (not the
    quotient by 255); the band scan recognises both and scans the same integers.'''
    left, right = synthetic_pair(60, 150, 77, 6)
    b, u = mods['b'], mods['u']
    want = b.band_scan(left, right, 15, 15, 8)
    for scaled in ((u.downsample(left, 1), u.downsample(right, 1)), (left / 255.0, right / 255.0)):
        got = b.band_scan(scaled[0], scaled[1], 15, 15, 8)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)
    # anything else is scanned as the floating-point image it is: same disparities here
    # (a rescaling does not change a normalised correlation), values to float32 accuracy
    got = b.band_scan(left / 254.0, right / 254.0, 15, 15, 8)
    assert np.array_equal(got[0], want[0])
    assert_surface_close(got[1], want[1], SURFACE_TOL, 'rescaled pair')


def test_band_scan_row_stride_and_sharding(mods, oracle):
    left, right = synthetic_pair(90, 200, 5, 8)
    b = mods['b']
    odisp, oval = oracle.band_scan(left, right, 31, 31, 16, stride_y=3)
    disp, val = b.band_scan(left, right, 31, 31, 16, row_stride=3)
    assert np.array_equal(disp, odisp)
    sc = b.BandScanner(90, 200, 31, 31, 16)
    sc.upload(left, right)
    sc.disp.fill_(-7)
    sc.run(rows=(10, 37))
    part = sc.disp.cpu().numpy()
    full, _ = oracle.band_scan(left, right, 31, 31, 16)
    assert np.array_equal(part[10:37], full[10:37])
    assert np.all(part[:10] == -7) and np.all(part[37:] == -7)


def test_band_scan_matches_reference_band_row(mods):
    '''Row 0 of the reference's spatial NCC of a 31-row band = the band scan's
    scores (golden g3).'''
    g = golden('g3_spatial_band')
    b = mods['b']
    for i in range(len(g['templates'])):
        left = np.zeros((31, 287), np.uint8)
        left[:, 128:159] = g['templates'][i]
        disp, val, scores = b.band_scan(left, g['bands'][i], 31, 31, 128, want_scores=True)
        assert_surface_close(scores[0, 128], g['surfaces'][i][0, :257], 1e-12, 'band row 0')
        assert disp[0, 128] == int(np.nanargmax(g['surfaces'][i][0, :257])) - 128


def test_g14_band_fold_matches_reference_rule(mods):
    '''Selection-rule mode of the band kernel (the benchmark's mode) against the
    reference's sequence_scan run over its spatial correlator.'''
    from helpers import check_band_fold_against_g14
    b = mods['b']
    check_band_fold_against_g14(
        lambda left, right, win, md, thr: b.band_scan(left / 255.0, right / 255.0, win, win, md,
                                                      correlation_threshold=thr))


def smooth_float_pair(h, w, seed, shift, contrast=1.0, offset=0.0):
    '''A float64 pair in the style of a downsampled photograph: smooth texture on top of
    large-scale brightness changes (window means far from the image mean), the right
    image displaced by `shift` plus a little noise.'''
    rng = np.random.default_rng(seed)
    base = rng.random((h, w + 2 * abs(shift) + 8))
    for _ in range(2):
        base = (np.roll(base, 1, 1) + 2 * base + np.roll(base, -1, 1)) / 4
        base = (np.roll(base, 1, 0) + 2 * base + np.roll(base, -1, 0)) / 4
    yy, xx = np.mgrid[0:h, 0:base.shape[1]]
    base = offset + contrast * (0.35 * base + 0.3 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 0.3)
    left = base[:, abs(shift):abs(shift) + w].copy()
    right = base[:, abs(shift) - shift:abs(shift) - shift + w] + contrast * 0.004 * rng.standard_normal((h, w))
    return left, right


@pytest.mark.parametrize('win,h,w,md,stride', [(31, 80, 420, 24, 1), (25, 70, 300, 40, 2), (12, 40, 280, 9, 1),
                                               (9, 33, 90, 5, 1), (20, 50, 200, 14, 1), (24, 60, 330, 8, 2),
                                               (5, 30, 100, 6, 1), (7, 30, 100, 6, 1), (10, 30, 150, 6, 1),
                                               (11, 30, 100, 6, 2), (15, 40, 100, 6, 1), (16, 40, 100, 6, 1),
                                               (21, 40, 130, 6, 1), (13, 40, 130, 6, 1)])
def test_band_scan_float_images_vs_oracle(mods, oracle, win, h, w, md, stride):
    '''Floating-point images (no 8-bit levels: what downsample() returns for factors above
    1) through svamd_band_scan_f64: disparities exactly as the float64 oracle in both
    selection modes, values and the score volume to the float32 bar.'''
    b = mods['b']
    left, right = smooth_float_pair(h, w, win * 7 + md, 5)
    disp, val, scores = b.band_scan(left, right, win, win, md, row_stride=stride, want_scores=True)
    odisp, oval, oscores = oracle.band_scan(left, right, win, win, md, stride_y=stride, want_scores=True)
    assert_surface_close(scores, oscores, SURFACE_TOL, 'float band scores')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, SURFACE_TOL, 'float band values')
    assert (disp == 5).mean() > 0.5
    for thr in (0.5, 0.97):
        disp, val = b.band_scan(left, right, win, win, md, correlation_threshold=thr, row_stride=stride)
        odisp, oval = oracle.band_scan(left, right, win, win, md, stride_y=stride, mode='fold', threshold=thr)
        assert np.array_equal(disp, odisp), thr
        assert_surface_close(val, oval, SURFACE_TOL, 'float band fold values')


def test_band_scan_float_images_hard_cases(mods, oracle):
    '''What float32 alone would get wrong: a pair far from zero with little contrast
    (1000 +- 0.02), near-ties between neighbouring offsets of a very smooth image
    (decided by the float64 second look), an exactly constant patch (NaN scores, as the
    reference's 0 / 0) and a nearly constant one (finite).'''
    b = mods['b']
    left, right = smooth_float_pair(60, 260, 3, -4, contrast=0.05, offset=1000.0)
    left[10:50, 100:150] = 1000.5
    left[12:44, 180:240] = 1000.25 + 1e-7 * np.random.default_rng(0).standard_normal((32, 60))
    for mode, thr in (('argmax', None), ('fold', 0.9)):
        disp, val, scores = b.band_scan(left, right, 15, 15, 12, correlation_threshold=thr, want_scores=True)
        odisp, oval, oscores = oracle.band_scan(left, right, 15, 15, 12, mode=mode, threshold=thr or 0.0,
                                                want_scores=True)
        assert np.isnan(oscores[12:30, 102:130]).all() and np.isfinite(oscores[14:28, 182:220]).all()
        assert_surface_close(scores, oscores, SURFACE_TOL, 'hard float band scores')
        assert np.array_equal(disp, odisp), mode
        assert_surface_close(val, oval, SURFACE_TOL, 'hard float band values')
    # a very smooth image: neighbouring offsets score within the guard band of each other
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:50, 0:300]
    smooth = 0.5 + 0.4 * np.sin(xx / 90.0 + yy / 70.0) + 1e-3 * rng.standard_normal((50, 300))
    disp, val = b.band_scan(smooth, np.roll(smooth, 1, axis=1), 21, 21, 6)
    odisp, oval = oracle.band_scan(smooth, np.roll(smooth, 1, axis=1), 21, 21, 6)
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, SURFACE_TOL, 'smooth float band values')


def test_band_scan_rejects_unsupported(mods):
    b = mods['b']
    with pytest.raises(TypeError):
        b.band_scan(np.zeros((40, 40), dtype=complex), np.zeros((40, 40)), 7, 7, 3)
    with pytest.raises(mods['n'].NativeError, match='exceeds 1024'):
        b.band_scan(np.zeros((80, 80), np.uint8), np.zeros((80, 80), np.uint8), 33, 32, 3)
    with pytest.raises(mods['n'].NativeError, match='exceeds 4096'):
        b.band_scan(np.zeros((80, 80), np.uint8), np.zeros((80, 80), np.uint8), 70, 64, 3, method='sums')
    with pytest.raises(mods['n'].NativeError, match='does not fit the LDS tile'):
        b.band_scan(np.zeros((230, 280), np.uint8), np.zeros((230, 280), np.uint8), 5, 200, 6)
    # the box-sum kernel keeps a tile's table and the right band under all offsets in LDS
    with pytest.raises(mods['n'].NativeError, match='does not fit the box-sum tile'):
        b.band_scan(np.zeros((230, 280), np.uint8), np.zeros((230, 280), np.uint8), 5, 200, 6, method='sums')
    with pytest.raises(mods['n'].NativeError, match='does not fit the box-sum tile'):
        b.band_scan(np.zeros((64, 300), np.uint8), np.zeros((64, 300), np.uint8), 31, 31, 40000, method='sums')
    with pytest.raises(ValueError):
        b.band_scan(np.zeros((40, 40), np.uint8), np.zeros((40, 40), np.uint8), 7, 7, 3, method='fastest')


@pytest.mark.parametrize('ww,wh,h,w,md,stride', [(8, 8, 40, 90, 10, 1), (12, 12, 50, 300, 17, 1), (25, 25, 60, 140, 9, 2),
                                                 (40, 25, 47, 330, 12, 1), (3, 5, 20, 70, 4, 1), (1, 1, 9, 40, 3, 1),
                                                 (2, 30, 64, 64, 8, 1), (5, 60, 90, 280, 6, 1), (64, 16, 40, 400, 130, 1)])
def test_band_scan_any_window_width(mods, oracle, ww, wh, h, w, md, stride):
    '''Window widths without a fixed-width kernel go through the run-time-width
    kernel: same scores, disparities and selection as the oracle.'''
    left, right = synthetic_pair(h, w, ww * 100 + wh, min(md, 12))
    b = mods['b']
    disp, val, scores = b.band_scan(left, right, ww, wh, md, row_stride=stride, want_scores=True)
    odisp, oval, oscores = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, want_scores=True)
    assert_surface_close(scores, oscores, 1e-12, 'band scores')
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'band values')
    disp, val = b.band_scan(left, right, ww, wh, md, correlation_threshold=0.4, row_stride=stride)
    odisp, oval = oracle.band_scan(left, right, ww, wh, md, stride_y=stride, mode='fold', threshold=0.4)
    assert np.array_equal(disp, odisp)
    assert_surface_close(val, oval, 1e-12, 'band fold values')


# ---- FFT route (power-of-two shapes) ------------------------------------------------
def test_fft_route_golden_rand_pairs(mods):
    for n in (256, 1024):
        g = golden(f'g4_spectral_rand{n}')
        t, r, idx = rand_pair(n)
        out = mods['s'].cross_correlate_2d_spectral(t, r)
        assert out.dtype == np.float64 and out.shape == (n, n)
        assert_surface_close(out.ravel()[idx], g['sample_val'], SURFACE_TOL, f'rand{n}')
        assert int(np.argmax(out)) == int(g['argmax'])
        assert abs(out.max() - float(g['max_val'])) < SURFACE_TOL


@pytest.mark.parametrize('tshape,rshape', [((128, 128), (128, 128)), ((64, 256), (64, 256)),
                                           ((128, 32), (128, 32)), ((100, 190), (128, 256)), ((40, 70), (128, 256)),
                                           ((300, 200), (256, 128)), ((100, 8), (8, 512))])
def test_fft_route_vs_oracle(mods, oracle, tshape, rshape):
    rng = np.random.default_rng(sum(tshape) + sum(rshape))
    t = rng.random(tshape).astype(np.float32)
    r = (rng.random(rshape) * 255).astype(np.float32)
    got = mods['s'].cross_correlate_2d_spectral(t, r)
    want = oracle.cross_correlate_2d_spectral_fft(t.astype(np.float64), r.astype(np.float64))
    assert_surface_close(got, want, SURFACE_TOL, f'fft {tshape} vs {rshape}')
    assert int(np.argmax(got)) == int(np.argmax(want))


def test_fft_route_batched_and_branches(mods, oracle):
    rng = np.random.default_rng(99)
    t = rng.random((5, 128, 128)).astype(np.float32)
    r = rng.random((5, 128, 128)).astype(np.float32)
    t[1] = 0.5                       # homogeneous template, different mean -> zeros
    t[2] = 0.25
    r[2] = 0.25                      # homogeneous template, same mean -> ones
    r[3] = 0.75                      # constant region -> NaN
    r[4] = np.roll(t[4], (5, 9), axis=(0, 1))
    out = mods['s'].cross_correlate_2d_spectral_batched(t, r)
    assert out.dtype == np.float32 and out.shape == (5, 128, 128)
    want0 = oracle.cross_correlate_2d_spectral_fft(t[0].astype(np.float64), r[0].astype(np.float64))
    assert_surface_close(out[0], want0, SURFACE_TOL, 'batch[0]')
    assert np.all(out[1] == 0.0)
    assert np.all(out[2] == 1.0)
    assert np.isnan(out[3]).all()
    assert np.unravel_index(np.argmax(out[4]), out[4].shape) == (5, 9)
    assert abs(out[4].max() - 1.0) < SURFACE_TOL


def test_fft_route_1024_shapes_and_offsets(mods, oracle):
    '''1024 x 1024 fast path: large DC offset, template smaller / larger than the
    region (zero-padding, cropping with full-template statistics).'''
    rng = np.random.default_rng(1024)
    r = (rng.random((1024, 1024)) * 40 + 180).astype(np.float32)        # mean >> deviation
    for tshape in ((1024, 1024), (700, 900), (1100, 1030)):
        t = (rng.random(tshape) * 40 + 100).astype(np.float32)
        if tshape == (1024, 1024):
            t[:, :] += np.linspace(0, 60, 1024, dtype=np.float32)[:, None]   # first row unrepresentative
        got = mods['s'].cross_correlate_2d_spectral(t, r)
        want = oracle.cross_correlate_2d_spectral_fft(t.astype(np.float64), r.astype(np.float64))
        assert_surface_close(got, want, SURFACE_TOL, f'fft1024 {tshape}')


# ---- displacement grids ------------------------------------------------------------------
@pytest.mark.parametrize('tag', ['scan_config', 'portal', 'multipass'])
def test_g10_depth_grids(mods, tag):
    '''Scan + grid accumulation on the device against the grids the reference's
    stereo driver saved: identical, bit for bit.'''
    import depth_grids
    g = golden('g10_cone_depth_grids')
    im = golden('g6_cone_images')
    left, right = im['left'] / 255.0, im['right'] / 255.0
    totals = depth_grids.DepthGrids(left.shape)
    for seq in json.loads(str(g[f'{tag}_config'])):
        totals.add_sequence(quiet(mods['v'].sequence_scan, left, right, seq))
    for got, key in zip(totals.average(), ('depth_grid', 'dp_x_grid', 'dp_y_grid', 'contributions_grid')):
        assert np.array_equal(got, g[f'{tag}_{key}']), key


@pytest.mark.parametrize('shape,n,max_side', [((70, 90), 500, 40), ((300, 520), 20000, 64), ((33, 700), 3000, 300),
                                              ((600, 40), 1, 700)])
def test_accumulate_boxes_random(mods, shape, n, max_side):
    '''Random overlapping boxes with irrational values: the device sums every
    pixel's boxes in list order, so it equals the sequential numpy painting
    exactly; two calls accumulate into the same totals.'''
    import depth_grids
    rng = np.random.default_rng(n)
    h, w = shape
    x0 = rng.integers(-10, w, size=n)
    y0 = rng.integers(-10, h, size=n)
    boxes = np.stack([np.maximum(x0, 0), np.minimum(x0 + rng.integers(0, max_side, size=n), w),
                      np.maximum(y0, 0), np.minimum(y0 + rng.integers(0, max_side, size=n), h)], axis=1)
    values = rng.standard_normal((n, 3))
    values[:, 2] = np.abs(values[:, 2]) * (rng.random(n) < 0.7)
    want = [np.zeros(shape) for _ in range(4)]
    totals = depth_grids.DepthGrids(shape)
    for part in (slice(0, n // 2), slice(n // 2, n)):
        seq = [np.zeros(shape) for _ in range(4)]
        for (bx0, bx1, by0, by1), (vx, vy, vm) in zip(boxes[part], values[part]):
            if bx0 < bx1 and by0 < by1:
                seq[0][by0:by1, bx0:bx1] += vm
                seq[1][by0:by1, bx0:bx1] += vx
                seq[2][by0:by1, bx0:bx1] += vy
                if vm > 0:
                    seq[3][by0:by1, bx0:bx1] += 1
        for t, q in zip(want, seq):
            t += q
        totals.add_boxes(boxes[part], values[part])
    got = [t.cpu().numpy() for t in (totals.depth, totals.dp_x, totals.dp_y, totals.contributions)]
    for a, b, name in zip(got, want, ('depth', 'dp_x', 'dp_y', 'contributions')):
        assert np.array_equal(a, b), name


@pytest.mark.parametrize('win,h,w,md,stride', [(31, 70, 300, 24, 1), (15, 64, 90, 10, 2), (7, 40, 50, 5, 3)])
def test_band_accumulate_vs_oracle(mods, oracle, win, h, w, md, stride):
    import depth_grids
    left, right = synthetic_pair(h, w, win + md, min(md, 12))
    left[5:5 + win, 3:3 + win] = 200                     # a constant window: no match
    disp, _ = mods['b'].band_scan(left, right, win, win, md, row_stride=stride)
    totals = depth_grids.DepthGrids((h, w))
    totals.add_band_scan(disp, win, win, stride)
    depth, dp_x, contributions = oracle.band_depth_grids(disp, (h, w), win, win, stride)
    assert np.array_equal(totals.depth.cpu().numpy(), depth)
    assert np.array_equal(totals.dp_x.cpu().numpy(), dp_x)
    assert np.array_equal(totals.contributions.cpu().numpy(), contributions)
    assert not totals.dp_y.any()
    avg = totals.average()
    contributions[contributions == 0] = np.inf
    assert np.array_equal(avg[0], depth / contributions) and np.array_equal(avg[3], contributions)


# ---- calibration peak finder ---------------------------------------------------------------
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_g11_find_peaks(mods, tag):
    '''Peak positions of the reference's find_peaks on synthetic dot grids: the
    same peaks at the same positions.'''
    import sv_calibration
    g = golden('g11_find_peaks')
    x, y = sv_calibration.find_peaks(g[f'{tag}_image'], neighborhood_size=10, threshold=0.2)
    assert np.array_equal(x, g[f'{tag}_x']) and np.array_equal(y, g[f'{tag}_y'])


@pytest.mark.parametrize('shape,size,thr', [((40, 56), 10, 0.2), ((33, 70), 3, 0.5), ((64, 64), 7, 0.0),
                                            ((5, 9), 10, 0.1)])
def test_peak_mask_vs_oracle_and_scipy(mods, oracle, shape, size, thr):
    import sv_calibration
    from scipy import ndimage
    rng = np.random.default_rng(size)
    data = rng.random(shape)
    if shape[0] > 14:
        data[10:14, 20:24] = 2.0                     # a plateau of equal maxima
    got = sv_calibration.peak_mask(data, size, thr)
    assert np.array_equal(got, oracle.peak_mask(data, size, thr))
    dmax, dmin = ndimage.maximum_filter(data, size), ndimage.minimum_filter(data, size)
    assert np.array_equal(got, (data == dmax) & ((dmax - dmin) > thr))


# ---- image preparation ---------------------------------------------------------------------
def test_g12_greyscale_with_nan(mods, tmp_path):
    '''The reference's loader on its own RGBA images: identical grey levels and
    NaN pixels, from the file as well as from decoded pixels.'''
    from PIL import Image
    g = golden('g12_greyscale_with_nan')
    for key in ('template', 'rm_template'):
        rgba = g[f'{key}_rgba']
        dev = mods['u']._greyscale_on_device(rgba, nan_alpha=True, want_u8=False)
        assert np.array_equal(dev.cpu().numpy(), g[f'{key}_grey'], equal_nan=True)
        Image.fromarray(rgba, 'RGBA').save(tmp_path / f'{key}.png')
        image = mods['u'].greyscale_with_nan(str(tmp_path / f'{key}.png'))
        assert isinstance(image, np.ndarray) and image.dtype == np.float64
        assert np.array_equal(image, g[f'{key}_grey'], equal_nan=True)
        assert np.array_equal(image.resident.cpu().numpy(), g[f'{key}_grey'], equal_nan=True)
    Image.fromarray(g['template_rgba'][..., :3].copy(), 'RGB').save(tmp_path / 'rgb.png')
    with pytest.raises(ValueError):
        mods['u'].greyscale_with_nan(str(tmp_path / 'rgb.png'))


@pytest.mark.parametrize('mode,channels', [('RGBA', 4), ('RGB', 3), ('LA', 2), ('L', 1)])
def test_greyscale_modes(mods, oracle, tmp_path, mode, channels):
    from PIL import Image
    rng = np.random.default_rng(channels)
    pixels = rng.integers(0, 256, size=(203, 331, channels), dtype=np.uint8)
    if channels == 1:
        pixels = pixels[..., 0]
    Image.fromarray(pixels, mode=mode).save(tmp_path / 'image.png')
    got = mods['u'].greyscale(str(tmp_path / 'image.png'))
    assert got.dtype == np.uint8
    assert np.array_equal(got, oracle.greyscale(pixels))
    assert np.array_equal(got, np.asarray(Image.open(tmp_path / 'image.png').convert('L')))


RESIZE_TOL = 1e-12   # float64; the Gaussian weights come from two different exp()s


@pytest.mark.parametrize('shape,factor,kind', [
    ((64, 80), 2, 'u8'), ((101, 77), 3, 'u8'), ((50, 64), 1, 'u8'), ((40, 200), 4, 'f64'), ((33, 47), 1.5, 'f64'),
    ((50, 300), 7, 'u8'), ((64, 64), 2, 'nan'), ((9, 2000), 1, 'f64'), ((30, 30), 0.5, 'f64'),
    ((1000, 1500), 2, 'u8'), ((700, 900), 5, 'nan'), ((4, 700), 3, 'u8')])
def test_downsample_vs_oracle(mods, oracle, shape, factor, kind):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    if kind == 'u8':
        image = rng.integers(0, 256, size=shape, dtype=np.uint8)
    else:
        image = rng.normal(size=shape) * 50
        if kind == 'nan':
            image[10:12, 20:23] = np.nan
    got = mods['u'].downsample(image, factor)
    want = oracle.downsample(image, factor)
    assert got.shape == want.shape and got.dtype == np.float64
    assert np.array_equal(np.isnan(got), np.isnan(want))
    if factor == 1:
        assert np.array_equal(got, want, equal_nan=True)        # no filter: the same four products
    else:
        m = ~np.isnan(want)
        scale = np.nanmax(np.abs(want))
        assert np.max(np.abs(got[m] - want[m])) <= RESIZE_TOL * scale
    assert np.array_equal(got.resident.cpu().numpy(), np.asarray(got), equal_nan=True)


@pytest.mark.parametrize('shape,out_shape', [((96, 200), (96, 100)), ((120, 64), (40, 64)), ((400, 400), (10, 10)),
                                             ((300, 500), (100, 125))])
def test_downsample_c_abi_shapes(mods, oracle, shape, out_shape):
    '''Output shapes the Python wrapper never asks for: one axis kept (separate
    filter passes), a reduction too large for the fused tile, unequal factors.'''
    nv = mods['n']
    torch = nv.torch_cuda()
    rng = np.random.default_rng(shape[0] + out_shape[1])
    image = rng.normal(size=shape) * 20 + 100
    d_in = nv.to_device(image)
    out = nv.empty(out_shape, torch.float64)
    nbytes = nv.load_library().svamd_downsample_workspace_bytes(*shape)
    ws = nv.workspace(nbytes)
    nv.call('svamd_downsample', nv.ptr(d_in), nv.F64, shape[0], shape[1], out_shape[0], out_shape[1], nv.ptr(out),
            nv.ptr(ws), nbytes, nv.stream_ptr())
    want = oracle.resize(image, out_shape)
    got = out.cpu().numpy()
    assert np.max(np.abs(got - want)) <= RESIZE_TOL * np.max(np.abs(want))


def test_downsample_rejects(mods):
    with pytest.raises(TypeError):
        mods['u'].downsample(np.zeros((8, 8), np.int32), 2)
    with pytest.raises(ValueError):
        mods['u'].downsample(np.zeros((8, 8, 3), np.uint8), 2)
    with pytest.raises(mods['n'].NativeError):
        mods['u'].downsample(np.zeros((600, 8)), 300)


def test_resident_images_feed_the_scan(mods):
    '''Images that come out of the device preparation are scanned from their
    device copies; the results are those of the same values uploaded afresh.'''
    g = golden('g6_cone_images')
    cfg = [{'window_width': 40, 'window_height': 40, 'scheme': [5, 1], 'scheme_shift_size': [10, 10],
            'window_overlap': 0.5, 'correlation_threshold': 0.6}, {'factor': 2}]
    left, right = (mods['u'].downsample(np.ascontiguousarray(g[k]), 2) for k in ('left', 'right'))
    assert left.resident is not None and left[::2].resident is None
    with pytest.raises(ValueError):
        left[0, 0] = 1.0
    a = quiet(mods['v'].sequence_scan, left, right, cfg)
    b = quiet(mods['v'].sequence_scan, np.array(left), np.array(right), cfg)
    assert len(a) == len(b) == 2
    for sa, sb in zip(a, b):
        assert list(sa) == list(sb)
        for wid in sa:
            assert (sa[wid]['dp_x'], sa[wid]['dp_y']) == (sb[wid]['dp_x'], sb[wid]['dp_y'])


# ---- driver scripts --------------------------------------------------------------------
def run_driver(module, argv, monkeypatch, capsys):
    import importlib
    import sys
    monkeypatch.setattr(sys, 'argv', [module] + argv)
    mod = importlib.import_module(module)
    mod.main()
    return capsys.readouterr().out


def test_driver_2d_xcorr(mods, tmp_path, monkeypatch, capsys):
    from PIL import Image
    g = golden('g1_spatial_template_region')
    t = g['template']
    rgba = np.zeros(t.shape + (4,), np.uint8)
    grey = np.nan_to_num(t, nan=0).astype(np.uint8)
    rgba[..., 0] = rgba[..., 1] = rgba[..., 2] = grey
    rgba[..., 3] = np.where(np.isnan(t), 0, 255)
    Image.fromarray(rgba, 'RGBA').save(tmp_path / 'template.PNG')
    Image.fromarray(g['region'], 'L').save(tmp_path / 'region.PNG')
    out = run_driver('test_2d_xcorr', ['--template', 'template.PNG', '--region', 'region.PNG', '--step', '5',
                                       '--images_dir', str(tmp_path), '--no_plot'], monkeypatch, capsys)
    assert 'Maximum cross-correlation at: (290, 150)' in out
    out = run_driver('test_2d_xcorr', ['--spectral', '--template', 'template.PNG', '--region', 'region.PNG',
                                       '--images_dir', str(tmp_path), '--no_plot'], monkeypatch, capsys)
    assert 'Maximum cross-correlation at: (288, 150)' in out


def test_driver_signal_offset(mods, monkeypatch, capsys):
    spatial = run_driver('signal_offset', ['--synthetic', '--no_plot'], monkeypatch, capsys)
    spectral = run_driver('signal_offset', ['--synthetic', '--no_plot', '--spectral'], monkeypatch, capsys)
    for out in (spatial, spectral):
        assert 'Max at: 44136' in out
        assert 'Number of signals shifted: 136' in out


def test_driver_1d_xcorr_and_numba_twins(mods, oracle, monkeypatch, capsys):
    import correlation
    import correlation_numba
    import test_1d_xcorr as drv
    monkeypatch.setattr(correlation, 'PRINT_PROGRESS', False)
    monkeypatch.setattr('sys.argv', ['test_1d_xcorr', '--no_plot'])
    results = dict(drv.main())
    out = capsys.readouterr().out
    for label in ('raw custom correlation', 'vectorised custom correlation', 'spectral correlation',
                  'Numpy correlation'):
        assert f'Time elapsed ({label})' in out
    # the normalised numpy correlation differs from the un-normalised custom one only by a factor
    x = np.arange(0, 4 * np.pi + drv.STEP_SIZE, drv.STEP_SIZE)
    want = oracle.cross_correlate_1d(np.cos(x), -np.sin(x))
    close_1d(results['vectorised custom correlation'], want, 'driver xc')
    close_1d(np.array(results['raw custom correlation']), want, 'driver raw')
    # the "Numba" twins: same values; unequal lengths print and return None
    v = np.random.default_rng(1).standard_normal(300)
    close_1d(correlation_numba.cross_correlate_1d_numba(v, v[::-1].copy()),
             oracle.cross_correlate_1d(v, v[::-1].copy()), 'numba twin')
    assert isinstance(correlation_numba.cross_correlate_1d_raw_numba(list(v), list(v)), list)
    assert correlation_numba.cross_correlate_1d_numba(v, v[:10]) is None
    assert 'Error: Vectors are of unequal length.' in capsys.readouterr().out
    import numba_tests
    numba_tests.main()
    out = capsys.readouterr().out
    assert 'Raw python, with Numba JIT; time elapsed' in out and 'Vectorisation, with Numba JIT; time elapsed' in out


def test_driver_music_patterns(mods, monkeypatch, capsys):
    '''Every snippet of the synthetic recording is found where it was cut out:
    with template = recording (nt samples) and signal = snippet starting at s,
    the circular correlation peaks at lag nt - 1 - s.'''
    import music_patterns_africa as drv
    out = run_driver('music_patterns_africa', ['--synthetic', '--no_plot'], monkeypatch, capsys)
    signal, _, starts = drv.synthetic_recording()
    for name, _, label in drv.SNIPPETS:
        assert f'{label} correlation complete' in out
        assert f'{label} maximum at: {len(signal) - 1 - starts[name]}' in out


def test_driver_image_comparison(mods, tmp_path, monkeypatch, capsys):
    from PIL import Image
    im = golden('g6_cone_images')
    g = golden('g6_cone_scan_config')
    Image.fromarray(im['left'], 'L').save(tmp_path / 'left_cone.tiff')
    Image.fromarray(im['right'], 'L').save(tmp_path / 'right_cone.tiff')
    (tmp_path / 'cfg.json').write_text(str(g['config']))
    out = run_driver('test_image_comparison',
                     ['--images', 'cone', '--config', 'cfg.json', '--hide_depth', '--depth_output', 'depth.json',
                      '--image_dir', str(tmp_path), '--config_dir', str(tmp_path), '--data_dir', str(tmp_path)],
                     monkeypatch, capsys)
    assert 'Time elapsed for sequence 1 scan' in out and 'Data saved to file depth.json' in out
    saved = json.loads((tmp_path / 'depth.json').read_text())
    dp_x = np.array(saved['dp_x_grid'])
    assert dp_x.shape == im['left'].shape
    # windows that moved in the reference's run paint their displacement into the grid
    want = g['seq0_stage0_windows']
    moved = want[(want[:, 5] != 0) | (want[:, 6] != 0)]
    assert len(moved) == 38 and np.count_nonzero(dp_x) > 0
    ref = golden('g10_cone_depth_grids')
    for key in ('depth_grid', 'dp_x_grid', 'dp_y_grid', 'contributions_grid'):
        assert np.array_equal(np.array(saved[key]), ref[f'scan_config_{key}']), key


def test_driver_image_comparison_downsampled(mods, oracle, tmp_path, monkeypatch, capsys):
    '''--ds_factor 2: the driver's images go through the device downsample (no
    scikit-image needed); grids as the oracle's scan of the oracle's downsampled pair
    paints them.'''
    from PIL import Image
    im = golden('g6_cone_images')
    cfg = [[{'window_width': 30, 'window_height': 30, 'scheme': [7, 1], 'scheme_shift_size': [4, 0],
             'window_overlap': 0.5, 'correlation_threshold': 0.8}]]
    Image.fromarray(im['left'], 'L').save(tmp_path / 'left_cone.tiff')
    Image.fromarray(im['right'], 'L').save(tmp_path / 'right_cone.tiff')
    (tmp_path / 'cfg.json').write_text(json.dumps(cfg))
    run_driver('test_image_comparison',
               ['--images', 'cone', '--config', 'cfg.json', '--ds_factor', '2', '--hide_depth', '--depth_output',
                'depth.json', '--image_dir', str(tmp_path), '--config_dir', str(tmp_path), '--data_dir', str(tmp_path)],
               monkeypatch, capsys)
    saved = json.loads((tmp_path / 'depth.json').read_text())
    shape = (im['left'].shape[0] // 2, im['left'].shape[1] // 2)
    left, right = oracle.downsample(im['left'], 2), oracle.downsample(im['right'], 2)
    want = oracle.depth_grids([oracle.sequence_scan(left, right, cfg[0])], shape)
    for got_key, grid in zip(('depth_grid', 'dp_x_grid', 'dp_y_grid', 'contributions_grid'), want):
        got = np.array(saved[got_key])
        assert got.shape == shape
        assert np.array_equal(got, grid), got_key
    assert np.count_nonzero(np.array(saved['dp_x_grid'])) > 0



# ---- process-level behaviour of the binding ---------------------------------------
def test_first_call_of_a_fresh_process_is_the_spectral_correlation():
    '''A process whose very first use of the package is cross_correlate_2d_spectral (no
    torch import of its own, no other call before): the library must come up on the HIP
    runtime PyTorch ships - one runtime in the process - and the call must work.'''
    import subprocess
    import sys
    from helpers import PKG
    code = (
        'import sys; sys.path.insert(0, %r)\n'
        'import numpy as np\n'
        'import correlation_spectral as s\n'
        'rng = np.random.default_rng(0)\n'
        't, r = rng.random((20, 24)), rng.random((20, 24))\n'
        'got = s.cross_correlate_2d_spectral(t, r)\n'
        'tn = (t - t.mean()) / (t.std() * t.size); rn = (r - r.mean()) / r.std()\n'
        'want = np.real(np.fft.ifft2(np.conj(np.fft.fft2(tn)) * np.fft.fft2(rn)))\n'
        'import _native\n'
        'hip = [p for p in _native.hip_runtimes() if "libamdhip64" in p]\n'
        'assert len(hip) == 1, hip\n'
        'assert np.abs(got - want).max() < 1e-12\n'
        'print("fresh ok")\n') % PKG
    done = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and 'fresh ok' in done.stdout, done.stderr[-2000:]


def test_loading_the_library_before_torch_is_refused():
    '''The order that gives the process two HIP runtimes (libsvamd.so mapped before
    PyTorch) is reported by the binding instead of failing at the first launch.'''
    import subprocess
    import sys
    from helpers import PKG
    code = (
        'import ctypes, os, sys; sys.path.insert(0, %r)\n'
        'ctypes.CDLL(os.path.join(%r, "libsvamd.so"))\n'
        'import _native\n'
        'try:\n'
        '    _native.load_library()\n'
        'except RuntimeError as exc:\n'
        '    assert "two HIP runtimes" in str(exc), exc\n'
        '    print("refused")\n') % (PKG, PKG)
    done = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and 'refused' in done.stdout, (done.stdout, done.stderr[-2000:])


def test_entry_points_follow_the_current_torch_stream(mods):
    '''Calls made under torch.cuda.stream(side stream) run on that stream (the binding
    hands the library torch's current stream) and give the same results as on the
    default stream.'''
    import torch
    rng = np.random.default_rng(5)
    t, r = rng.random((24, 30)), rng.random((24, 30))
    v1, v2 = rng.random(3000), rng.random(3000)
    want_2d = mods['s'].cross_correlate_2d_spectral(t, r)
    want_1d = quiet(mods['c'].cross_correlate_1d, v1, v2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert mods['n'].stream_ptr().value == side.cuda_stream
        got_2d = mods['s'].cross_correlate_2d_spectral(t, r)
        got_1d = quiet(mods['c'].cross_correlate_1d, v1, v2)
    side.synchronize()
    assert np.array_equal(got_2d, want_2d) and np.array_equal(got_1d, want_1d)
