'''Parity at the benchmark's full sizes, where the oracle cannot run the whole
workload in seconds: sampled slices against the oracle, and properties that do not
depend on the size (planted disparities, determinism, sharding, shifts, direct dot
products at sampled output positions).  GPU only; everything goes through the C ABI.
'''
import sys

import numpy as np
import pytest

from helpers import ROOT, assert_surface_close

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)

WIN, MAX_DISP, THRESHOLD = 31, 128, 0.5


@pytest.fixture(scope='module')
def dense():
    '''The benchmark's own 4096 x 4096 pair, scanned once in each mode.'''
    import band_scan as bs
    import bench
    left, right = bench.synthetic_pair(4096, 4096, seed=1000)
    scanner = bs.BandScanner(4096, 4096, WIN, WIN, MAX_DISP)
    scanner.upload(left, right)
    scanner.run(THRESHOLD)
    fold = (scanner.disp.cpu().numpy().copy(), scanner.val.cpu().numpy().copy())
    scanner.run(None)
    argmax = (scanner.disp.cpu().numpy().copy(), scanner.val.cpu().numpy().copy())
    return dict(left=left, right=right, scanner=scanner, fold=fold, argmax=argmax)


@pytest.mark.parametrize('row0', [0, 2017, 4066 - 6])
def test_dense_scan_bands_match_the_oracle(dense, oracle, row0):
    '''Six window rows at the top, in the middle and at the bottom of the 4096 x
    4096 scan against the oracle's scan of the image rows those windows cover:
    disparities exactly, correlations to 1e-12.'''
    rows = 6
    crop = slice(row0, row0 + rows + WIN - 1)
    for mode, key, thr in (('fold', 'fold', THRESHOLD), ('argmax', 'argmax', 0.0)):
        odisp, oval = oracle.band_scan(dense['left'][crop], dense['right'][crop], WIN, WIN, MAX_DISP, mode=mode,
                                       threshold=thr)
        disp, val = dense[key]
        assert np.array_equal(disp[row0:row0 + rows], odisp), mode
        assert_surface_close(val[row0:row0 + rows], oval, 1e-12, f'band values, {mode}')


def test_dense_scan_recovers_planted_disparities(dense):
    '''bench.synthetic_pair displaces 256-pixel tiles of the right image by a
    constant each (right[x] = left[x - d] plus noise of +-2 grey levels).  A window
    whose box at offset d lies inside the tile displaced by d has an exact match there
    (near a tile border two tiles can offer one each); the scan must pick one of them.'''
    rng = np.random.default_rng(1000)
    # replay the generator's draws: base image first, then one integer per tile
    rng.integers(0, 256, size=(4096, 4096 + 256), dtype=np.uint8)
    disp, val = dense['argmax']
    nwx = 4096 - WIN + 1
    columns = np.arange(nwx)
    hits = total = 0
    for ty in range(0, 4096, 256):
        rows = slice(ty, min(ty + 256, 4096) - WIN + 1)
        block = disp[rows]
        found = np.zeros(block.shape, dtype=bool)
        offered = np.zeros(nwx, dtype=bool)
        for tx in range(0, 4096, 256):
            d = int(rng.integers(-100, 101))
            inside = (columns + d >= tx) & (columns + d + WIN <= min(tx + 256, 4096))
            offered |= inside
            found[:, inside] |= block[:, inside] == d
        hits += int(found[:, offered].sum())
        total += int(offered.sum()) * block.shape[0]
        assert np.all(val[rows][:, offered] > 0.99)
    assert total > 10_000_000
    assert hits == total, (hits, total)


def test_dense_scan_is_deterministic_and_shards(dense):
    scanner = dense['scanner']
    scanner.run(THRESHOLD)
    again = scanner.disp.cpu().numpy()
    assert np.array_equal(again, dense['fold'][0])
    assert np.array_equal(scanner.val.cpu().numpy(), dense['fold'][1], equal_nan=True)
    # three bands of window rows, as three ranks would scan them
    scanner.disp.fill_(-9)
    nwy = scanner.nwy
    for begin, end in ((0, 1355), (1355, 2711), (2711, nwy)):
        scanner.run(THRESHOLD, rows=(begin, end))
    assert np.array_equal(scanner.disp.cpu().numpy(), dense['fold'][0])


def test_dense_scan_by_box_sums_is_identical(dense):
    """The box-sum kernel (per-offset summed-area tables of the product image) on the whole
    4096 x 4096 pair: disparities and correlations equal to the multiply-add kernel's bit
    for bit, in both selection modes, and for a band of window rows as a rank would scan it."""
    scanner = dense['scanner']
    assert scanner.box_sums_supported()
    scanner.run(THRESHOLD, method='sums')
    assert np.array_equal(scanner.disp.cpu().numpy(), dense['fold'][0])
    assert np.array_equal(scanner.val.cpu().numpy(), dense['fold'][1], equal_nan=True)
    scanner.run(None, method='sums')
    assert np.array_equal(scanner.disp.cpu().numpy(), dense['argmax'][0])
    assert np.array_equal(scanner.val.cpu().numpy(), dense['argmax'][1], equal_nan=True)
    scanner.disp.fill_(-9)
    scanner.run(THRESHOLD, rows=(1355, 2711), method='sums')
    assert np.array_equal(scanner.disp.cpu().numpy()[1355:2711], dense['fold'][0][1355:2711])
    assert (scanner.disp.cpu().numpy()[:1355] == -9).all() and (scanner.disp.cpu().numpy()[2711:] == -9).all()


def test_dense_scan_selection_rule_follows_from_the_argmax(dense):
    '''image_scan's rule keeps the centre unless a candidate beats both the
    threshold and the centre by more than the tolerance: wherever the plain
    arg-max is below the threshold the rule must report no displacement, and a
    reported displacement always carries a correlation above the threshold.'''
    fdisp, fval = dense['fold']
    adisp, aval = dense['argmax']
    low = aval < THRESHOLD
    assert np.all(fdisp[low] == 0)
    moved = fdisp != 0
    assert np.all(fval[moved] > THRESHOLD)
    assert np.all(fval[moved] <= aval[moved] + 1e-15)


def test_batched_fft_1024_shift_and_samples(mods_spectral):
    '''Config 5's shape: 1024 x 1024 pairs through the LDS FFT.  A region that is
    its template rolled by (dy, dx) peaks there with correlation 1; sampled surface
    elements equal the direct circular dot product of the reference's formula.'''
    spec = mods_spectral
    rng = np.random.default_rng(55)
    n, batch = 1024, 6
    templates = rng.random((batch, n, n), dtype=np.float32)
    shifts = [(0, 0), (1, 1023), (511, 512), (1000, 3), (17, 700), (640, 640)]
    regions = np.stack([np.roll(templates[b], s, axis=(0, 1)) for b, s in enumerate(shifts)])
    regions[5] = rng.random((n, n), dtype=np.float32)            # one unrelated pair
    out = spec.cross_correlate_2d_spectral_batched(templates, regions)
    for b, s in enumerate(shifts[:5]):
        peak = np.unravel_index(np.argmax(out[b]), out[b].shape)
        assert tuple(int(v) for v in peak) == s
        assert abs(float(out[b][peak]) - 1.0) < 1e-5
    # direct evaluation at sampled positions (float64 on the host)
    for b in (2, 5):
        t = templates[b].astype(np.float64)
        r = regions[b].astype(np.float64)
        tn = (t - t.mean()) / (t.std() * t.size)
        rn = (r - r.mean()) / r.std()
        for _ in range(12):
            k, l = (int(v) for v in rng.integers(0, n, size=2))
            want = float(np.sum(tn * np.roll(rn, (-k, -l), axis=(0, 1))))
            assert abs(float(out[b, k, l]) - want) <= 1e-5 * max(1.0, abs(want)), (b, k, l)


@pytest.fixture(scope='module')
def mods_spectral():
    import correlation_spectral
    import _native
    _native.torch_cuda()
    return correlation_spectral


@pytest.mark.parametrize('batch,tshape,rshape', [
    (1, (256, 256), (256, 256)), (3, (512, 512), (512, 512)), (1, (4096, 4096), (4096, 4096)),
    (2, (256, 512), (256, 512)), (1, (2048, 256), (2048, 256)), (1, (4096, 1024), (4096, 1024)),
    (2, (256, 4096), (256, 4096)), (2, (300, 200), (512, 256)), (1, (700, 3000), (512, 2048)),
    (1, (4096, 4000), (4096, 4096))])
def test_fft_every_wide_size_and_mixed_shapes(mods_spectral, oracle, batch, tshape, rshape):
    """Every transform length of the N / 16-threads-per-transform kernels (256 ... 4096) as
    rows and as columns, square and not, batches, zero-padded and cropped templates:
    first and last pair of the batch against the float64 FFT evaluation of
    correlation_spectral.py:37-54, float32 surfaces to 1e-5."""
    rng = np.random.default_rng(tshape[0] * 7 + rshape[1])
    t = rng.random((batch,) + tshape, dtype=np.float32)
    r = rng.random((batch,) + rshape, dtype=np.float32) * 255
    out = mods_spectral.cross_correlate_2d_spectral_batched(t, r)
    assert out.shape == (batch,) + rshape and out.dtype == np.float32
    for b in sorted({0, batch - 1}):
        want = oracle.cross_correlate_2d_spectral_fft(t[b].astype(np.float64), r[b].astype(np.float64))
        assert_surface_close(out[b], want, 1e-5, f'wide fft {tshape} over {rshape}, pair {b}')


@pytest.mark.parametrize('n', [256, 512, 2048])
def test_fft_wide_branches_and_planted_shift(mods_spectral, n):
    """The reference's branches at the wide sizes (correlation_spectral.py:39-49): flat
    template with a different / the same mean, constant region, and a rolled copy."""
    rng = np.random.default_rng(n)
    t = rng.random((5, n, n), dtype=np.float32)
    r = rng.random((5, n, n), dtype=np.float32)
    t[1] = 0.5
    t[2] = 0.25
    r[2] = 0.25
    r[3] = 0.75
    shift = (n - 3, n // 2 + 1)
    r[4] = np.roll(t[4], shift, axis=(0, 1))
    out = mods_spectral.cross_correlate_2d_spectral_batched(t, r)
    assert np.all(out[1] == 0.0)
    assert np.all(out[2] == 1.0)
    assert np.isnan(out[3]).all()
    assert tuple(int(v) for v in np.unravel_index(np.argmax(out[4]), out[4].shape)) == shift
    assert abs(float(out[4].max()) - 1.0) < 1e-5
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() < 0.1


def test_million_pairs_sampled_against_the_oracle(oracle):
    '''2^20 circular 31 x 31 pairs in one call (the reference-faithful dense-scan
    arithmetic): a pair of identical boxes peaks at 1 at index 0, a random sample
    of the rest matches the oracle in both accumulation precisions.'''
    import torch
    import _native as nv
    rng = np.random.default_rng(4)
    H = W = 1024
    left = rng.random((H, W))
    right = np.roll(left, 7, axis=1) + 0.01 * rng.random((H, W))
    n = 1 << 20
    ty = rng.integers(0, H - 31, size=n)
    tx = rng.integers(128, W - 31 - 128, size=n)
    dx = rng.integers(-128, 129, size=n)
    desc = np.empty((n, 8), dtype=np.int64)
    desc[:, 0] = ty * W + tx
    desc[:, 4] = ty * W + tx + dx
    desc[:, 1] = desc[:, 5] = W
    desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = 31
    dl, dr, dd = nv.to_device(left), nv.to_device(right), nv.to_device(desc)
    sample = rng.integers(0, n, size=150)
    want = np.array([oracle.cross_correlate_2d_spectral(
        left[ty[p]:ty[p] + 31, tx[p]:tx[p] + 31],
        right[ty[p]:ty[p] + 31, tx[p] + dx[p]:tx[p] + dx[p] + 31], want='max') for p in sample])
    for precision, tol in ((nv.ACC_F32, 1e-5), (nv.ACC_F64, 1e-11)):
        out = nv.empty(n, torch.float64)
        arg = nv.empty(n, torch.int64)
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, 31, 1e-4, 1e-9, precision,
                nv.ptr(out), nv.ptr(arg), None, None, nv.stream_ptr())
        got = out.cpu().numpy()
        assert_surface_close(got[sample], want, tol, f'sampled pairs, precision {precision}')
        assert np.all(np.isfinite(got)) and got.max() <= 1.0 + 1e-6
        # the planted shift: regions displaced by exactly 7 columns hold the template itself
        same = np.nonzero(dx == 7)[0]
        assert len(same) > 1000
        assert np.all(got[same] > 0.99)
        assert np.all(arg.cpu().numpy()[same] == 0)
    # identical boxes
    desc_same = desc[:4096].copy()
    desc_same[:, 4] = desc_same[:, 0]
    d2 = nv.to_device(desc_same)
    out = nv.empty(4096, torch.float64)
    arg = nv.empty(4096, torch.int64)
    nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dl), nv.ptr(d2), 4096, 31, 1e-4, 1e-9, nv.ACC_F32, nv.ptr(out),
            nv.ptr(arg), None, None, nv.stream_ptr())
    assert np.all(np.abs(out.cpu().numpy() - 1.0) < 1e-5) and np.all(arg.cpu().numpy() == 0)


def test_long_1d_correlation_sampled(oracle):
    '''Config 2's length (n = 44 000, 2n - 1 lags): sampled lags against the
    reference's formula evaluated directly (correlation.py:63-70: the second vector
    padded with n - 1 zeros either side, both means removed, divided by n), the
    maximum where the planted delay puts it.'''
    import contextlib
    import io
    import correlation
    rng = np.random.default_rng(2)
    n, delay = 44000, 137
    a = rng.standard_normal(n)
    b = np.roll(a, delay) + 0.05 * rng.standard_normal(n)
    with contextlib.redirect_stdout(io.StringIO()):
        got = correlation.cross_correlate_1d(a, b)
        short = correlation.cross_correlate_1d(a[:500], b[:500], normalised=True)
    assert_surface_close(short, oracle.cross_correlate_1d(a[:500], b[:500], normalised=True), 1e-5, 'short 1D')
    assert got.shape == (2 * n - 1,)
    assert int(np.argmax(got)) == n - 1 + delay
    scale = float(np.max(np.abs(got)))
    padded = np.concatenate([np.zeros(n - 1), b, np.zeros(n - 1)])
    a0 = a - a.mean()
    for i in rng.integers(0, 2 * n - 1, size=40):
        want = float(np.sum(a0 * (padded[i:i + n] - b.mean())) / n)
        assert abs(float(got[i]) - want) <= 1e-5 * scale, int(i)


def test_dense_scan_through_sequence_scan(oracle):
    '''The dense configuration through the reference's own entry point: 31 x 31 windows at
    every pixel of a 256 x 256 pair, 257 candidate offsets each (16.7 million circular
    pairs, derived on the device a band of window rows at a time).  The oracle needs
    about a millisecond per pair, so it checks a sample of windows - corners, borders,
    interior - with its own enumeration and selection rule: position exactly,
    correlation to 1e-11.  Over all windows: the planted shift is found wherever the
    window and its match lie inside the image, and the result does not depend on how
    the rows were cut into chunks.'''
    import contextlib
    import io
    import sv_image_comparison as v
    from oracle import ncc_oracle
    rng = np.random.default_rng(21)
    n, win, shift = 256, 31, 9
    left = rng.integers(0, 256, size=(n, n), dtype=np.uint8) / 255.0
    right = np.roll(left, shift, axis=1)
    cfg = [{'window_width': win, 'window_height': win, 'scheme': [257, 1], 'scheme_shift_size': [1, 0],
            'window_overlap': 0.96, 'correlation_threshold': 0.5}]
    saved = v.LOG_WINDOW_SUMMARY, v.GRID_CHUNK_PAIRS
    v.LOG_WINDOW_SUMMARY = False
    try:
        results = []
        for chunk in (1 << 20, 3_000_017):
            v.GRID_CHUNK_PAIRS = chunk
            with contextlib.redirect_stdout(io.StringIO()):
                windows = v.sequence_scan(left, right, cfg)[0]
            results.append((windows._dp_x.copy(), windows._dp_y.copy()))
    finally:
        v.LOG_WINDOW_SUMMARY, v.GRID_CHUNK_PAIRS = saved
    assert len(windows) == n * n
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    dp_x = results[0][0].reshape(n, n)
    assert not results[0][1].any()
    # windows that are whole and whose match is whole: the shift is exact there
    assert (dp_x[:n - win + 1, :n - win + 1 - shift] == shift).all()
    sample = [0, n - 1, n * (n - 1), n * n - 1, 5 * n + 128, 128 * n + 3, 128 * n + n - 2, (n - 2) * n + 77]
    sample += [int(i) for i in rng.integers(0, n * n, size=24)]
    for window_id in sample:
        row, column = divmod(window_id, n)
        centre = (column + win / 2, row + win / 2)
        regions = []
        for c in ncc_oracle.candidate_centres(centre[0], win, centre[1], win, (257, 1), (1, 0)):
            box = ncc_oracle.window_box(c[0], c[1], win, win, n, n)
            if box is not None:
                regions.append((c, right[box[2]:box[3], box[0]:box[1]]))
        pos, best = ncc_oracle._fold(left[row:row + win, column:column + win], centre, regions, 0.5,
                                     ncc_oracle.CENTRE_TOL)
        info = windows[window_id]
        assert (info['row'], info['column'], info['centre']) == (row, column, centre)
        assert len(info['target_regions']) == len(regions)
        assert (info['dp_x'], info['dp_y']) == (pos[0] - centre[0], pos[1] - centre[1]), window_id


def test_dense_float_scan_after_downsample_matches_the_oracle(oracle):
    '''The dense scan after the reference's own preprocessing: the 4096 x 4096 pair
    through utils.downsample(image, 2) (float64 pixels that are no 8-bit levels any
    more), then the band scan of the 2048 x 2048 results - bands of window rows at the
    top, in the middle and at the bottom against the oracle's float64 scan: disparities
    exactly, correlations to the float32 bar.'''
    import contextlib
    import io
    import band_scan as bs
    import bench
    import utils
    left, right = bench.synthetic_pair(4096, 4096, seed=1000)
    with contextlib.redirect_stdout(io.StringIO()):
        left, right = np.asarray(utils.downsample(left, 2)), np.asarray(utils.downsample(right, 2))
    with pytest.raises(TypeError):
        bs.as_uint8_image(left)
    scanner = bs.BandScanner(2048, 2048, WIN, WIN, 64)
    scanner.upload(left, right)
    assert scanner.kind == 'f64'
    scanner.run(THRESHOLD)
    disp, val = scanner.disp.cpu().numpy(), scanner.val.cpu().numpy()
    rows = 4
    for row0 in (0, 1009, 2018 - rows):
        crop = slice(row0, row0 + rows + WIN - 1)
        odisp, oval = oracle.band_scan(left[crop], right[crop], WIN, WIN, 64, mode='fold', threshold=THRESHOLD)
        assert np.array_equal(disp[row0:row0 + rows], odisp), row0
        assert_surface_close(val[row0:row0 + rows], oval, 1e-5, 'float band values')
    assert (disp != 0).mean() > 0.5
