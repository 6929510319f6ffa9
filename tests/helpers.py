import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'stereo-vision_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The product modules keep the reference's flat module names (correlation,
# correlation_spectral, sv_image_comparison, utils), so the package directory
# itself goes on sys.path; the oracle is imported as the package `oracle`.
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rand_pair(n):
    '''The seeded n x n float32 template/region pair (and sample indices) whose
    reference outputs g4_spectral_rand<n>.npz holds (same generator calls as
    tests/golden/make_golden.py).'''
    g = np.random.default_rng(1000 + n)
    t = g.random((n, n), dtype=np.float32)
    r = g.random((n, n), dtype=np.float32)
    idx = g.integers(0, n * n, size=4096)
    return t, r, idx


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def assert_surface_close(got, want, tol=1e-5, what=''):
    '''NCC surfaces live on [-1, 1]; the bar is 1e-5 relative with the scale
    floored at 1 (so 1e-5 absolute near zero crossings), NaNs in the same
    places.'''
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f'{what}: shape {got.shape} != {want.shape}'
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), f'{what}: NaN pattern differs ({gn.sum()} vs {wn.sum()})'
    err = np.abs(got[~wn] - want[~wn])
    bound = tol * np.maximum(1.0, np.abs(want[~wn]))
    if err.size:
        worst = np.max(err / bound)
        assert worst <= 1.0, f'{what}: max err {err.max():.3e} exceeds {tol} relative (x{worst:.2f})'


def oracle_select_local(tbase, rbase, desc, is_centre, skip, win_off, threshold, max_side):
    '''Stand-in for the device path with the same contract, built on the oracle.'''
    import ctypes
    from oracle import ncc_oracle as orc
    import sv_image_comparison as v
    lib = orc.lib()
    left, right = tbase, rbase          # numpy arrays in this test
    d = np.ascontiguousarray(desc, dtype=np.int64)
    corr = np.full(len(d), np.nan)
    for p, (to, tp, th, tw, ro, rp, rh, rw) in enumerate(d):
        if rh <= 0 or rw <= 0:
            continue
        ty, tx = divmod(int(to), int(tp))
        ry, rx = divmod(int(ro), int(rp))
        corr[p] = orc.cross_correlate_2d_spectral(left[ty:ty + th, tx:tx + tw],
                                                  right[ry:ry + rh, rx:rx + rw], want='max')
    nwin = len(win_off) - 1
    best_pair = np.empty(nwin, dtype=np.int64)
    best_val = np.empty(nwin)
    c_u8 = ctypes.POINTER(ctypes.c_uint8)
    c_i64 = ctypes.POINTER(ctypes.c_int64)
    c_d = ctypes.POINTER(ctypes.c_double)
    cen = np.ascontiguousarray(is_centre, dtype=np.uint8)
    skp = np.ascontiguousarray(skip, dtype=np.uint8)
    off = np.ascontiguousarray(win_off, dtype=np.int64)
    lib.orc_scan_fold(corr.ctypes.data_as(c_d), cen.ctypes.data_as(c_u8), skp.ctypes.data_as(c_u8),
                      off.ctypes.data_as(c_i64), ctypes.c_long(nwin), ctypes.c_double(threshold),
                      ctypes.c_double(v.CENTRE_TOL), best_pair.ctypes.data_as(c_i64),
                      best_val.ctypes.data_as(c_d))
    return best_pair, best_val


def enumerate_grid_pairs(tables, ncol, nx, nrow, ny, row_begin, row_end, left_pitch, right_pitch):
    '''What libsvamd's grid_enumerate_kernel derives from the per-axis tables, thread
    by thread, in numpy: (desc [P, 8], is_centre [P], pair_slot [P], win_off [nwin + 1])
    for window rows [row_begin, row_end).  Raises if two candidates claim one pair
    index or one stays unclaimed, i.e. if the tables do not add up.'''
    row_first, xcand, xwin = tables['row_first'], tables['xcand'], tables['xwin']
    ycand, ywin = tables['ycand'], tables['ywin']
    npairs = int(row_first[row_end] - row_first[row_begin])
    nwin = (row_end - row_begin) * ncol
    desc = np.full((npairs, 8), -1, dtype=np.int64)
    is_centre = np.zeros(npairs, dtype=np.uint8)
    pair_slot = np.full(npairs, -1, dtype=np.int32)
    win_off = np.zeros(nwin + 1, dtype=np.int64)
    for w in range(nwin):
        r, c = row_begin + w // ncol, w % ncol
        xw, yw = xwin[c].astype(np.int64), ywin[r].astype(np.int64)
        own_ok = int(xw[2] & yw[2])
        first = int(row_first[r] - row_first[row_begin] + yw[2] * xw[5] + yw[3] * xw[6] - yw[4] * xw[7])
        win_off[w] = first
        if w == nwin - 1:
            win_off[nwin] = first + own_ok + yw[3] * xw[3] - yw[4] * xw[4]
        for slot in range(1 + nx * ny):
            if slot == 0:
                xc, yc, valid, rank = xcand[c, 0], ycand[r, 0], own_ok != 0, 0
            else:
                iy, ix = divmod(slot - 1, nx)
                xc, yc = xcand[c, 1 + ix].astype(np.int64), ycand[r, 1 + iy].astype(np.int64)
                xrej, yrej, xac, yac = xc[2] & 1, yc[2] & 1, xc[2] & 2, yc[2] & 2
                valid = not (xrej or yrej) and not (xac and yac)
                rank = int(own_ok + yc[3] * xw[3] + xc[3] - (yc[4] * xw[4] + (xc[4] if yac else 0)))
            if not valid:
                continue
            p = first + rank
            assert 0 <= p < npairs and pair_slot[p] == -1, 'two candidates share a pair index'
            desc[p] = (yw[0] * left_pitch + xw[0], left_pitch, yw[1] - yw[0], xw[1] - xw[0],
                       yc[0] * right_pitch + xc[0], right_pitch, yc[1] - yc[0], xc[1] - xc[0])
            is_centre[p] = slot == 0
            pair_slot[p] = slot
    assert (pair_slot >= 0).all(), 'a pair index stays unclaimed'
    return desc, is_centre, pair_slot, win_off


def oracle_scan_grid_rows(left, right, grid, tables, row_begin, row_end, threshold):
    '''Stand-in for sv_image_comparison._scan_grid_rows (the svamd_scan_grid call) with the
    same contract: pairs from the tables as the device derives them, correlations and
    the selection rule from the oracle.'''
    desc, is_centre, pair_slot, win_off = enumerate_grid_pairs(
        tables, grid.ncol, grid.ax.count, grid.nrow, grid.ay.count, row_begin, row_end, left.shape[1],
        right.shape[1])
    skip = (desc[:, 6] <= 0) | (desc[:, 7] <= 0)
    best_pair, best_val = oracle_select_local(left, right, desc, is_centre, skip, win_off, threshold, grid.max_side)
    slot = np.where(best_pair >= 0, pair_slot[np.maximum(best_pair, 0)], -1).astype(np.int32)
    return slot, np.where(best_pair >= 0, best_val, 0.0)


def check_band_fold_against_g14(band_scan_fold):
    '''g14_dense_spatial_fold: the reference's sequence_scan (its own selection rule)
    over its spatial correlator at full overlap, on a dense 31 x 31, 25-offset
    config.  `band_scan_fold(left_u8, right_u8, win, max_disp, threshold)` must
    return (disparity, correlation) per window origin; every window whose template
    is full-size is compared: disparity exactly, correlation to 1e-12.'''
    import json
    g = golden('g14_dense_spatial_fold')
    cfg = json.loads(str(g['config']))[0][0]
    win, md, thr = cfg['window_width'], (cfg['scheme'][0] - 1) // 2, cfg['correlation_threshold']
    left, right = g['left'], g['right']
    h, w = left.shape
    disp, val = band_scan_fold(left, right, win, md, thr)
    assert disp.shape == (h - win + 1, w - win + 1)
    rows = g['seq0_stage0_windows']      # id, row, column, cx, cy, dp_x, dp_y, ...
    calls = g['seq0_calls']              # id, pos_x, pos_y, corr_max, n_regions
    checked = moved = 0
    for wrow, call in zip(rows, calls):
        y, x = int(wrow[1]), int(wrow[2])
        if y > h - win or x > w - win:
            continue                     # template clipped by the image border: not a band-scan window
        assert wrow[6] == 0
        assert disp[y, x] == wrow[5], f'window ({y},{x}): disparity {disp[y, x]} != reference {wrow[5]}'
        assert abs(val[y, x] - call[3]) <= 1e-12, f'window ({y},{x}): {val[y, x]} vs {call[3]}'
        checked += 1
        moved += wrow[5] != 0
    assert checked == (h - win + 1) * (w - win + 1)
    assert moved > checked // 4          # the fixture exercises non-centre selections
