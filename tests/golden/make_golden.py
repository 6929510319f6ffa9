#!/usr/bin/env python3
'''Generates the golden input/output vectors under tests/golden/ by importing
the reference implementation from /root/reference and running it on small
inputs.  Only data (inputs and the reference's outputs) is written; no
reference source travels.

Run once in the build container (the reference is not present on the GPU box):

    MPLBACKEND=Agg python3 -B tests/golden/make_golden.py [--only g1,g4,...]

The reference's `utils.py` imports scikit-image, which is not installed here;
only `downsample` needs it and the hot path never calls it, so an empty
placeholder module is registered for that import before loading `utils`.
'''
import argparse
import io
import contextlib
import json
import os
import sys
import time
import types

import numpy as np

REF = os.environ.get('SV_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sk = types.ModuleType('skimage')
    skt = types.ModuleType('skimage.transform')
    skt.resize = None
    sk.transform = skt
    sys.modules.setdefault('skimage', sk)
    sys.modules.setdefault('skimage.transform', skt)
    sys.path.insert(0, REF)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    import correlation
    import correlation_spectral
    import sv_image_comparison
    import utils
    return correlation, correlation_spectral, sv_image_comparison, utils


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def grey_u8(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)')


# --------------------------------------------------------------------------
def g1_spatial_template_region(ref):
    corr, _, _, utils = ref
    region = grey_u8(f'{REF}/images/region.PNG')
    template = utils.greyscale_with_nan(f'{REF}/images/template.PNG')
    t0 = time.time()
    out = quiet(corr.cross_correlate_2d, template, region, 5, 5)
    print(f'  g1 reference time {time.time() - t0:.1f}s, nan={np.isnan(out).sum()},'
          f' max_pos={utils.max_pos(out, 5, 5)}')
    mp = np.array(utils.max_pos(out, 5, 5), dtype=np.int64)
    save('g1_spatial_template_region', template=template, region=region,
         step=np.array([5, 5]), surface=out, max_pos=mp)


def g2_spatial_nan_template(ref):
    corr, _, _, utils = ref
    region = grey_u8(f'{REF}/images/region.PNG')
    template = utils.greyscale_with_nan(f'{REF}/images/rm_template.png')
    crop_a = np.ascontiguousarray(region[250:433, 700:862])
    out_a = quiet(corr.cross_correlate_2d, template, crop_a, 7, 7)
    crop_b = np.ascontiguousarray(region[300:433, 745:862])
    out_b = quiet(corr.cross_correlate_2d, template, crop_b, 1, 1)
    # asymmetric steps on the same crop
    out_c = quiet(corr.cross_correlate_2d, template, crop_a, 3, 11)
    save('g2_spatial_nan_template', template=template,
         region_a=crop_a, step_a=np.array([7, 7]), surface_a=out_a,
         region_b=crop_b, step_b=np.array([1, 1]), surface_b=out_b,
         region_c=crop_a, step_c=np.array([3, 11]), surface_c=out_c)


def g3_spatial_band(ref):
    corr, _, _, _ = ref
    rng = np.random.default_rng(3)
    n = 6
    templates = rng.integers(0, 256, size=(n, 31, 31), dtype=np.uint8)
    bands = rng.integers(0, 256, size=(n, 31, 287), dtype=np.uint8)
    # plant each template in its band at a known offset, with noise
    for i in range(n):
        off = int(rng.integers(0, 257))
        noisy = templates[i].astype(np.int32) + rng.integers(-20, 21, size=(31, 31))
        bands[i, :, off:off + 31] = np.clip(noisy, 0, 255).astype(np.uint8)
    surfaces = np.stack([
        quiet(corr.cross_correlate_2d, templates[i].astype(np.float64), bands[i])
        for i in range(n)])
    save('g3_spatial_band', templates=templates, bands=bands, surfaces=surfaces)


def g4_spectral_2d(ref):
    _, spec, _, utils = ref
    rng = np.random.default_rng(4)
    fn = spec.cross_correlate_2d_spectral
    arrays = {}

    def case(tag, t, r):
        with np.errstate(all='ignore'):
            arrays[f'{tag}_t'] = t
            arrays[f'{tag}_r'] = r
            arrays[f'{tag}_out'] = fn(t, r)

    case('sq12', rng.random((12, 12)), rng.random((12, 12)))
    case('sq25', rng.random((25, 25)), rng.random((25, 25)))
    case('sq31_u8', rng.integers(0, 256, (31, 31), dtype=np.uint8),
         rng.integers(0, 256, (31, 31), dtype=np.uint8))
    case('crop50v25', rng.random((50, 50)), rng.random((25, 25)))
    case('pad12x8v12', rng.random((12, 8)), rng.random((12, 12)))
    case('crop12v12x7', rng.random((12, 12)), rng.random((12, 7)))
    case('mixed9x14v11x6', rng.random((9, 14)), rng.random((11, 6)))
    const = np.full((12, 12), 0.25)
    case('const_t_same_mean', const, const.copy())
    case('const_t_diff_mean', const, rng.random((12, 12)))
    case('const_r', rng.random((12, 12)), np.full((12, 12), 0.5))
    case('lowstd_t', 0.3 + 1e-5 * rng.random((10, 10)), rng.random((10, 10)))
    # matched content: region is a circular shift of the template
    base = rng.random((20, 20))
    case('rolled20', base, np.roll(base, (3, 7), axis=(0, 1)))
    save('g4_spectral_small', **arrays)

    # the template/region images, full surface decimated + arg-max
    region = grey_u8(f'{REF}/images/region.PNG')
    template = grey_u8(f'{REF}/images/template.PNG')
    out = fn(template, region)
    idx = rng.integers(0, out.size, size=8192)
    save('g4_spectral_images', template=template, region=region,
         max_pos=np.array(utils.max_pos(out), dtype=np.int64),
         max_val=np.array(out.max()),
         sample_idx=idx, sample_val=out.ravel()[idx],
         rows=out[::16].copy())

    for n in (256, 1024):
        # inputs are regenerated by the tests from the same seeded generator
        # float32-valued data handed to the reference as float64, so that the
        # golden surface carries float64 accuracy
        t, r, idx = rand_pair(n)
        out = fn(t.astype(np.float64), r.astype(np.float64))
        save(f'g4_spectral_rand{n}',
             t_checksum=np.array(t.astype(np.float64).sum()),
             r_checksum=np.array(r.astype(np.float64).sum()),
             argmax=np.array(np.argmax(out)), max_val=np.array(out.max()),
             sample_idx=idx, sample_val=out.ravel()[idx])


def rand_pair(n):
    '''The seeded n x n float32 template/region pair (and sample indices) whose
    reference outputs g4_spectral_rand<n>.npz holds.'''
    g = np.random.default_rng(1000 + n)
    t = g.random((n, n), dtype=np.float32)
    r = g.random((n, n), dtype=np.float32)
    idx = g.integers(0, n * n, size=4096)
    return t, r, idx


def g5_1d(ref):
    corr, spec, _, _ = ref
    rng = np.random.default_rng(5)
    n = 200
    base = rng.standard_normal(n + 17)
    v1 = base[17:].copy()
    v2 = base[:n].copy() + 0.05 * rng.standard_normal(n)
    arrays = dict(v1=v1, v2=v2)
    arrays['xc'] = quiet(corr.cross_correlate_1d, v1, v2)
    arrays['xc_norm'] = quiet(corr.cross_correlate_1d, v1, v2, True)
    arrays['xc_raw_norm'] = np.array(quiet(corr.cross_correlate_1d_raw, list(v1), list(v2), True))
    arrays['xc_spec'] = spec.cross_correlate_1d_spectral(v1, v2)
    # unequal lengths for the spectral variant
    t = rng.standard_normal(37)
    s = rng.standard_normal(301)
    s[100:137] += 2 * t
    arrays['t_short'] = t
    arrays['s_long'] = s
    arrays['xc_spec_uneq'] = spec.cross_correlate_1d_spectral(t, s)
    # analytic cos / -sin signals (the generator of the reference's 1D demo)
    step = np.pi / 2000
    x = np.arange(0, 4 * np.pi, step)
    c, s_ = np.cos(x), -np.sin(x)
    arrays['cs_n'] = np.array(len(x))
    full = quiet(corr.cross_correlate_1d, c, s_, True)
    arrays['cs_xc_norm_dec'] = full[::16].copy()
    arrays['cs_xc_norm_argmax'] = np.array(np.argmax(full))
    fs = spec.cross_correlate_1d_spectral(c, s_)
    arrays['cs_spec_dec'] = fs[::16].copy()
    arrays['cs_spec_argmax'] = np.array(np.argmax(fs))
    save('g5_1d', **arrays)


# --------------------------------------------------------------------------
def run_scan(ref, left, right, config):
    '''Runs the reference's sequence_scan per sequence, recording every
    image_scan call's (window id, returned position, returned correlation).'''
    _, _, svic, _ = ref
    svic.LOG_WINDOW_SUMMARY = False
    calls = []
    orig = svic.image_scan

    def recording(window_info, window_id, corr_threshold=0):
        pos, val = orig(window_info, window_id, corr_threshold)
        calls.append((window_id, float(pos[0]), float(pos[1]), float(val),
                      len(window_info['target_regions'])))
        return pos, val
    svic.image_scan = recording
    out = {}
    try:
        for s, seq_cfg in enumerate(config):
            calls.clear()
            t0 = time.time()
            res = quiet(svic.sequence_scan, left, right, seq_cfg)
            dt = time.time() - t0
            out[f'seq{s}_calls'] = np.array(calls, dtype=np.float64).reshape(-1, 5)
            for st, windows in enumerate(res):
                rows = []
                centres = []
                sizes = []
                for wid, w in windows.items():
                    sc = w['stage_centres']
                    ss = w['stage_sizes']
                    rows.append((wid, w['row'], w['column'], w['centre'][0], w['centre'][1],
                                 w['dp_x'], w['dp_y'], len(sc), len(ss), len(w['target_regions'])))
                    cpad = [(np.nan, np.nan)] * 4
                    spad = [(-1, -1)] * 4
                    for i, c in enumerate(sc):
                        cpad[i] = c
                    for i, z in enumerate(ss):
                        spad[i] = z
                    centres.append(cpad)
                    sizes.append(spad)
                out[f'seq{s}_stage{st}_windows'] = np.array(rows, dtype=np.float64).reshape(-1, 10)
                out[f'seq{s}_stage{st}_centres'] = np.array(centres, dtype=np.float64).reshape(-1, 4, 2)
                out[f'seq{s}_stage{st}_sizes'] = np.array(sizes, dtype=np.int64).reshape(-1, 4, 2)
            nz = sum(1 for w in res[0].values() if w['dp_x'] != 0 or w['dp_y'] != 0)
            print(f'    seq {s}: {dt:.1f}s, {len(res[0])} stage-0 windows, '
                  f'{len(calls)} image_scan calls, {nz} non-zero dp')
        out['n_seq'] = np.array(len(config))
    finally:
        svic.image_scan = orig
    return out


def g6_scans(ref):
    left = grey_u8(f'{REF}/images-p2-uncal/left_cone.tiff')
    right = grey_u8(f'{REF}/images-p2-uncal/right_cone.tiff')
    save('g6_cone_images', left=left, right=right)
    lf, rf = left / 255.0, right / 255.0
    for cfg_name in ('scan_config_portal', 'scan_config', 'scan_config_example', 'scan_config_test_2'):
        with open(f'{REF}/configs/{cfg_name}.json') as f:
            cfg = json.load(f)
        print(f'  cone x {cfg_name}')
        out = run_scan(ref, lf, rf, cfg)
        save(f'g6_cone_{cfg_name}', config=np.array(json.dumps(cfg)), **out)
    with open(f'{REF}/configs/scan_config_test_1.json') as f:
        cfg = json.load(f)
    for name, (y0, x0) in (('box', (400, 350)), ('tuscany', (300, 500))):
        l = grey_u8(f'{REF}/images-p2-uncal/left_{name}.tiff')[y0:y0 + 256, x0:x0 + 256]
        r = grey_u8(f'{REF}/images-p2-uncal/right_{name}.tiff')[y0:y0 + 256, x0:x0 + 256]
        l, r = np.ascontiguousarray(l), np.ascontiguousarray(r)
        print(f'  {name} crop x scan_config_test_1')
        out = run_scan(ref, l / 255.0, r / 255.0, cfg)
        save(f'g6_{name}256_scan_config_test_1', left=l, right=r,
             config=np.array(json.dumps(cfg)), **out)


def g9_multipass(ref):
    '''Two-stage (multi-pass) configs beyond the shipped scan_config.json:
    factor 3, overlapping windows, a 2-D scheme.'''
    d = np.load(os.path.join(OUT, 'g6_cone_images.npz'))
    lf, rf = d['left'] / 255.0, d['right'] / 255.0
    cfg = [[{'window_width': 48, 'window_height': 48, 'scheme': [5, 3],
             'scheme_shift_size': [0, 0], 'window_overlap': 0.25,
             'correlation_threshold': 0.3},
            {'factor': 3, 'correlation_threshold': 0}],
           [{'window_width': 40, 'window_height': 30, 'scheme': [7, 1],
             'scheme_shift_size': [6, 0], 'window_overlap': 0,
             'correlation_threshold': 0.5},
            {'factor': 2, 'correlation_threshold': 0}]]
    out = run_scan(ref, lf, rf, cfg)
    save('g9_cone_multipass', config=np.array(json.dumps(cfg)), **out)


def g7_dense(ref):
    rng = np.random.default_rng(0)
    left = rng.random((128, 160))
    right = np.roll(left, 5, axis=1)
    cfg = [[{'window_width': 31, 'window_height': 31, 'scheme': [33, 1],
             'scheme_shift_size': [1, 0], 'window_overlap': 0.96,
             'correlation_threshold': 0.5}]]
    out = run_scan(ref, left, right, cfg)
    save('g7_dense_roll5', left=left, right=right, config=np.array(json.dumps(cfg)), **out)


def g8_dense_u8(ref):
    '''Dense single-row-scheme scan on a small uint8-valued pair with a
    piece-wise constant disparity, for the band kernel's reference-faithful mode.'''
    rng = np.random.default_rng(8)
    h, w = 72, 120
    base = rng.integers(0, 256, size=(h, w + 32)).astype(np.float64)
    # mild smoothing so neighbouring offsets are correlated
    k = np.array([1, 2, 1]) / 4
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), 1, base)
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), 0, base)
    base = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    left = base[:, 16:16 + w].copy()
    right = np.zeros_like(left)
    right[:h // 2] = base[:h // 2, 16 - 6:16 - 6 + w]      # disparity +6
    right[h // 2:] = base[h // 2:, 16 + 9:16 + 9 + w]      # disparity -9
    noise = rng.integers(-3, 4, size=right.shape)
    right = np.clip(right.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    cfg = [[{'window_width': 31, 'window_height': 31, 'scheme': [25, 1],
             'scheme_shift_size': [1, 0], 'window_overlap': 0.96,
             'correlation_threshold': 0.6}]]
    out = run_scan(ref, left / 255.0, right / 255.0, cfg)
    save('g8_dense_u8', left=left, right=right, config=np.array(json.dumps(cfg)), **out)


def g10_depth_grids(ref):
    '''The displacement grids the reference's stereo driver (test_image_comparison.py
    main()) saves with --depth_output, for the cone pair and three configs.

    The driver is run as it is, in a scratch directory that holds the images and
    the config under the relative paths it expects, with two adjustments at its
    edges: `downsample` (which needs scikit-image) is replaced in the driver's
    namespace by the uint8 / 255 conversion that defines the scan's input in
    these fixtures, and numpy's removed alias `np.Infinity` is restored.'''
    import importlib
    import shutil
    import tempfile
    from PIL import Image
    if not hasattr(np, 'Infinity'):
        np.Infinity = np.inf
    drv = importlib.import_module('test_image_comparison')
    drv.downsample = lambda image, factor: image / 255.0
    d = np.load(os.path.join(OUT, 'g6_cone_images.npz'))
    g9 = json.loads(str(np.load(os.path.join(OUT, 'g9_cone_multipass.npz'))['config']))
    with open(f'{REF}/configs/scan_config.json') as f:
        two_stage = json.load(f)
    with open(f'{REF}/configs/scan_config_portal.json') as f:
        portal = json.load(f)
    arrays = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    try:
        os.makedirs(f'{tmp}/images-p2-uncal')
        os.makedirs(f'{tmp}/configs')
        os.makedirs(f'{tmp}/depth-data')
        Image.fromarray(d['left'], 'L').save(f'{tmp}/images-p2-uncal/left_cone.tiff')
        Image.fromarray(d['right'], 'L').save(f'{tmp}/images-p2-uncal/right_cone.tiff')
        os.chdir(tmp)
        for tag, cfg in (('scan_config', two_stage), ('portal', portal), ('multipass', g9)):
            with open('configs/cfg.json', 'w') as f:
                json.dump(cfg, f)
            argv = sys.argv
            sys.argv = ['test_image_comparison.py', '--images', 'cone', '--config', 'cfg.json',
                        '--depth_output', 'out.json', '--hide_depth']
            try:
                quiet(drv.main)
            finally:
                sys.argv = argv
            with open('depth-data/out.json') as f:
                saved = json.load(f)
            arrays[f'{tag}_config'] = np.array(json.dumps(cfg))
            for key in ('depth_grid', 'dp_x_grid', 'dp_y_grid', 'contributions_grid'):
                arrays[f'{tag}_{key}'] = np.array(saved[key], dtype=np.float64)
            print(f'    {tag}: depth max {np.max(arrays[tag + "_depth_grid"]):.4f}, '
                  f'painted {np.isfinite(arrays[tag + "_contributions_grid"]).sum()} px')
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    save('g10_cone_depth_grids', **arrays)


def dot_grid_image(h, w, seed):
    '''A synthetic calibration image: bright Gaussian dots on a jittered grid over
    a noisy dark background, uint8.'''
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 20 + 6 * rng.standard_normal((h, w))
    for gy in range(70, h - 60, 60):
        for gx in range(80, w - 60, 65):
            cy, cx = gy + rng.uniform(-6, 6), gx + rng.uniform(-6, 6)
            img += 200 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 5.0 ** 2))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def g11_find_peaks(ref):
    '''The reference's calibration peak finder (sv_calibration.find_peaks) on two
    synthetic dot-grid images (its own calibration images are not in the tree).'''
    import importlib
    cal = importlib.import_module('sv_calibration')
    arrays = {}
    for tag, (h, w, seed) in (('a', (330, 470, 11)), ('b', (256, 512, 12))):
        img = dot_grid_image(h, w, seed)
        with np.errstate(all='ignore'):
            x, y = cal.find_peaks(img, neighborhood_size=10, threshold=0.2)
        print(f'    image {tag}: {len(x)} peaks')
        arrays[f'{tag}_image'] = img
        arrays[f'{tag}_x'] = np.asarray(x, dtype=np.float64)
        arrays[f'{tag}_y'] = np.asarray(y, dtype=np.float64)
    save('g11_find_peaks', **arrays)


def g13_spatial_nan_region(ref):
    '''A float region with NaN pixels (and a template with one): the reference yields
    NaN for every placement whose snapshot touches a NaN of the region.'''
    corr = ref[0]
    rng = np.random.default_rng(13)
    template = rng.random((9, 7)) * 200
    template[2, 3] = np.nan
    region = rng.random((30, 41)) * 255
    region[5:8, 10:12] = np.nan
    region[29, 40] = np.nan
    region[17, 0] = np.nan
    arrays = dict(template=template, region=region)
    for tag, (sx, sy) in (('a', (1, 1)), ('b', (3, 2))):
        with np.errstate(all='ignore'):
            arrays[f'surface_{tag}'] = quiet(corr.cross_correlate_2d, template, region, sx, sy)
        arrays[f'step_{tag}'] = np.array([sx, sy])
        print(f'  g13 {tag}: NaNs {np.isnan(arrays[f"surface_{tag}"]).sum()} of {arrays[f"surface_{tag}"].size}')
    save('g13_spatial_nan_region', **arrays)


def g14_dense_spatial_fold(ref):
    '''The reference's own selection rule (sv_image_comparison.py:236-255) over the
    SPATIAL correlator: sequence_scan on a small dense uint8 / 255 pair with its
    x_corr_2d hook set to the reference's cross_correlate_2d at full overlap (one
    output element: steps larger than the region), which is what the dense band
    kernel evaluates.  cross_correlate_2d raises when a region clipped by the image
    border is smaller than the template; the hook turns that into "no candidate"
    (an all-NaN surface, which image_scan's comparisons skip), the only adjustment.'''
    corr, _, svic, _ = ref
    rng = np.random.default_rng(14)
    h, w = 47, 112
    base = rng.integers(0, 256, size=(h, w + 40)).astype(np.float64)
    k = np.array([1, 2, 3, 2, 1]) / 9        # strong smoothing: neighbouring offsets score close to the best
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), 1, base)
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), 0, base)
    base = np.clip(np.rint((base - 127.5) * 3 + 127.5), 0, 255).astype(np.uint8)
    left = base[:, 20:20 + w].copy()
    right = np.zeros_like(left)
    right[:, :w // 3] = base[:, 20 - 1:20 - 1 + w // 3]                 # disparity +1: centre within CENTRE_TOL
    right[:, w // 3:2 * w // 3] = base[:, 20 + w // 3 + 9:20 + 2 * w // 3 + 9]    # disparity -9
    right[:, 2 * w // 3:] = base[:, 20 + 2 * w // 3:20 + w]             # disparity 0
    noise = rng.integers(-6, 7, size=right.shape)
    right = np.clip(right.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    cfg = [[{'window_width': 31, 'window_height': 31, 'scheme': [25, 1],
             'scheme_shift_size': [1, 0], 'window_overlap': 0.96,
             'correlation_threshold': 0.6}]]
    big = 10 ** 6

    def spatial_full_overlap(t, r):
        if t.shape[0] > r.shape[0] or t.shape[1] > r.shape[1]:
            return np.full((1, 1), np.nan)
        with np.errstate(all='ignore'):
            return corr.cross_correlate_2d(t, r, big, big)
    orig = svic.x_corr_2d
    svic.x_corr_2d = spatial_full_overlap
    try:
        out = run_scan(ref, left / 255.0, right / 255.0, cfg)
    finally:
        svic.x_corr_2d = orig
    save('g14_dense_spatial_fold', left=left, right=right, config=np.array(json.dumps(cfg)), **out)


def g12_greyscale_with_nan(ref):
    '''The reference's loader on its two RGBA images: decoded pixels in, grey
    levels with NaN for transparent pixels out.'''
    from PIL import Image
    utils = ref[3]
    arrays = {}
    for key, name in (('template', 'template.PNG'), ('rm_template', 'rm_template.png')):
        path = f'{REF}/images/{name}'
        arrays[f'{key}_rgba'] = np.asarray(Image.open(path))
        arrays[f'{key}_grey'] = utils.greyscale_with_nan(path)
        print(f'  g12 {name}: {arrays[f"{key}_rgba"].shape}, NaNs {np.isnan(arrays[f"{key}_grey"]).sum()}')
    save('g12_greyscale_with_nan', **arrays)


GROUPS = {
    'g1': g1_spatial_template_region,
    'g2': g2_spatial_nan_template,
    'g3': g3_spatial_band,
    'g4': g4_spectral_2d,
    'g5': g5_1d,
    'g6': g6_scans,
    'g7': g7_dense,
    'g8': g8_dense_u8,
    'g9': g9_multipass,
    'g10': g10_depth_grids,
    'g11': g11_find_peaks,
    'g12': g12_greyscale_with_nan,
    'g13': g13_spatial_nan_region,
    'g14': g14_dense_spatial_fold,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=','.join(GROUPS))
    args = ap.parse_args()
    ref = load_reference()
    for g in args.only.split(','):
        print(f'[{g}]')
        t0 = time.time()
        GROUPS[g](ref)
        print(f'  done in {time.time() - t0:.1f}s')


if __name__ == '__main__':
    main()
