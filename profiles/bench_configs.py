#!/usr/bin/env python3
'''Secondary measurements: the workloads next to the headline one (which is
bench.py's).  Prints one JSON line per measurement.  GPU times are wall clock of
the host-level call (upload + kernels + download) unless marked "device".

  config 1  spatial cross_correlate_2d, NaN-masked 123x111 template and a
            135x136 template over a 433x862 uint8 region, steps 5 and 1
  config 2  1D cross-correlation, n = 44000, spatial and spectral
  config 3  sequence_scan with the portal scan config (12x12 windows, 11x1
            scheme) on the 560x790 cone pair and on a synthetic 960x1080 pair
  config 5  batched 1024x1024 circular NCC through the LDS FFT, B = 8 and 64
  n1-n4     displacement grids, peak finder, image preparation, long 1D search
'''
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
sys.path.insert(0, ROOT)

import torch                     # noqa: E402
import _native as nv             # noqa: E402
import correlation               # noqa: E402
import correlation_spectral      # noqa: E402
import sv_image_comparison       # noqa: E402

def numpy_fft_circular_ncc(template, region):
    '''The reference's route on the CPU (correlation_spectral.py:32-55): global
    normalisation, then real(ifft2(conj(fft2(t, s)) * fft2(r))) with numpy's FFT.'''
    t = (template - template.mean()) / (template.std() * template.size)
    r = (region - region.mean()) / region.std()
    return np.real(np.fft.ifft2(np.conj(np.fft.fft2(t, s=region.shape)) * np.fft.fft2(r)))


PORTAL = [{'window_width': 12, 'window_height': 12, 'scheme': [11, 1], 'scheme_shift_size': [1, 0],
           'window_overlap': 0, 'correlation_threshold': 0.93}]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def timed(fn, *a, repeat=3, **k):
    best = None
    for _ in range(repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = quiet(fn, *a, **k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def emit(**kw):
    print(json.dumps(kw), flush=True)


def config1():
    rng = np.random.default_rng(1)
    region = rng.integers(0, 256, size=(433, 862), dtype=np.uint8)
    masked = rng.integers(0, 256, size=(123, 111)).astype(np.float64)
    masked[rng.random(masked.shape) < 0.7] = np.nan
    full = region[150:285, 290:426].astype(np.float64)
    for name, tmpl in (('123x111 NaN-masked', masked), ('135x136', full)):
        for step in (5, 1):
            dt, out = timed(correlation.cross_correlate_2d, tmpl, region, step, step)
            emit(config=1, what=f'cross_correlate_2d {name} template, 433x862 uint8 region, step {step}',
                 seconds=dt, evals=out.size, evals_per_s=out.size / dt,
                 gflops=2 * tmpl.size * out.size / dt / 1e9)


def config1_device():
    '''Device time of svamd_cross_correlate_2d alone (statistics, tables,
    extension, sliding kernel, epilogue), buffers resident.'''
    lib = nv.load_library()
    rng = np.random.default_rng(2)
    for (th, tw), (rh, rw), step in (((135, 136), (433, 862), 1), ((135, 136), (433, 862), 5),
                                     ((123, 111), (433, 862), 5), ((31, 31), (2048, 2048), 1),
                                     ((31, 31), (2048, 2048), 4), ((64, 64), (4096, 4096), 1),
                                     ((64, 64), (4096, 4096), 8), ((200, 200), (1000, 1000), 1)):
        region = nv.to_device(rng.integers(0, 256, size=(rh, rw), dtype=np.uint8))
        tmpl = nv.to_device(rng.random((th, tw)) * 255)
        orows, ocols = -(-rh // step), -(-rw // step)
        out = nv.empty((orows, ocols), torch.float64)
        ws = nv.workspace(lib.svamd_cross_correlate_2d_workspace_bytes(th, tw, rh, rw))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            nv.call('svamd_cross_correlate_2d', nv.ptr(tmpl), th, tw, nv.ptr(region), nv.U8, rh, rw, step, step,
                    nv.ptr(out), nv.ptr(ws), ws.numel(), nv.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        emit(config=1, what=f'svamd_cross_correlate_2d device time (all its kernels), {th}x{tw} template, '
                            f'{rh}x{rw} uint8 region, step {step}',
             device_ms=ms, evals_per_s=orows * ocols / ms * 1e3,
             tflops=2.0 * th * tw * orows * ocols / ms * 1e3 / 1e12)


def config2():
    rng = np.random.default_rng(0)
    n = 44000
    s1 = rng.standard_normal(n)
    s2 = np.roll(s1, 137) + 0.1 * rng.standard_normal(n)
    dt, out = timed(correlation.cross_correlate_1d, s1, s2)
    emit(config=2, what='cross_correlate_1d n=44000 (2n-1 lags)', seconds=dt, argmax=int(np.argmax(out)),
         lags_per_s=out.size / dt, gflops=2 * n * out.size / dt / 1e9)
    dt, out = timed(correlation_spectral.cross_correlate_1d_spectral, s1, s2)
    emit(config=2, what='cross_correlate_1d_spectral n=44000', seconds=dt, argmax=int(np.argmax(out)),
         lags_per_s=out.size / dt)
    t0 = time.perf_counter()
    lib = nv.load_library()
    d1, d2 = nv.to_device(s1), nv.to_device(s2)
    out = nv.empty(2 * n - 1, torch.float64)
    ws = nv.workspace(lib.svamd_cross_correlate_1d_workspace_bytes(n))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(4):
        if it == 1:
            e0.record()
        nv.call('svamd_cross_correlate_1d', nv.ptr(d1), nv.ptr(d2), n, 0, nv.ptr(out), nv.ptr(ws), ws.numel(),
                nv.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    emit(config=2, what='svamd_cross_correlate_1d device time, n=44000', device_ms=ms,
         tflops=2.0 * n * (2 * n - 1) / ms * 1e3 / 1e12)
    t0 = time.perf_counter()
    np.correlate(s2 - s2.mean(), s1 - s1.mean(), mode='full')
    emit(config=2, what='cpu numpy.correlate(full) of the same signals (1 thread)',
         seconds=time.perf_counter() - t0)


def scan_counts(res):
    pairs = sum(len(w['target_regions']) for w in res[0].values())
    moved = sum(1 for w in res[0].values() if w['dp_x'] or w['dp_y'])
    return len(res[0]), pairs, moved


def config3():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g6_cone_images.npz'))
    left, right = g['left'] / 255.0, g['right'] / 255.0
    dt, res = timed(sv_image_comparison.sequence_scan, left, right, PORTAL)
    nwin, pairs, moved = scan_counts(res)
    emit(config=3, what='sequence_scan portal config, cone 560x790', seconds=dt, windows=nwin, pairs=pairs,
         nonzero=moved, pairs_per_s=pairs / dt, surface_elems_per_s=pairs * 144 / dt)
    import bench
    l, r = bench.synthetic_pair(1080, 960, 5)
    dt, res = timed(sv_image_comparison.sequence_scan, l / 255.0, r / 255.0, PORTAL, repeat=2)
    nwin, pairs, moved = scan_counts(res)
    emit(config=3, what='sequence_scan portal config, synthetic 960x1080', seconds=dt, windows=nwin,
         pairs=pairs, nonzero=moved, pairs_per_s=pairs / dt, surface_elems_per_s=pairs * 144 / dt)
    # device-only time of the pair kernel for the cone scan
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import enumerate_grid_pairs
    grid = sv_image_comparison._Stage0Grid(left, right, 12, 12, (11, 1), (1, 0), 0)
    desc = nv.to_device(enumerate_grid_pairs(grid.tables(), grid.ncol, 11, grid.nrow, 1, 0, grid.nrow,
                                             left.shape[1], right.shape[1])[0])
    dl, dr = nv.to_device(left), nv.to_device(right)
    out = nv.empty(desc.shape[0], torch.float64)
    for precision, tag in ((nv.ACC_F32, 'f32'), (nv.ACC_F64, 'f64')):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 2:
                e0.record()
            nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(desc), desc.shape[0], 12, 1e-4, 1e-9,
                    precision, nv.ptr(out), None, None, None, nv.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        emit(config=3, what=f'svamd_scan_pairs ({tag}) device time, cone portal pairs', device_ms=ms,
             pairs=int(desc.shape[0]), pairs_per_s=desc.shape[0] / ms * 1e3)


def config4b():
    '''Reference-faithful dense scan arithmetic: circular 31x31 surfaces (961
    evaluations of 961 products each per pair) + maximum, device time of the
    pair kernel on one million pairs drawn like a dense scan (same-row offsets).'''
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
    out = nv.empty(n, torch.float64)
    for precision, tag in ((nv.ACC_F32, 'f32'), (nv.ACC_F64, 'f64')):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 1:
                e0.record()
            nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, 31, 1e-4, 1e-9, precision,
                    nv.ptr(out), None, None, None, nv.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        emit(config='4b', what=f'svamd_scan_pairs ({tag}) device time, 2^20 pairs of 31x31 (circular surface + max)',
             device_ms=ms, pairs_per_s=n / ms * 1e3, evals_per_s=n * 961 / ms * 1e3,
             tflops=n * 961 * 961 * 2 / ms * 1e3 / 1e12)


def config5():
    lib = nv.load_library()
    n = 1024
    for batch in (8, 64):
        g = torch.Generator(device='cuda').manual_seed(0)
        t = torch.rand((batch, n, n), generator=g, device='cuda', dtype=torch.float32)
        r = torch.rand((batch, n, n), generator=g, device='cuda', dtype=torch.float32)
        out = torch.empty((batch, n, n), device='cuda', dtype=torch.float32)
        ws = nv.workspace(lib.svamd_xcorr2d_fft_workspace_bytes(batch, n, n))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            nv.call('svamd_xcorr2d_fft', nv.ptr(t), n, n, nv.ptr(r), n, n, batch, 1e-4, 1e-9, nv.ptr(out),
                    nv.F32, nv.ptr(ws), ws.numel(), nv.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        pairs_per_s = batch / ms * 1e3
        emit(config=5, what=f'svamd_xcorr2d_fft batch {batch} of 1024x1024 float32 pairs, device time',
             device_ms=ms, pairs_per_s=pairs_per_s, surface_elems_per_s=pairs_per_s * n * n,
             algorithmic_GBps=pairs_per_s * 12.6e6 / 1e9, hbm_frac_of_8TBps=pairs_per_s * 12.6e6 / 8e12)
    # CPU: scipy/numpy FFT of one pair, the reference's own route (1 thread)
    a = np.random.default_rng(0).random((n, n))
    b = np.random.default_rng(1).random((n, n))
    t0 = time.perf_counter()
    numpy_fft_circular_ncc(a, b)
    emit(config=5, what='cpu numpy FFT route, one 1024x1024 float64 pair (1 thread)',
         seconds=time.perf_counter() - t0)


def config_grids():
    '''Displacement-grid accumulation: the dense band scan's 4096x4096 disparity
    map, and a box list shaped like a 25x25 / overlap 0.5 scan of a 4096x4096
    pair (about 116k windows), device time.'''
    import depth_grids
    lib = nv.load_library()
    H = W = 4096
    rng = np.random.default_rng(3)
    disp = nv.to_device(rng.integers(-128, 129, size=(H - 30, W - 30)).astype(np.int32))
    totals = depth_grids.DepthGrids((H, W))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(4):
        if it == 1:
            e0.record()
        totals.add_band_scan(disp, 31, 31)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    emit(config='N1', what='svamd_band_accumulate, 4066x4066 windows of 31x31 into 4096x4096 grids, device time',
         device_ms=ms, windows_per_s=disp.numel() / ms * 1e3)
    xs = np.arange(0, W, 12)
    ys = np.arange(0, H, 12)
    gx, gy = np.meshgrid(xs, ys)
    boxes = np.stack([gx.ravel(), np.minimum(gx.ravel() + 25, W), gy.ravel(), np.minimum(gy.ravel() + 25, H)],
                     axis=1).astype(np.int32)
    values = rng.standard_normal((boxes.shape[0], 3))
    values[:, 2] = np.hypot(values[:, 0], values[:, 1])
    n = boxes.shape[0]
    d_boxes, d_values = nv.to_device(boxes), nv.to_device(values)
    ws = nv.workspace(lib.svamd_accumulate_boxes_workspace_bytes(n, H, W, 25, 25))
    for it in range(4):
        if it == 1:
            e0.record()
        nv.call('svamd_accumulate_boxes', nv.ptr(d_boxes), nv.ptr(d_values), n, H, W, 25, 25,
                nv.ptr(totals.depth), nv.ptr(totals.dp_x), nv.ptr(totals.dp_y), nv.ptr(totals.contributions),
                nv.ptr(ws), ws.numel(), nv.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    emit(config='N1', what=f'svamd_accumulate_boxes, {n} boxes of 25x25 into 4096x4096 grids, device time',
         device_ms=ms, boxes_per_s=n / ms * 1e3)
    t0 = time.perf_counter()
    seq = [np.zeros((H, W)) for _ in range(4)]
    for (bx0, bx1, by0, by1), (vx, vy, vm) in zip(boxes[:20000], values[:20000]):
        seq[0][by0:by1, bx0:bx1] += vm
        seq[1][by0:by1, bx0:bx1] += vx
        seq[2][by0:by1, bx0:bx1] += vy
        seq[3][by0:by1, bx0:bx1] += 1
    dt = time.perf_counter() - t0
    emit(config='N1', what='cpu numpy painting loop (the reference\'s form), 20000 of those boxes, 1 thread',
         seconds=dt, boxes_per_s=20000 / dt)


def config_peaks():
    '''Calibration peak finder on a synthetic 1600x2400 dot-grid image (101x101
    Gaussian template, 10x10 neighbourhoods), wall clock of find_peaks.'''
    import sv_calibration
    rng = np.random.default_rng(7)
    h, w = 1600, 2400
    yy, xx = np.mgrid[0:h, 0:w]
    img = 20 + 6 * rng.standard_normal((h, w))
    for gy in range(100, h - 80, 140):
        for gx in range(100, w - 80, 140):
            cy, cx = gy + rng.uniform(-6, 6), gx + rng.uniform(-6, 6)
            ys, xs = slice(int(cy) - 30, int(cy) + 31), slice(int(cx) - 30, int(cx) + 31)
            img[ys, xs] += 200 * np.exp(-((yy[ys, xs] - cy) ** 2 + (xx[ys, xs] - cx) ** 2) / 50.0)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    dt, (x, y) = timed(sv_calibration.find_peaks, img, 10, 0.2)
    emit(config='N2', what='find_peaks, 1600x2400 uint8 image, 101x101 template (call incl. transfers)',
         seconds=dt, peaks=len(x), tflops_equiv=2.0 * 101 * 101 * h * w / dt / 1e12)
    t0 = time.perf_counter()
    grid = np.linspace(-5, 5, 101)
    xg, yg = np.meshgrid(grid, grid)
    numpy_fft_circular_ncc(np.exp(-(xg ** 2 + yg ** 2)), img.astype(np.float64))
    emit(config='N2', what='cpu numpy FFT correlation of the same pair alone (1 thread)',
         seconds=time.perf_counter() - t0)


def config_audio():
    '''Long 1D circular correlation, the shape of the reference's audio search: a
    5,000,000-sample recording as the template argument, a 200,000-sample snippet
    as the signal.'''
    rng = np.random.default_rng(5)
    nt, ns, start = 5_000_000, 200_000, 3_141_592
    recording = np.rint(3000 * rng.standard_normal(nt))
    snippet = recording[start:start + ns] + 100 * rng.standard_normal(ns)
    dt, out = timed(correlation_spectral.cross_correlate_1d_spectral, recording, snippet, repeat=2)
    emit(config='N4', what='cross_correlate_1d_spectral, 5,000,000-sample template, 200,000-sample signal '
         '(call incl. transfers)', seconds=dt, argmax=int(np.argmax(out)), expected_argmax=nt - 1 - start,
         tflops=2.0 * ns * (nt + ns - 1) / dt / 1e12)
    from scipy import fft as sfft
    t0 = time.perf_counter()
    n = nt + ns - 1
    t = (recording - recording.mean()) / (recording.std() * nt)
    padded = np.concatenate([np.zeros(nt - 1), snippet])
    ref = np.real(sfft.ifft(np.conj(sfft.fft(t, n)) * sfft.fft(padded, n)))
    emit(config='N4', what='cpu scipy.fft route of the same pair (1 thread)', seconds=time.perf_counter() - t0,
         argmax=int(np.argmax(ref)), max_abs_difference=float(np.max(np.abs(ref - out))),
         scale=float(np.max(np.abs(ref))))


def config_prepare():
    '''Image preparation on the device: RGBA -> grey with NaN transparency and a
    factor-2 downsample of a 4096x4096 image; device time between events and the
    bytes each step must move (input read once, output written once).'''
    import utils
    from scipy import ndimage
    rng = np.random.default_rng(11)
    h = w = 4096
    rgba = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    d_rgba = nv.to_device(rgba)
    grey = nv.empty((h, w), torch.float64)
    grey8 = nv.empty((h, w), torch.uint8)
    out = nv.empty((h // 2, w // 2), torch.float64)
    nbytes = nv.load_library().svamd_downsample_workspace_bytes(h, w)
    ws = nv.workspace(nbytes)

    def device_ms(fn, repeat=10):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(repeat):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / repeat

    ms = device_ms(lambda: nv.call('svamd_greyscale', nv.ptr(d_rgba), 4, h, w, 1, nv.ptr(grey8), nv.ptr(grey),
                                   nv.stream_ptr()))
    moved = h * w * (4 + 1 + 8)
    emit(config='N3', what='greyscale with NaN transparency, 4096x4096 RGBA -> uint8 + float64 (device)',
         ms=ms, gbytes_per_s=moved / ms / 1e6, fraction_of_hbm_peak=moved / ms / 1e6 / 8000)
    for code, src, label in ((nv.U8, grey8, 'uint8'), (nv.F64, grey, 'float64')):
        ms = device_ms(lambda: nv.call('svamd_downsample', nv.ptr(src), code, h, w, h // 2, w // 2, nv.ptr(out),
                                       nv.ptr(ws), nbytes, nv.stream_ptr()))
        moved = h * w * (1 if code == nv.U8 else 8) + (h // 2) * (w // 2) * 8
        emit(config='N3', what=f'downsample by 2, 4096x4096 {label} -> 2048x2048 float64 (device: value range, weights, fused filter + sampling)',
             ms=ms, algorithmic_gbytes_per_s=moved / ms / 1e6)
    host = grey8.cpu().numpy()
    dt, _ = timed(utils.downsample, host, 2)
    emit(config='N3', what='downsample(image, 2) call on a host uint8 4096x4096 image (incl. transfers)', seconds=dt)
    t0 = time.perf_counter()
    img = host * (1.0 / 255)
    ndimage.zoom(ndimage.gaussian_filter(img, 0.5, mode='mirror'), 0.5, order=1, mode='mirror', grid_mode=True)
    emit(config='N3', what='cpu scipy.ndimage route of the same image (what scikit-image resize runs, 1 thread)',
         seconds=time.perf_counter() - t0)


if __name__ == '__main__':
    which = sys.argv[1:] or ['1', '1d', '2', '3', '4b', '5', 'n1', 'n2', 'n3', 'n4']
    for c in which:
        {'1': config1, '1d': config1_device, '2': config2, 'n1': config_grids, 'n2': config_peaks, 'n3': config_prepare, 'n4': config_audio, '3': config3, '4b': config4b, '5': config5}[c]()
