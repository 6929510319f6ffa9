#!/usr/bin/env python3
'''Timing of cross_correlate_2d_spectral on large pairs (host call: upload, kernels,
download of the float64 surface) and of the device part alone, next to numpy's FFT
evaluation of the same reference lines on one host core.  One JSON line per case.'''
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
import correlation_spectral as cs  # noqa: E402


def numpy_route(t, r):
    tt = (t - t.mean()) / (t.std() * t.size)
    rr = (r - r.mean()) / r.std()
    return np.real(np.fft.ifft2(np.conj(np.fft.fft2(tt, s=r.shape)) * np.fft.fft2(rr)))


def main():
    rng = np.random.default_rng(0)
    lib = nv.load_library()
    for tshape, rshape in (((2048, 2048), (2048, 2048)), ((4096, 4096), (4096, 4096)), ((1000, 1200), (1000, 1200)),
                           ((1500, 2000), (1500, 2000)), ((700, 900), (1000, 1200))):
        t, r = rng.random(tshape), rng.random(rshape)
        cs.cross_correlate_2d_spectral(t, r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cs.cross_correlate_2d_spectral(t, r)
        call_s = time.perf_counter() - t0
        # device part: the any-shape entry point on resident float64 arrays
        dt, dr = nv.to_device(t), nv.to_device(r)
        dout = nv.empty(rshape, torch.float64)
        ws = nv.workspace(lib.svamd_cross_correlate_2d_spectral_workspace_bytes(*tshape, *rshape))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(4):
            if it == 1:
                e0.record()
            nv.call('svamd_cross_correlate_2d_spectral', nv.ptr(dt), nv.F64, *tshape, nv.ptr(dr), nv.F64, *rshape,
                    1e-4, 1e-9, nv.ptr(dout), nv.ptr(ws), ws.numel(), nv.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / 3
        t0 = time.perf_counter()
        want = numpy_route(t, r)
        cpu_s = time.perf_counter() - t0
        err = float(np.max(np.abs(out - want)))
        print(json.dumps({'what': f'cross_correlate_2d_spectral template {tshape} region {rshape}',
                          'call_ms': call_s * 1e3, 'device_ms_any_shape_entry': dev_ms,
                          'numpy_fft_one_core_ms': cpu_s * 1e3, 'max_abs_err_vs_numpy_f64': err}), flush=True)


if __name__ == '__main__':
    main()
