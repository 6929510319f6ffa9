#!/usr/bin/env python3
'''svamd_xcorr2d_fft on shapes of 256 ... 4096 points per side: largest deviation from
scipy's float64 FFT evaluation of the reference lines (correlation_spectral.py:37-54) and
device time (HIP events).  One JSON line per case.  FFT_WIDE_CASES=quick runs a subset.
(profiles/r03_fft_wide_ab.jsonl is this script's output from the commit that still held the
workgroup-per-transform kernels and the tuned 1024 x 1024 kernels next to the new ones.)'''
import ctypes
import json
import os
import sys

import numpy as np
import scipy.fft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))

import torch                     # noqa: E402
import _native as nv             # noqa: E402


def reference(t, r):
    t, r = t.astype(np.float64), r.astype(np.float64)
    tt = (t - t.mean()) / (t.std() * t.size)
    rr = (r - r.mean()) / r.std()
    ft = scipy.fft.rfft2(tt[:r.shape[0], :r.shape[1]], s=r.shape, workers=16)
    return scipy.fft.irfft2(np.conj(ft) * scipy.fft.rfft2(rr, workers=16), s=r.shape, workers=16)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return float(np.median(best))


def main():
    lib = nv.load_library()
    quick = os.environ.get('FFT_WIDE_CASES') == 'quick'
    cases = [  # batch, template shape, region shape
        (1, (256, 256), (256, 256)), (3, (512, 512), (512, 512)), (2, (1024, 1024), (1024, 1024)),
        (1, (2048, 2048), (2048, 2048)), (1, (4096, 4096), (4096, 4096)),
        (2, (256, 512), (256, 512)), (1, (2048, 256), (2048, 256)), (1, (512, 4096), (512, 4096)),
        (2, (300, 200), (512, 256)), (1, (700, 3000), (512, 2048)), (1, (4096, 1024), (4096, 1024)),
    ]
    if not quick:
        cases += [(64, (512, 512), (512, 512)), (64, (256, 256), (256, 256)), (64, (1024, 1024), (1024, 1024)),
                  (16, (2048, 2048), (2048, 2048)), (4, (4096, 4096), (4096, 4096))]
    g = torch.Generator(device='cuda').manual_seed(0)
    for batch, (th, tw), (H, W) in cases:
        t = torch.rand((batch, th, tw), generator=g, device='cuda', dtype=torch.float32)
        r = torch.rand((batch, H, W), generator=g, device='cuda', dtype=torch.float32)
        out = torch.empty((batch, H, W), device='cuda', dtype=torch.float32)
        ws = nv.workspace(lib.svamd_xcorr2d_fft_workspace_bytes(batch, H, W))

        def run():
            nv.call('svamd_xcorr2d_fft', nv.ptr(t), th, tw, nv.ptr(r), H, W, batch, 1e-4, 1e-9, nv.ptr(out),
                    nv.F32, nv.ptr(ws), ws.numel(), nv.stream_ptr())

        line = {'batch': batch, 'template': [th, tw], 'region': [H, W]}
        reps = 3 if batch * H * W >= 1 << 24 else 10
        line['wide_ms'] = timed(run, reps)
        if batch <= 4:
            got = out.cpu().numpy()
            err = 0.0
            for b in (0, batch - 1):
                want = reference(t[b].cpu().numpy(), r[b].cpu().numpy())
                err = max(err, float(np.max(np.abs(got[b] - want))))
            line['wide_max_abs_err'] = err
        moved = batch * H * W * 4 * 3                      # two float32 inputs + one float32 surface
        line['algorithmic_TBps'] = moved / (line['wide_ms'] * 1e-3) / 1e12
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
