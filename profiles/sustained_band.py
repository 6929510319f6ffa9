#!/usr/bin/env python3
'''How the band kernel's duration depends on what surrounds the launch: spaced
single launches, launches queued back to back, and back to back with the table
kernels (K1) in between as in bench.py.  HIP events around every launch.'''
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import bench            # noqa: E402
import band_scan as bs  # noqa: E402

H = W = 4096
left, right = bench.synthetic_pair(H, W, 1000)
sc = bs.BandScanner(H, W, 31, 31, 128)
sc.upload(left, right)
sc.prepare()
sc.run(0.5, prepared=True)
torch.cuda.synchronize()


def series(n, prepare, sync):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        if prepare:
            sc.prepare()
        a.record()
        sc.run(0.5, prepared=True)
        b.record()
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev])


for name, prepare, sync in (('spaced (sync after every launch)', False, True),
                            ('back to back', False, False),
                            ('back to back with K1 in between', True, False),
                            ('spaced again', False, True)):
    t = series(12, prepare, sync)
    print(f'{name:<36} ms: {np.round(t, 1)}  mean of last 8: {t[4:].mean():.2f}')
