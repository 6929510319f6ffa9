#!/usr/bin/env python3
'''Throughput of svamd_scan_pairs against the box side: random square boxes of one
size per call, both accumulation precisions, device time between HIP events.

    python3 profiles/pair_sizes.py
'''
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import _native as nv
rng = np.random.default_rng(4)
H = W = 1024
left = rng.random((H, W)); right = rng.random((H, W))
dl, dr = nv.to_device(left), nv.to_device(right)
SIDES = [int(a) for a in sys.argv[1:]] or [8, 12, 16, 20, 24, 25, 31, 32, 40, 48, 60, 64]
for side in SIDES:
    n = max(1 << 14, min(1 << 20, int(2e11 / (2.0 * side ** 4))))
    ty = rng.integers(0, H - side, size=n); tx = rng.integers(0, W - side, size=n); rx = rng.integers(0, W - side, size=n)
    desc = np.empty((n, 8), dtype=np.int64)
    desc[:, 0] = ty * W + tx; desc[:, 4] = ty * W + rx; desc[:, 1] = desc[:, 5] = W
    desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = side
    dd = nv.to_device(desc); out = nv.empty(n, torch.float64)
    for prec, tag in ((nv.ACC_F32, 'f32'), (nv.ACC_F64, 'f64')):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 1: e0.record()
            nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, side, 1e-4, 1e-9, prec, nv.ptr(out), None, None, None, nv.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(f'side {side:3d} {tag} n={n:8d} {ms:8.3f} ms  {n / ms * 1e3:12.0f} pairs/s  {n * 2.0 * side ** 4 / ms / 1e9:7.1f} TFLOP/s', flush=True)
