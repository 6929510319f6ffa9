'''svamd_scan_pairs (float32) on random boxes of sides 12, 16 and 31 taken from images of 128^2 ..
4096^2 pixels: the time does not depend on where the pixels live.
'''
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import _native as nv
rng = np.random.default_rng(4)
for H in (128, 256, 1024, 4096):
    W = H
    left = rng.random((H, W)); right = rng.random((H, W))
    dl, dr = nv.to_device(left), nv.to_device(right)
    for side in (12, 16, 31):
        n = 1 << 20 if side < 31 else 1 << 17
        ty = rng.integers(0, H - side, size=n); tx = rng.integers(0, W - side, size=n); rx = rng.integers(0, W - side, size=n)
        desc = np.empty((n, 8), dtype=np.int64)
        desc[:, 0] = ty * W + tx; desc[:, 4] = ty * W + rx; desc[:, 1] = desc[:, 5] = W
        desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = side
        dd = nv.to_device(desc); out = nv.empty(n, torch.float64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 1: e0.record()
            nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, side, 1e-4, 1e-9, nv.ACC_F32, nv.ptr(out), None, None, None, nv.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(f'image {H:5d}^2 side {side:3d} n={n:8d} {ms:8.3f} ms  {n * 2.0 * side ** 4 / ms / 1e9:7.1f} TFLOP/s', flush=True)
