'''2^20 random square pairs of one side (default 12) through svamd_scan_pairs in float32, three
calls: the program the counter passes of profiles/r02_pair12_counters.txt were taken over.

    python3 profiles/pair_one_side.py [side]
'''
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import _native as nv
rng = np.random.default_rng(4)
H = W = 1024
side = int(sys.argv[1]) if len(sys.argv) > 1 else 12
left = rng.random((H, W)); right = rng.random((H, W))
dl, dr = nv.to_device(left), nv.to_device(right)
n = 1 << 20
ty = rng.integers(0, H - side, size=n); tx = rng.integers(0, W - side, size=n); rx = rng.integers(0, W - side, size=n)
desc = np.empty((n, 8), dtype=np.int64)
desc[:, 0] = ty * W + tx; desc[:, 4] = ty * W + rx; desc[:, 1] = desc[:, 5] = W
desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = side
dd = nv.to_device(desc); out = nv.empty(n, torch.float64)
for it in range(3):
    nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, side, 1e-4, 1e-9, nv.ACC_F32, nv.ptr(out), None, None, None, nv.stream_ptr())
torch.cuda.synchronize()
