#!/usr/bin/env python3
'''Where a workgroup of the float32 pair kernel spends its time: loads a library
built with -DSV_PAIR_STAMPS (phase clocks accumulated per workgroup by thread 0),
runs 2^18 pairs of 31x31 and prints each phase's share.

    hipcc ... -DSV_PAIR_STAMPS -c csrc/scan_pairs.hip ; link as variants_tmp/lib_stamps.so
    python3 profiles/stamp_pairs.py variants_tmp/lib_stamps.so
'''
import ctypes
import os
import sys

import numpy as np
import torch

PHASES_MFMA = ['pixels arrived', 'statistics', 'staging', 'next requests issued', 'products', 'maximum',
               'descriptor of the next pair', '-']
PHASES = ['first pass: loads issued and summed', 'loads arrived, means reduced', 'deviations, LDS copies',
          'row-pair image', 'surfaces', 'maxima', 'descriptor read', 'template slot zeroed']


def main(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i, i64, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    lib.svamd_scan_pairs.restype = i
    lib.svamd_scan_pairs.argtypes = [vp, vp, vp, i64, i, d, d, i, vp, vp, vp, vp, vp]
    lib.svamd_debug_pair_stamps.restype = i
    lib.svamd_debug_pair_stamps.argtypes = [vp, i]
    rng = np.random.default_rng(4)
    H = 1024
    W = int(os.environ.get('STAMP_W', 1024))        # row pitch of the images, in elements
    side = int(os.environ.get('STAMP_SIDE', 31))
    left = torch.from_numpy(rng.random((H, W))).cuda()
    right = torch.from_numpy(rng.random((H, W))).cuda()
    n = 1 << 18
    ty, tx = rng.integers(0, H - side, size=n), rng.integers(0, W - side, size=n)
    if os.environ.get('STAMP_SAME_PLACE'):       # every pair at one position: all loads hit the nearest cache
        ty[:], tx[:] = 100, 200
    desc = np.empty((n, 8), dtype=np.int64)
    desc[:, 0] = ty * W + tx
    desc[:, 4] = ty * W + rng.integers(0, W - side, size=n)
    desc[:, 1] = desc[:, 5] = W
    desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = side
    dd = torch.from_numpy(desc).cuda()
    out = torch.empty(n, dtype=torch.float64, device='cuda')
    stream = vp(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = lib.svamd_scan_pairs(left.data_ptr(), right.data_ptr(), dd.data_ptr(), n, side, 1e-4, 1e-9, 0,
                                  out.data_ptr(), None, None, None, stream)
        assert rc == 0

    run()
    torch.cuda.synchronize()
    lib.svamd_debug_pair_stamps(None, 1)
    run()
    torch.cuda.synchronize()
    stamps = (ctypes.c_ulonglong * 8)()
    lib.svamd_debug_pair_stamps(stamps, 0)
    total = sum(stamps)
    names = PHASES_MFMA if side <= 32 and os.environ.get('SVAMD_PAIR_MFMA') != '0' else PHASES
    for name, v in zip(names, stamps):
        print(f'{name:<32} {v / n:10.0f} clocks per pair  {100.0 * v / total:5.1f} %')


if __name__ == '__main__':
    main(sys.argv[1])
