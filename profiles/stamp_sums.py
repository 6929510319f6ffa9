#!/usr/bin/env python3
'''Where the wavefronts of the box-sum band kernel spend their clocks: loads a library whose
csrc/band_sums.hip was compiled with -DSVAMD_SUMS_STAMPS (lane 0 of every wavefront adds its
clocks per phase to eight counters handed in through the `scores` argument), scans the
benchmark's 4096 x 4096 pair once and prints each phase's share.

    hipcc ... -DSVAMD_SUMS_STAMPS -c csrc/band_sums.hip ; link as variants/lib_sums_stamps.so
    python3 profiles/stamp_sums.py variants/lib_sums_stamps.so
'''
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ['loop head + wait for the slowest phase C', 'A: products, row and block prefix sums',
          'wait for the slowest phase A', 'B: block carries + wait for the slowest', '-', '-', 'C: scores and selection', '-']


def main(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i, d, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.svamd_band_scan_u8_workspace_bytes.restype = sz
    lib.svamd_band_scan_u8_workspace_bytes.argtypes = [i, i, i, i]
    lib.svamd_band_scan_u8.restype = i
    lib.svamd_band_scan_u8.argtypes = [vp, vp, i, i, i, i, i, i, i, i, d, d, i, i, i, vp, vp, vp, vp, sz, vp]
    H = W = 4096
    win, md = 31, 128
    left, right = bench.synthetic_pair(H, W, 1000)
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    nwy = nwx = H - win + 1
    disp = torch.empty((nwy, nwx), dtype=torch.int32, device='cuda')
    val = torch.empty((nwy, nwx), dtype=torch.float64, device='cuda')
    ws = torch.empty(lib.svamd_band_scan_u8_workspace_bytes(H, W, win, win), dtype=torch.uint8, device='cuda')
    stamps = torch.zeros(8, dtype=torch.int64, device='cuda')
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.svamd_band_scan_u8(dl.data_ptr(), dr.data_ptr(), H, W, win, win, 1, 1, md, 1, 0.5, 0.01, 0, nwy, 2,
                                disp.data_ptr(), val.data_ptr(), stamps.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().astype(np.float64)
    for name, v in zip(PHASES, t):
        print(f'{name:<48} {100.0 * v / t.sum():5.1f} %')


if __name__ == '__main__':
    main(sys.argv[1])
