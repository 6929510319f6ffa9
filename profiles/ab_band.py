#!/usr/bin/env python3
'''A/B timing of band-kernel builds in ONE process on one device: each library
given on the command line is loaded separately and the 4096x4096 / 31x31 / 257
offset scan is timed with HIP events in interleaved rounds.

    python3 profiles/ab_band.py stereo-vision_amd/libsvamd.so /path/to/variant.so ...

A path may carry extra flag bits for svamd_band_scan_u8 as path@FLAGS (e.g. @2 =
tile-at-a-time kernel, @0x6400 = streaming kernel with 100-row segments).
'''
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H = W = 4096
WIN, MD = 31, int(os.environ.get('AB_MAX_DISP', 128))
WIN_W = int(os.environ.get('AB_WIN_W', WIN))      # widths above 32 take the run-time-width kernel


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i, d, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.svamd_band_scan_u8_workspace_bytes.restype = sz
    lib.svamd_band_scan_u8_workspace_bytes.argtypes = [i, i, i, i]
    lib.svamd_band_scan_u8.restype = i
    lib.svamd_band_scan_u8.argtypes = [vp, vp, i, i, i, i, i, i, i, i, d, d, i, i, i, vp, vp, vp, vp, sz, vp]
    lib.svamd_last_error.restype = ctypes.c_char_p
    return lib


def main():
    paths = sys.argv[1:]
    extra = [int(p.split('@')[1], 0) if '@' in p else 0 for p in paths]
    libs = [load(p.split('@')[0]) for p in paths]
    left, right = bench.synthetic_pair(H, W, 1000)
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    nwy, nwx = H - WIN + 1, W - WIN_W + 1
    disp = torch.empty((nwy, nwx), dtype=torch.int32, device='cuda')
    val = torch.empty((nwy, nwx), dtype=torch.float64, device='cuda')
    ws = torch.empty(libs[0].svamd_band_scan_u8_workspace_bytes(H, W, WIN_W, WIN), dtype=torch.uint8, device='cuda')
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(lib, flags):
        rc = lib.svamd_band_scan_u8(dl.data_ptr(), dr.data_ptr(), H, W, WIN_W, WIN, 1, 1, MD, 1, 0.5, 0.01,
                                    0, nwy, flags, disp.data_ptr(), val.data_ptr(), None, ws.data_ptr(),
                                    ws.numel(), stream)
        assert rc == 0, lib.svamd_last_error()

    ref = None
    for lib, fl in zip(libs, extra):
        run(lib, fl)
        torch.cuda.synchronize()
        out = disp.clone()
        if ref is None:
            ref = out
        else:
            assert os.environ.get('AB_NO_CHECK') or torch.equal(ref, out), 'variants disagree'
    rounds = 6
    times = [[] for _ in libs]
    for _ in range(rounds):
        for k, (lib, fl) in enumerate(zip(libs, extra)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(lib, 1 | fl)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1))
    for p, t in zip(paths, times):
        print(f'win {WIN_W}x{WIN} {os.path.basename(p):<40} median {np.median(t):8.3f} ms  min {np.min(t):8.3f}  all {np.round(t, 2)}')


if __name__ == '__main__':
    main()
