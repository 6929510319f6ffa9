#!/usr/bin/env python3
'''A/B timing of libsvamd builds on the batched FFT correlation (AB_BATCH pairs, default
64, of AB_H x AB_W, default 1024 x 1024; float32 output), HIP events, interleaved rounds:

    python3 profiles/ab_fft.py stereo-vision_amd/libsvamd.so /path/to/variant.so ...
'''
import ctypes
import os
import sys

import numpy as np
import torch


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i, d, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.svamd_xcorr2d_fft_workspace_bytes.restype = sz
    lib.svamd_xcorr2d_fft_workspace_bytes.argtypes = [i, i, i]
    lib.svamd_xcorr2d_fft.restype = i
    lib.svamd_xcorr2d_fft.argtypes = [vp, i, i, vp, i, i, i, d, d, vp, i, vp, sz, vp]
    lib.svamd_last_error.restype = ctypes.c_char_p
    return lib


def main():
    paths = sys.argv[1:]
    libs = [load(p) for p in paths]
    batch = int(os.environ.get('AB_BATCH', 64))
    H, W = int(os.environ.get('AB_H', 1024)), int(os.environ.get('AB_W', 1024))
    g = torch.Generator(device='cuda').manual_seed(0)
    t = torch.rand((batch, H, W), generator=g, device='cuda', dtype=torch.float32)
    r = torch.rand((batch, H, W), generator=g, device='cuda', dtype=torch.float32)
    out = torch.empty((batch, H, W), device='cuda', dtype=torch.float32)
    ws = torch.empty(libs[0].svamd_xcorr2d_fft_workspace_bytes(batch, H, W), dtype=torch.uint8, device='cuda')
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # AB_FLUSH=1: a 1 GiB fill before every call pushes the buffers out of the 256 MB Infinity Cache
    flush = torch.empty(1 << 30, dtype=torch.uint8, device='cuda') if os.environ.get('AB_FLUSH') else None

    def run(lib):
        if flush is not None:
            flush.fill_(1)
        rc = lib.svamd_xcorr2d_fft(t.data_ptr(), H, W, r.data_ptr(), H, W, batch, 1e-4, 1e-9, out.data_ptr(), 1,
                                   ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, lib.svamd_last_error()

    ref = None
    for p, lib in zip(paths, libs):
        run(lib)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            print(f'{os.path.basename(p)}: max |difference| to the first library {float((out - ref).abs().max()):.3e}')
    times = [[] for _ in libs]
    for _ in range(6):
        for k, lib in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(lib)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 3)
    for p, tm in zip(paths, times):
        print(f'{os.path.basename(p):<40} median {np.median(tm):8.4f} ms  min {np.min(tm):8.4f}')


if __name__ == '__main__':
    main()
