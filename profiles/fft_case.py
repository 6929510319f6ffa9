#!/usr/bin/env python3
'''Runs svamd_xcorr2d_fft a few times on one case, for rocprofv3 --kernel-trace:

    rocprofv3 --kernel-trace -d DIR -o NAME -- python3 profiles/fft_case.py BATCH H W [REPS]
'''
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))

import torch                     # noqa: E402
import _native as nv             # noqa: E402


def main():
    batch, H, W = (int(a) for a in sys.argv[1:4])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    lib = nv.load_library()
    g = torch.Generator(device='cuda').manual_seed(0)
    t = torch.rand((batch, H, W), generator=g, device='cuda', dtype=torch.float32)
    r = torch.rand((batch, H, W), generator=g, device='cuda', dtype=torch.float32)
    out = torch.empty((batch, H, W), device='cuda', dtype=torch.float32)
    ws = nv.workspace(lib.svamd_xcorr2d_fft_workspace_bytes(batch, H, W))
    for _ in range(reps):
        nv.call('svamd_xcorr2d_fft', nv.ptr(t), H, W, nv.ptr(r), H, W, batch, 1e-4, 1e-9, nv.ptr(out), nv.F32,
                nv.ptr(ws), ws.numel(), nv.stream_ptr())
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
