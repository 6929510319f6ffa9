#!/usr/bin/env python3
'''Runs svamd_cross_correlate_2d a few times on one case, for rocprofv3 --kernel-trace:

    rocprofv3 --kernel-trace -d DIR -o NAME -- python3 profiles/a1_case.py TH TW RH RW STEP [REPS]
'''
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))

import torch                     # noqa: E402
import _native as nv             # noqa: E402


def main():
    th, tw, rh, rw, step = (int(a) for a in sys.argv[1:6])
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    lib = nv.load_library()
    rng = np.random.default_rng(0)
    t = nv.to_device(rng.random((th, tw)))
    r = nv.to_device(rng.integers(0, 256, size=(rh, rw), dtype=np.uint8))
    out = nv.empty((-(-rh // step), -(-rw // step)), torch.float64)
    ws = nv.workspace(lib.svamd_cross_correlate_2d_workspace_bytes(th, tw, rh, rw))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(reps + 1):
        if it == 1:
            e0.record()
        nv.call('svamd_cross_correlate_2d', nv.ptr(t), th, tw, nv.ptr(r), nv.U8, rh, rw, step, step, nv.ptr(out),
                nv.ptr(ws), ws.numel(), nv.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    print(f'{th}x{tw} over {rh}x{rw} step {step}: {e0.elapsed_time(e1) / reps:.4f} ms per call')


if __name__ == '__main__':
    main()
