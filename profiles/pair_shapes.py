'''Which workgroup shape suits the float32 pair kernel at each box side: columns per task
(8 or 4), wavefronts per workgroup (1-4) and pairs per workgroup, set through
SVAMD_PAIR_SHAPE for the measurement.  Random square boxes of one side per call, device
time between HIP events, results checked against the default shape's.  Prints the default
and every candidate that fits, best first.'''
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import _native as nv                            # noqa: E402

rng = np.random.default_rng(4)
H = W = 1024
left = rng.random((H, W))
right = rng.random((H, W))
dl, dr = nv.to_device(left), nv.to_device(right)
sides = [int(a) for a in sys.argv[1:]] or [17, 20, 22, 24, 25, 28, 31, 36, 40, 44, 48, 50, 56, 60, 64]


def run(dd, n, side, out):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 1:
            e0.record()
        nv.call('svamd_scan_pairs', nv.ptr(dl), nv.ptr(dr), nv.ptr(dd), n, side, 1e-4, 1e-9, nv.ACC_F32, nv.ptr(out),
                None, None, None, nv.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2


for side in sides:
    n = max(1 << 14, min(1 << 19, int(1e11 / (2.0 * side ** 4))))
    ty = rng.integers(0, H - side, size=n)
    tx = rng.integers(0, W - side, size=n)
    rx = rng.integers(0, W - side, size=n)
    desc = np.empty((n, 8), dtype=np.int64)
    desc[:, 0] = ty * W + tx
    desc[:, 4] = ty * W + rx
    desc[:, 1] = desc[:, 5] = W
    desc[:, 2] = desc[:, 3] = desc[:, 6] = desc[:, 7] = side
    dd = nv.to_device(desc)
    out = nv.empty(n, torch.float64)
    os.environ.pop('SVAMD_PAIR_SHAPE', None)
    run(dd, n, side, out)                     # clocks up
    base_ms = run(dd, n, side, out)
    want = out.clone()
    flops = n * 2.0 * side ** 4
    rows = []
    for G in (4, 8):
        cols = (side + G - 1) // G * G
        tasks = (side + 1) // 2 * (cols // G)
        for waves in (1, 2, 3, 4):
            lanes = 64 * waves
            for slots in sorted({1, max(1, lanes // tasks)}):
                if slots > 16:
                    continue
                os.environ['SVAMD_PAIR_SHAPE'] = f'{G},{waves},{slots}'
                out.fill_(float('nan'))
                try:
                    ms = run(dd, n, side, out)
                except nv.NativeError:
                    continue
                same = bool(torch.equal(out, want)) or float((out - want).abs().max()) < 1e-5
                util = slots * tasks / (lanes * (-(-tasks // lanes) if slots == 1 else 1))
                rows.append((ms, G, waves, slots, util, same))
    rows.sort()
    print(f'side {side}: default {base_ms:.3f} ms = {flops / base_ms / 1e9:.1f} TFLOP/s', flush=True)
    for ms, G, waves, slots, util, same in rows[:5]:
        print(f'    G={G} waves={waves} slots={slots} lanes used {util:.2f}: {ms:.3f} ms = {flops / ms / 1e9:.1f} TFLOP/s'
              f'{"" if same else "  RESULTS DIFFER"}', flush=True)
