'''End-to-end time of sequence_scan on the cone pair with the shipped portal
config (12x12 windows, 11x1 scheme; 34 122 pairs), as the host sees it: plain
numpy images (uploaded by the call) and resident images (what utils.downsample
returns), plus the share of the stage's kernels.  One JSON line per case.'''
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'stereo-vision_amd'), ROOT]
import torch                                    # noqa: E402
import sv_image_comparison                      # noqa: E402
import utils                                    # noqa: E402
import _native as nv                            # noqa: E402

PORTAL = [{'window_width': 12, 'window_height': 12, 'scheme': [11, 1], 'scheme_shift_size': [1, 0],
           'window_overlap': 0, 'correlation_threshold': 0.93}]


def timed(fn, repeat=20):
    best = []
    for _ in range(repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = fn()
        torch.cuda.synchronize()
        best.append(time.perf_counter() - t0)
    best.sort()
    return best[0], best[len(best) // 2], out


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g6_cone_images.npz'))
    left, right = g['left'] / 255.0, g['right'] / 255.0
    for tag, l, r in (('numpy images', left, right),
                      ('resident images', utils.ResidentImage(left, nv.to_device(left)),
                       utils.ResidentImage(right, nv.to_device(right)))):
        lo, med, res = timed(lambda: sv_image_comparison.sequence_scan(l, r, PORTAL))
        moved = sum(1 for w in res[0].values() if w['dp_x'] or w['dp_y'])
        print(json.dumps(dict(what=f'sequence_scan cone/portal, {tag}', min_ms=lo * 1e3, median_ms=med * 1e3,
                              windows=len(res[0]), nonzero=moved)), flush=True)
        lo, med, res = timed(lambda: len(sv_image_comparison.sequence_scan(l, r, PORTAL)[0]))
        print(json.dumps(dict(what=f'... without reading the windows, {tag}', min_ms=lo * 1e3,
                              median_ms=med * 1e3)), flush=True)


if __name__ == '__main__':
    main()
