'''The dense band scan on floating-point images (svamd_band_scan_f64) next to the 8-bit
kernel on the same geometry: n x n pair (default 4096), 31 x 31 window, 257 offsets,
selection rule.  The float pair is the 8-bit synthetic pair smoothed by a 3 x 3 binomial
(values that are no 8-bit levels).  Prints device times, the share of windows that went
to the float64 second look, and whether both scans agree on the disparities of the
8-bit pair when it is fed as floats (scaled by 1/254, i.e. not recognisable as 8-bit).'''
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'stereo-vision_amd'), ROOT]
import torch                                    # noqa: E402
import bench                                    # noqa: E402
import band_scan as bs                          # noqa: E402
import sv_image_comparison as v                 # noqa: E402


def smooth(image):
    f = image.astype(np.float64) / 255.0
    f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    return (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4


def device_ms(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    left, right = bench.synthetic_pair(n, n, seed=1000)
    evals = bench.count_evals(n, n, 31, 128)
    sc = bs.BandScanner(n, n, 31, 31, 128)
    sc.upload(left, right)
    sc.prepare()
    ms = device_ms(lambda: sc.run(0.5, prepared=True))
    want = sc.disp.cpu().numpy().copy()
    print(json.dumps(dict(what=f'band scan {n}x{n}, 8-bit kernel', kernel_ms=ms,
                          tflops=evals * 1922 / ms * 1e3 / 1e12)), flush=True)
    for tag, l, r in (('8-bit pair / 254', left / 254.0, right / 254.0), ('smoothed pair', smooth(left), smooth(right))):
        sc.upload(l, r)
        assert sc.kind == 'f64'
        prep = device_ms(sc.prepare)
        for guard in (bs.BAND_GUARD_BAND, 0.0):
            v.SCAN_PRECISION = 'mixed' if guard else 'fp32'
            ms = device_ms(lambda: sc.run(0.5, prepared=True))
            out = dict(what=f'band scan {n}x{n}, float kernel, {tag}', guard_band=guard, prepare_ms=prep, scan_ms=ms,
                       tflops_float32_part=evals * 1922 / ms * 1e3 / 1e12)
            if guard:
                out['windows_evaluated_again'] = sc.rechecked()
                out['windows'] = int(sc.disp.numel())
                out['disparities_equal_8bit_scan'] = bool(np.array_equal(sc.disp.cpu().numpy(), want)) \
                    if tag.startswith('8-bit') else None
            print(json.dumps(out), flush=True)
        v.SCAN_PRECISION = 'mixed'


if __name__ == '__main__':
    main()
