'''Which HIP runtimes does a process hold after libsvamd.so and PyTorch are both
loaded, depending on who came first?  Each order runs in a fresh child process,
lists the libamdhip64 / libhsa-runtime64 files in /proc/self/maps and then makes
one call into the library (a circular NCC of a small pair, checked against
numpy's FFT route).  One JSON line per order.

  svamd_first      ctypes.CDLL(libsvamd.so) before `import torch`
  torch_imported   `import torch` (no GPU call), then the library
  torch_initialised  `import torch; torch.cuda.init()`, then the library
  module           `import correlation_spectral` and nothing else first (what a user does)
'''
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'stereo-vision_amd')

CHILD = r'''
import ctypes, json, os, sys
sys.path.insert(0, {pkg!r})
order = {order!r}
LIB = os.path.join({pkg!r}, 'libsvamd.so')

def runtimes():
    found = set()
    with open('/proc/self/maps') as maps:
        for line in maps:
            path = line.split()[-1]
            if 'libamdhip64' in path or 'libhsa-runtime64' in path:
                found.add(path)
    return sorted(found)

report = dict(order=order)
if order == 'svamd_first':
    ctypes.CDLL(LIB)
    report['after_library'] = runtimes()
    import torch
    report['after_torch_import'] = runtimes()
elif order == 'torch_imported':
    import torch
    report['after_torch_import'] = runtimes()
    ctypes.CDLL(LIB)
    report['after_library'] = runtimes()
elif order == 'torch_initialised':
    import torch
    torch.cuda.init()
    ctypes.CDLL(LIB)
    report['after_library'] = runtimes()
try:
    import numpy as np
    import correlation_spectral
    report['after_module_import'] = runtimes()
    rng = np.random.default_rng(0)
    t, r = rng.random((20, 24)), rng.random((20, 24))
    got = correlation_spectral.cross_correlate_2d_spectral(t, r)
    tn = (t - t.mean()) / (t.std() * t.size)
    rn = (r - r.mean()) / r.std()
    want = np.real(np.fft.ifft2(np.conj(np.fft.fft2(tn)) * np.fft.fft2(rn)))
    report['max_error'] = float(np.abs(got - want).max())
    report['call'] = 'ok'
except Exception as exc:          # the failure is the finding
    report['call'] = f'{{type(exc).__name__}}: {{exc}}'
report['final'] = runtimes()
print(json.dumps(report))
'''


def main():
    for order in ('svamd_first', 'torch_imported', 'torch_initialised', 'module'):
        code = CHILD.format(pkg=PKG, order=order)
        done = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
        lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
        if lines:
            print(lines[-1], flush=True)
        else:
            print(json.dumps(dict(order=order, rc=done.returncode, stderr=done.stderr[-600:])), flush=True)


if __name__ == '__main__':
    main()
