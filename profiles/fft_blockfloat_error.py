'''What would 16-bit block-floating-point spectra cost in accuracy?  The three-pass
circular NCC of a 1024 x 1024 pair (row transforms, column transforms with the
conjugate product and the inverse column transform, inverse row transforms) is
emulated in numpy with float32 arithmetic, once with float32 intermediates (what the
kernels store between passes today) and once with both intermediates - the half
spectra after pass 1 and the product after pass 2 - rounded to blocks of `block`
complex values sharing one exponent with 16-bit mantissas.  Errors are against a float64
evaluation, in the units of the parity bar (absolute, surfaces live on [-1, 1]; the bar
is 1e-5).  Inputs: the seeded random pair of the g4 rand1024 fixture, a planted circular
shift of it, and a smooth (photograph-like spectrum) pair.  CPU only.'''
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests')]
from helpers import rand_pair               # noqa: E402


def block_float(z, block, bits=16):
    '''z: complex64 [..., n] -> the same values with, per run of `block` elements along the
    last axis, one shared power-of-two scale and (bits)-bit signed mantissas for the real
    and imaginary parts.'''
    shape = z.shape
    v = z.reshape(-1, block)
    peak = np.maximum(np.abs(v.real).max(axis=1), np.abs(v.imag).max(axis=1))
    exp = np.ceil(np.log2(np.maximum(peak, 1e-30)))
    scale = np.exp2(exp - (bits - 1)).astype(np.float32)[:, None]
    lim = 2 ** (bits - 1) - 1
    re = np.clip(np.rint(v.real / scale), -lim, lim) * scale
    im = np.clip(np.rint(v.imag / scale), -lim, lim) * scale
    return (re + 1j * im).astype(np.complex64).reshape(shape)


def surface(t, r, quantise=None):
    tn = ((t - t.mean(dtype=np.float64)) / (t.std(dtype=np.float64) * t.size)).astype(np.float32)
    rn = ((r - r.mean(dtype=np.float64)) / r.std(dtype=np.float64)).astype(np.float32)
    # pass 1: rows (half spectra), float32
    T = np.fft.rfft(tn, axis=1).astype(np.complex64)
    R = np.fft.rfft(rn, axis=1).astype(np.complex64)
    if quantise:
        T, R = quantise(T[:, :512]), quantise(R[:, :512])       # (the Nyquist column left out of the comparison)
        T = np.concatenate([T, np.fft.rfft(tn, axis=1).astype(np.complex64)[:, 512:]], axis=1)
        R = np.concatenate([R, np.fft.rfft(rn, axis=1).astype(np.complex64)[:, 512:]], axis=1)
    # pass 2: columns, product, inverse columns
    P = np.fft.ifft(np.conj(np.fft.fft(T, axis=0).astype(np.complex64)) *
                    np.fft.fft(R, axis=0).astype(np.complex64), axis=0).astype(np.complex64)
    if quantise:
        P = np.concatenate([quantise(np.ascontiguousarray(P[:, :512])), P[:, 512:]], axis=1)
    # pass 3: inverse rows
    return np.fft.irfft(P, n=1024, axis=1).astype(np.float32)


def exact(t, r):
    t, r = t.astype(np.float64), r.astype(np.float64)
    tn = (t - t.mean()) / (t.std() * t.size)
    rn = (r - r.mean()) / r.std()
    return np.real(np.fft.ifft2(np.conj(np.fft.fft2(tn)) * np.fft.fft2(rn)))


def main():
    t, r, _ = rand_pair(1024)
    rng = np.random.default_rng(5)
    smooth = rng.random((1024, 1024)).astype(np.float32)
    for _ in range(6):
        smooth = (np.roll(smooth, 1, 1) + 2 * smooth + np.roll(smooth, -1, 1)) / 4
        smooth = (np.roll(smooth, 1, 0) + 2 * smooth + np.roll(smooth, -1, 0)) / 4
    yy, xx = np.mgrid[0:1024, 0:1024]
    smooth = (smooth + 0.3 * np.sin(xx / 150.0) * np.cos(yy / 90.0)).astype(np.float32)
    cases = (('g4 rand1024 pair', t, r), ('planted shift of the rand1024 template', t, np.roll(t, (37, 210), (0, 1))),
             ('smooth pair, shifted', smooth, np.roll(smooth, (5, 9), (0, 1)) + 0.01 * rng.random((1024, 1024)).astype(np.float32)))
    for name, a, b in cases:
        want = exact(a, b)
        row = dict(case=name, surface_max=float(np.abs(want).max()),
                   float32_intermediates=float(np.abs(surface(a, b) - want).max()))
        for block in (16, 64):
            got = surface(a, b, lambda z: block_float(z, block))
            row[f'blockfloat16_block{block}'] = float(np.abs(got - want).max())
            row[f'argmax_kept_block{block}'] = bool(np.argmax(got) == np.argmax(want))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
