'''Model of the wide transforms of csrc/fft_xcorr.hip (N = 256 ... 4096, 16 elements per
thread, N / 16 threads per transform): simulates every thread's registers and the LDS
exchanges with the index formulas of the kernels, checks the result against numpy's FFT,
and counts LDS bank conflicts of every exchange pattern under the swizzle.

Banking rules (8-byte accesses): reads are served per half wavefront (32 lanes, 32
eight-byte bank pairs = 256 bytes), writes per 16 consecutive lanes (16 bank pairs = 128
bytes).  Run: python profiles/fft_wide_model.py
'''
import itertools

import numpy as np


def radices(log2n):
    r = [4] * (log2n // 2)
    if log2n & 1:
        r.append(2)
    return r


def pos_of_freq(k, log2n):
    '''Position at which the in-place forward transform leaves frequency k.'''
    n = 1 << log2n
    pos, span = 0, n
    for r in radices(log2n):
        span //= r
        pos += (k % r) * span
        k //= r
    return pos


# ---- swizzles: idx(p) = p ^ f(p), XOR-linear -----------------------------------------
def swizzle(p, log2n):
    if log2n == 12:
        return p ^ ((p >> 4) & 15) ^ (((p >> 8) & 1) << 4)
    if log2n == 11:
        return p ^ ((p >> 4) & 7) ^ (((p >> 7) & 1) * 0b01001) ^ (((p >> 8) & 1) << 4)
    if log2n == 10:
        return p ^ ((p >> 4) & 3) ^ (((p >> 6) & 1) * 0b00101) ^ (((p >> 7) & 3) << 3)
    if log2n == 9:
        return p ^ ((p >> 4) & 1) ^ (((p >> 5) & 1) * 0b00011) ^ (((p >> 6) & 7) << 2)
    if log2n == 8:
        return p ^ ((p >> 4) & 15)
    raise ValueError(log2n)


class Plan:
    def __init__(self, log2n):
        self.log2n = log2n
        self.n = 1 << log2n
        self.t = self.n // 16
        self.sb = log2n - 8
        self.s = 1 << self.sb
        self.lc = self.s            # length of the blocks of the last phase (1: no last phase)

    def p1(self, t, m):             # phase A registers
        return t + self.t * m

    def p2(self, t, m):             # phase B registers
        return (self.n // 16) * (t >> self.sb) + (t & (self.s - 1)) + self.s * m

    def p3(self, t, i):             # phase C registers: i = lc * u + e
        u, e = divmod(i, self.lc)
        return self.lc * (t + self.t * u) + e


def w(n, k):
    return np.exp(-2j * np.pi * k / n)


def dif4(a, b, c, d, w1, w2, w3):
    apc, amc, bpd, bmd = a + c, a - c, b + d, b - d
    return apc + bpd, (amc - 1j * bmd) * w1, (apc - bpd) * w2, (amc + 1j * bmd) * w3


def dit4(a, b, c, d, w1, w2, w3):
    b, c, d = b * np.conj(w1), c * np.conj(w2), d * np.conj(w3)
    apc, amc, bpd, bmd = a + c, a - c, b + d, b - d
    return apc + bpd, amc + 1j * bmd, apc - bpd, amc - 1j * bmd


def forward(x, log2n):
    '''x: natural order; returns the LDS image after the last phase (by position).'''
    pl = Plan(log2n)
    n, T, S = pl.n, pl.t, pl.s
    lds = np.zeros(n, complex)
    regs = np.zeros((T, 16), complex)
    for t in range(T):
        v = [x[pl.p1(t, m)] for m in range(16)]
        for m0 in range(4):
            j = t + T * m0
            v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12] = dif4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12],
                                                          w(n, j), w(n, 2 * j), w(n, 3 * j))
        for g in range(4):
            v[4 * g:4 * g + 4] = dif4(*v[4 * g:4 * g + 4], w(n, 4 * t), w(n, 8 * t), w(n, 12 * t))
        for m in range(16):
            lds[swizzle(pl.p1(t, m), log2n)] = v[m]
    for t in range(T):
        s = t & (S - 1)
        v = [lds[swizzle(pl.p2(t, m), log2n)] for m in range(16)]
        for m0 in range(4):
            j = s + S * m0
            v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12] = dif4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12],
                                                          w(n, 16 * j), w(n, 32 * j), w(n, 48 * j))
        for g in range(4):
            v[4 * g:4 * g + 4] = dif4(*v[4 * g:4 * g + 4], w(n, 64 * s), w(n, 128 * s), w(n, 192 * s))
        regs[t] = v
    if pl.lc == 1:
        out = np.zeros(n, complex)
        for t in range(T):
            for m in range(16):
                out[pl.p2(t, m)] = regs[t][m]
        return out
    for t in range(T):
        for m in range(16):
            lds[swizzle(pl.p2(t, m), log2n)] = regs[t][m]
    out = np.zeros(n, complex)
    for t in range(T):
        v = [lds[swizzle(pl.p3(t, i), log2n)] for i in range(16)]
        v = last_phase_forward(v, pl.lc)
        for i in range(16):
            out[pl.p3(t, i)] = v[i]
    return out


def last_phase_forward(v, lc):
    v = list(v)
    for u in range(16 // lc):
        b = v[lc * u:lc * u + lc]
        if lc == 2:
            b = [b[0] + b[1], b[0] - b[1]]
        elif lc == 4:
            b = list(dif4(*b, 1, 1, 1))
        elif lc == 8:
            for e in range(2):
                b[e], b[e + 2], b[e + 4], b[e + 6] = dif4(b[e], b[e + 2], b[e + 4], b[e + 6],
                                                          w(8, e), w(8, 2 * e), w(8, 3 * e))
            for q in range(4):
                b[2 * q], b[2 * q + 1] = b[2 * q] + b[2 * q + 1], b[2 * q] - b[2 * q + 1]
        elif lc == 16:
            for e in range(4):
                b[e], b[e + 4], b[e + 8], b[e + 12] = dif4(b[e], b[e + 4], b[e + 8], b[e + 12],
                                                           w(16, e), w(16, 2 * e), w(16, 3 * e))
            for g in range(4):
                b[4 * g:4 * g + 4] = dif4(*b[4 * g:4 * g + 4], 1, 1, 1)
        v[lc * u:lc * u + lc] = b
    return v


def last_phase_inverse(v, lc):
    v = list(v)
    for u in range(16 // lc):
        b = v[lc * u:lc * u + lc]
        if lc == 2:
            b = [b[0] + b[1], b[0] - b[1]]
        elif lc == 4:
            b = list(dit4(*b, 1, 1, 1))
        elif lc == 8:
            for q in range(4):
                b[2 * q], b[2 * q + 1] = b[2 * q] + b[2 * q + 1], b[2 * q] - b[2 * q + 1]
            for e in range(2):
                b[e], b[e + 2], b[e + 4], b[e + 6] = dit4(b[e], b[e + 2], b[e + 4], b[e + 6],
                                                          w(8, e), w(8, 2 * e), w(8, 3 * e))
        elif lc == 16:
            for g in range(4):
                b[4 * g:4 * g + 4] = dit4(*b[4 * g:4 * g + 4], 1, 1, 1)
            for e in range(4):
                b[e], b[e + 4], b[e + 8], b[e + 12] = dit4(b[e], b[e + 4], b[e + 8], b[e + 12],
                                                           w(16, e), w(16, 2 * e), w(16, 3 * e))
        v[lc * u:lc * u + lc] = b
    return v


def inverse(y, log2n):
    '''y: by position (digit-reversed spectrum); returns natural order, unscaled.'''
    pl = Plan(log2n)
    n, T, S = pl.n, pl.t, pl.s
    lds = np.zeros(n, complex)
    regs = np.zeros((T, 16), complex)
    if pl.lc > 1:
        for t in range(T):
            v = last_phase_inverse([y[pl.p3(t, i)] for i in range(16)], pl.lc)
            for i in range(16):
                lds[swizzle(pl.p3(t, i), log2n)] = v[i]
        for t in range(T):
            regs[t] = [lds[swizzle(pl.p2(t, m), log2n)] for m in range(16)]
    else:
        for t in range(T):
            regs[t] = [y[pl.p2(t, m)] for m in range(16)]
    for t in range(T):
        s = t & (S - 1)
        v = list(regs[t])
        for g in range(4):
            v[4 * g:4 * g + 4] = dit4(*v[4 * g:4 * g + 4], w(n, 64 * s), w(n, 128 * s), w(n, 192 * s))
        for m0 in range(4):
            j = s + S * m0
            v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12] = dit4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12],
                                                          w(n, 16 * j), w(n, 32 * j), w(n, 48 * j))
        for m in range(16):
            lds[swizzle(pl.p2(t, m), log2n)] = v[m]
    out = np.zeros(n, complex)
    for t in range(T):
        v = [lds[swizzle(pl.p1(t, m), log2n)] for m in range(16)]
        for g in range(4):
            v[4 * g:4 * g + 4] = dit4(*v[4 * g:4 * g + 4], w(n, 4 * t), w(n, 8 * t), w(n, 12 * t))
        for m0 in range(4):
            j = t + T * m0
            v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12] = dit4(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12],
                                                          w(n, j), w(n, 2 * j), w(n, 3 * j))
        for m in range(16):
            out[pl.p1(t, m)] = v[m]
    return out


def conflicts(log2n):
    '''Worst number of distinct addresses per bank pair, per pattern, for reads (32 lanes,
    32 bank pairs) and writes (16 lanes, 16 bank pairs).  Several transforms side by side
    in one wavefront (N < 1024) sit n + 16 elements apart.'''
    pl = Plan(log2n)
    lanes = 64
    report = {}
    for name, fn in (('p1', pl.p1), ('p2', pl.p2), ('p3', pl.p3)):
        if name == 'p3' and pl.lc == 1:
            continue
        worst_r = worst_w = 1
        for wave0 in range(0, max(pl.t, lanes), lanes):
            for m in range(16):
                addr = []
                for lane in range(lanes):
                    gt = wave0 + lane
                    tr, t = divmod(gt, pl.t)
                    addr.append(tr * (pl.n + 16) + swizzle(fn(t, m), log2n))
                for grp, banks in ((32, 32), (16, 16)):
                    for g0 in range(0, lanes, grp):
                        per = {}
                        for a in addr[g0:g0 + grp]:
                            per.setdefault(a % banks, set()).add(a)
                        ways = max(len(v) for v in per.values())
                        if grp == 32:
                            worst_r = max(worst_r, ways)
                        else:
                            worst_w = max(worst_w, ways)
        report[name] = (worst_r, worst_w)
    return report


def main():
    rng = np.random.default_rng(0)
    for log2n in range(8, 13):
        n = 1 << log2n
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        ref = np.fft.fft(x)
        got = forward(x, log2n)
        err_f = max(abs(got[pos_of_freq(k, log2n)] - ref[k]) for k in range(n))
        back = inverse(got, log2n) / n
        err_b = np.abs(back - x).max()
        perm = sorted(swizzle(p, log2n) for p in range(n))
        assert perm == list(range(n)), 'swizzle is not a permutation'
        print(f'N = {n}: forward error {err_f:.2e}, round trip {err_b:.2e}, '
              f'conflicts (read ways, write ways) {conflicts(log2n)}')


if __name__ == '__main__':
    main()
