'''Driver: times the list-based and array-based 1D correlators and their
"Numba" twins on sine signals.  Counterpart of the reference's numba_tests.py
(same signal sizes and output lines); here all four run on the device.'''
from datetime import datetime

import numpy as np

from correlation import cross_correlate_1d, cross_correlate_1d_raw
from correlation_numba import cross_correlate_1d_numba, cross_correlate_1d_raw_numba

RAW_STEP_SIZE = np.pi / 1000
VEC_STEP_SIZE = np.pi / 20000


def main():
    for step, rows in ((RAW_STEP_SIZE, (('Raw python, without Numba JIT', cross_correlate_1d_raw),
                                        ('Raw python, with Numba JIT', cross_correlate_1d_raw_numba))),
                       (VEC_STEP_SIZE, (('Vectorisation, without Numba JIT', cross_correlate_1d),
                                        ('Vectorisation, with Numba JIT', cross_correlate_1d_numba)))):
        x_vec = np.arange(0, 4 * np.pi + step, step)
        template, signal = np.sin(x_vec), -np.sin(x_vec)
        for label, fn in rows:
            start = datetime.now()
            _ = fn(template, signal)
            print(f'{label}; time elapsed: {datetime.now() - start}')


if __name__ == '__main__':
    main()
