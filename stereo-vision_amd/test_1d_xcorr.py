'''Driver: cross-correlates two analytic 1D test signals with every 1D
correlator and plots the signals and the outputs.  Counterpart of the
reference's test_1d_xcorr.py; --no_plot is an addition.'''
from argparse import ArgumentParser
from datetime import datetime

import numpy as np

from correlation import cross_correlate_1d, cross_correlate_1d_raw
from correlation_spectral import cross_correlate_1d_spectral

# Adjust these for different tests
STEP_SIZE = np.pi / 2000
TEMPLATE_FUNC = np.cos
SIGNAL_FUNC = lambda x: -np.sin(x)   # noqa: E731


def main():
    parser = ArgumentParser()
    parser.add_argument('--no_plot', action='store_true', default=False)
    args = parser.parse_args()
    end = 4 * np.pi
    x_vec = np.arange(0, end + STEP_SIZE, STEP_SIZE)
    output_x_vec = np.arange(0, 2 * end + STEP_SIZE, STEP_SIZE)
    template, signal = TEMPLATE_FUNC(x_vec), SIGNAL_FUNC(x_vec)

    def numpy_full(t, s):
        return np.correlate((t - np.mean(t)) / (np.std(t) * t.size), (s - np.mean(s)) / np.std(s), mode='full')

    results = []
    for label, title, fn in (('raw custom correlation', 'raw custom correlation', cross_correlate_1d_raw),
                             ('vectorised custom correlation', 'vectorised custom correlation', cross_correlate_1d),
                             ('spectral correlation', 'spectral correlation', cross_correlate_1d_spectral),
                             ('Numpy correlation', 'numpy "full" correlation', numpy_full)):
        start = datetime.now()
        results.append((title, fn(template, signal)))
        print(f'Time elapsed ({label}): {datetime.now() - start}')
    if args.no_plot:
        return results

    from matplotlib import pyplot as plt
    _, axis = plt.subplots(5)
    axis[0].plot(x_vec, signal, label='signal')
    axis[0].plot(x_vec, template, label='template')
    axis[0].legend()
    axis[0].set_title('signal and template')
    for ax, (title, values) in zip(axis[1:], results):
        ax.plot(output_x_vec[:len(values)], values)
        ax.set_title(title)
    plt.show()
    return results


if __name__ == '__main__':
    main()
