'''Driver: cross-correlates two sensor signals and derives the sample shift, time
shift and distance.  Counterpart of the reference's signal_offset.py (--spectral);
--data_dir, --synthetic and --no_plot are additions (the reference's data files
are not distributed with it).'''
from argparse import ArgumentParser
from datetime import datetime

import numpy as np

from correlation import cross_correlate_1d
from correlation_spectral import cross_correlate_1d_spectral

SIGNAL_FREQ = 44_000
SIGNAL_SPEED = 333


def load_signals(args):
    if args.synthetic:
        rng = np.random.default_rng(0)
        signal_1 = rng.standard_normal(SIGNAL_FREQ)
        signal_2 = np.roll(signal_1, 137) + 0.1 * rng.standard_normal(SIGNAL_FREQ)
        return signal_1, signal_2
    # one value per line below a one-line header, as pandas.read_csv(...).squeeze() reads it
    return (np.loadtxt(f'{args.data_dir}/sensor1Data.txt', skiprows=1),
            np.loadtxt(f'{args.data_dir}/sensor2Data.txt', skiprows=1))


def main():
    parser = ArgumentParser()
    parser.add_argument('--spectral', action='store_true', default=False)
    parser.add_argument('--data_dir', default='1d-data')
    parser.add_argument('--synthetic', action='store_true', default=False)
    parser.add_argument('--no_plot', action='store_true', default=False)
    args = parser.parse_args()
    signal_1, signal_2 = load_signals(args)

    start = datetime.now()
    if args.spectral:
        correlation = cross_correlate_1d_spectral(signal_1, signal_2)
    else:
        correlation = cross_correlate_1d(signal_1, signal_2)
    print(f'Time elapsed : {datetime.now() - start}')

    peak = np.argmax(correlation)
    shift_num = np.abs(peak - len(signal_1))
    shift_time = shift_num / SIGNAL_FREQ
    print(f'Max at: {peak}')
    print(f'Max value: {correlation[peak]}')
    print(f'Number of signals shifted: {shift_num}')
    print(f'Signal shift time: {shift_time}')
    print(f'Signal shift distance: {shift_time * SIGNAL_SPEED}')
    if args.no_plot:
        return

    from matplotlib import pyplot as plt
    _, axis = plt.subplots(3)
    for ax, data, title in zip(axis, (signal_1, signal_2, correlation),
                               ('Signal 1', 'Signal 2', 'Cross-correlation')):
        ax.plot(data)
        ax.set_title(title)
    plt.show()


if __name__ == '__main__':
    main()
