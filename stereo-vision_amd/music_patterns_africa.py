'''Driver: finds occurrences of short audio snippets in a long recording with the
circular cross-correlation.  Counterpart of the reference's
music_patterns_africa.py; --audio_dir, --synthetic and --no_plot are additions
(the reference's .wav files are not distributed with it).'''
from argparse import ArgumentParser
from datetime import datetime

import numpy as np

from correlation_spectral import cross_correlate_1d_spectral

SNIPPETS = (('first_verse', 'Correlation with first verse', 'First verse'),
            ('drum_snare', 'Correlation with snare drum', 'Snare drum'),
            ('hurry', 'Correlation with "Hurry boy"', '"Hurry boy"'),
            ('rains', 'Correlation with "I bless the rains down in Africa"', '"Rains"'))


def left_channel(samples):
    '''Stereo snippets contribute their left channel only.'''
    samples = np.asarray(samples)
    return samples[:, 0] if samples.ndim == 2 else samples


def synthetic_recording(seconds=20, rate=44_100, seed=0):
    '''A noise recording and four snippets cut out of it at known positions.'''
    rng = np.random.default_rng(seed)
    signal = np.rint(3000 * rng.standard_normal(seconds * rate)).astype(np.int16)
    starts = {'first_verse': 2 * rate, 'drum_snare': 5 * rate + 123, 'hurry': 9 * rate + 7, 'rains': 14 * rate}
    lengths = {'first_verse': rate, 'drum_snare': rate // 10, 'hurry': rate // 2, 'rains': 2 * rate}
    snippets = {name: signal[starts[name]:starts[name] + lengths[name]].copy() for name in starts}
    return signal, snippets, starts


def main():
    parser = ArgumentParser()
    parser.add_argument('--audio_dir', default='audio')
    parser.add_argument('--synthetic', action='store_true', default=False)
    parser.add_argument('--no_plot', action='store_true', default=False)
    args = parser.parse_args()
    if args.synthetic:
        signal, snippets, _ = synthetic_recording()
    else:
        from scipy.io import wavfile
        _, signal = wavfile.read(f'{args.audio_dir}/africa.wav')
        snippets = {name: left_channel(wavfile.read(f'{args.audio_dir}/africa_{name}.wav')[1])
                    for name, _, _ in SNIPPETS}
    correlations = []
    for name, _, label in SNIPPETS:
        start = datetime.now()
        correlations.append(cross_correlate_1d_spectral(signal, snippets[name]))
        print(f'{label} correlation complete; time elapsed: {datetime.now() - start}')
        print(f'{label} maximum at: {int(np.argmax(correlations[-1]))}')
    if args.no_plot:
        return
    from matplotlib import pyplot as plt
    _, axes = plt.subplots(2, 2)
    for ax, correlation, (_, title, _) in zip(axes.ravel(), correlations, SNIPPETS):
        ax.plot(correlation)
        ax.set_title(title)
    plt.show()


if __name__ == '__main__':
    main()
