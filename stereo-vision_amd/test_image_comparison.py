'''Driver: runs every sequence of a scan config on a left/right image pair,
accumulates the per-window displacements into full-resolution grids (on the
device, depth_grids.DepthGrids), averages them by the number of contributing
windows and optionally saves / plots them.
Counterpart of the reference's test_image_comparison.py with the same flags;
--image_dir / --config_dir / --data_dir are additions.'''
import json
from argparse import ArgumentParser
from datetime import datetime

import numpy as np

from depth_grids import DepthGrids
from sv_image_comparison import sequence_scan, plot_sequence_output
from utils import downsample, greyscale


def calc_max_shift_magnitude(config):
    spans = []
    for axis, key in ((0, 'window_width'), (1, 'window_height')):
        unit = config['scheme_shift_size'][axis] or config[key]
        spans.append(unit * (config['scheme'][axis] - 1) / 2)
    return np.sqrt(spans[0] ** 2 + spans[1] ** 2)


def main():
    parser = ArgumentParser()
    parser.add_argument('--images', default='test_2_noise')
    parser.add_argument('--ds_factor', type=int, default=1)
    parser.add_argument('--config', default='scan_config.json')
    parser.add_argument('--depth_output', default=None)
    parser.add_argument('--depth_input', default=None)
    parser.add_argument('--shift_plot_type', default='boxes')
    parser.add_argument('--sequence_plots', action='store_true', default=False)
    parser.add_argument('--hide_depth', action='store_true', default=False)
    parser.add_argument('--image_dir', default='images-p2-uncal')
    parser.add_argument('--config_dir', default='configs')
    parser.add_argument('--data_dir', default='depth-data')
    args = parser.parse_args()

    with open(f'{args.config_dir}/{args.config}') as f:
        config = json.load(f)

    if args.depth_input is None:
        images = []
        for side in ('left', 'right'):
            image = greyscale(f'{args.image_dir}/{side}_{args.images}.tiff')
            images.append(downsample(image, args.ds_factor))
        left_image, right_image = images
        totals = DepthGrids(left_image.shape)
        for i, seq_config in enumerate(config):
            start = datetime.now()
            seq_results = sequence_scan(left_image, right_image, seq_config)
            print(f'Time elapsed for sequence {i + 1} scan: {datetime.now() - start}')
            if args.sequence_plots:
                plot_sequence_output(left_image, right_image, seq_results,
                                     max_shift_magnitude=calc_max_shift_magnitude(seq_config[0]),
                                     shift_plot_type=args.shift_plot_type)
            totals.add_sequence(seq_results)
        # average by the number of windows that contributed to each pixel
        total_depth_grid, total_dp_x_grid, total_dp_y_grid, total_contributions_grid = totals.average()
        if args.depth_output is not None:
            with open(f'{args.data_dir}/{args.depth_output}', 'w') as f:
                json.dump({
                    'depth_grid': total_depth_grid.tolist(),
                    'dp_x_grid': total_dp_x_grid.tolist(),
                    'dp_y_grid': total_dp_y_grid.tolist(),
                    'contributions_grid': total_contributions_grid.tolist(),
                }, f)
            print(f'Data saved to file {args.depth_output}')
        else:
            print('No output file specified.')
    else:
        with open(f'{args.data_dir}/{args.depth_input}') as f:
            saved = json.load(f)
        total_depth_grid = np.array(saved['depth_grid'])
        total_contributions_grid = np.array(saved['contributions_grid'])
        print(f'Data read from file {args.depth_input}')

    if not args.hide_depth:
        show_depth(total_depth_grid, total_contributions_grid)


def show_depth(depth, contributions):
    '''The averaged displacement magnitude as a surface (image orientation: row 0 at
    the back) and, side by side, as an image next to the contribution counts.'''
    from matplotlib import cm
    from matplotlib import pyplot as plt
    rows, cols = depth.shape
    surface = plt.figure().add_subplot(projection='3d')
    surface.plot_surface(*np.meshgrid(np.arange(cols), np.arange(rows)), depth, cmap=cm.coolwarm)
    surface.set(xlabel='x', ylabel='y', ylim=(rows, 0))
    _, panels = plt.subplots(1, 2)
    for panel, grid in zip(panels, (depth, contributions)):
        panel.imshow(grid)
    plt.show()


if __name__ == '__main__':
    main()
