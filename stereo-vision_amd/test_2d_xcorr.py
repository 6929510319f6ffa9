'''Driver: cross-correlates a 2D template with a region, prints the position of the
maximum and (optionally) plots template, region and surface.  Counterpart of the
reference's test_2d_xcorr.py with the same flags (--spectral --template --region
--step); --images_dir and --no_plot are additions.'''
from argparse import ArgumentParser
from datetime import datetime

import numpy as np

from correlation import cross_correlate_2d
from correlation_spectral import cross_correlate_2d_spectral
from utils import greyscale, greyscale_with_nan, max_pos


def main():
    parser = ArgumentParser()
    parser.add_argument('--spectral', action='store_true', default=False)
    parser.add_argument('--template', default='rm_template.png')
    parser.add_argument('--region', default='rm_region.png')
    parser.add_argument('--step', type=int, default=5)
    parser.add_argument('--images_dir', default='images')
    parser.add_argument('--no_plot', action='store_true', default=False)
    args = parser.parse_args()

    region = greyscale(f'{args.images_dir}/{args.region}')
    start = datetime.now()
    if args.spectral:
        template = greyscale(f'{args.images_dir}/{args.template}')
        step = 1
        region_cor = cross_correlate_2d_spectral(template, region)
    else:
        # transparent template pixels become NaN and are left out of the correlation
        template = greyscale_with_nan(f'{args.images_dir}/{args.template}')
        step = args.step
        region_cor = cross_correlate_2d(template, region, step_x=step, step_y=step)
    print(f'Time elapsed : {datetime.now() - start}')

    region_cor_max = max_pos(region_cor, step, step)
    print(f'Maximum cross-correlation at: {region_cor_max}')
    if args.no_plot:
        return

    from matplotlib import pyplot as plt
    from matplotlib.patches import Rectangle
    figure = plt.figure(figsize=(3, 1))
    figure.add_subplot(3, 1, 1)
    plt.imshow(template)
    figure.add_subplot(3, 1, 2)
    plt.gca().add_patch(Rectangle(region_cor_max, *template.T.shape, linewidth=2, edgecolor='r',
                                  facecolor='none'))
    plt.imshow(region)
    figure.add_subplot(3, 1, 3)
    plt.imshow(region_cor)
    plt.show()


if __name__ == '__main__':
    main()
