'''Driver: cross-correlates a 2D template with a region, prints the position of the
maximum and (optionally) plots template, region and surface.  Counterpart of the
reference's test_2d_xcorr.py with the same flags (--spectral --template --region
--step); --images_dir and --no_plot are additions.'''
import argparse
import datetime

from correlation import cross_correlate_2d
from correlation_spectral import cross_correlate_2d_spectral
from utils import greyscale, greyscale_with_nan, max_pos


def options(argv=None):
    cli = argparse.ArgumentParser(description=__doc__)
    cli.add_argument('--spectral', action='store_true', help='circular correlation of the whole images')
    cli.add_argument('--template', default='rm_template.png')
    cli.add_argument('--region', default='rm_region.png')
    cli.add_argument('--step', type=int, default=5, help='stride of the spatial correlation')
    cli.add_argument('--images_dir', default='images')
    cli.add_argument('--no_plot', action='store_true')
    return cli.parse_args(argv)


def correlate(opts):
    '''Returns (template, region, surface, step).  The spatial route leaves the
    template's transparent pixels out of the correlation (they load as NaN); the
    spectral route correlates plain grey levels at every position.'''
    region_path = f'{opts.images_dir}/{opts.region}'
    template_path = f'{opts.images_dir}/{opts.template}'
    region = greyscale(region_path)
    began = datetime.datetime.now()
    if opts.spectral:
        template, step = greyscale(template_path), 1
        surface = cross_correlate_2d_spectral(template, region)
    else:
        template, step = greyscale_with_nan(template_path), opts.step
        surface = cross_correlate_2d(template, region, step_x=step, step_y=step)
    print(f'Time elapsed : {datetime.datetime.now() - began}')
    return template, region, surface, step


def show(template, region, surface, corner):
    from matplotlib import pyplot
    from matplotlib.patches import Rectangle
    figure = pyplot.figure(figsize=(3, 1))
    for slot, image in enumerate((template, region, surface), start=1):
        axes = figure.add_subplot(3, 1, slot)
        axes.imshow(image)
        if slot == 2:       # the template's footprint at the best match
            axes.add_patch(Rectangle(corner, template.shape[1], template.shape[0], linewidth=2, edgecolor='r',
                                     facecolor='none'))
    pyplot.show()


def main():
    opts = options()
    template, region, surface, step = correlate(opts)
    best = max_pos(surface, step, step)
    print(f'Maximum cross-correlation at: {best}')
    if not opts.no_plot:
        show(template, region, surface, best)


if __name__ == '__main__':
    main()
