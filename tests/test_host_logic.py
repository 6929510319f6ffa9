'''Host side of the stereo scan on the CPU: window / region enumeration, batching
into pair descriptors, stage hand-over and result assembly of
sv_image_comparison.sequence_scan, with the device call replaced by the CPU
oracle (same contract).  Compared against the reference's own outputs (golden
vectors), so this pins everything but the kernels.'''
import contextlib
import io
import json

import numpy as np
import pytest

from helpers import golden, oracle_scan_grid_rows, oracle_select_local


@pytest.fixture()
def host_only(monkeypatch, oracle):
    import _native as nv
    import sv_image_comparison as v
    monkeypatch.setattr(v, '_select_local', oracle_select_local)
    monkeypatch.setattr(v, '_scan_grid_rows', oracle_scan_grid_rows)
    monkeypatch.setattr(nv, 'to_device', lambda a: np.ascontiguousarray(a))
    return v


def run_and_compare(v, g, left, right):
    cfg = json.loads(str(g['config']))
    for s in range(int(g['n_seq'])):
        with contextlib.redirect_stdout(io.StringIO()):
            res = v.sequence_scan(left, right, cfg[s])
        for st, windows in enumerate(res):
            want = g[f'seq{s}_stage{st}_windows']
            centres = g[f'seq{s}_stage{st}_centres']
            sizes = g[f'seq{s}_stage{st}_sizes']
            assert list(windows) == [int(i) for i in want[:, 0]]
            for i, w in enumerate(windows.values()):
                row = want[i]
                assert (w['row'], w['column']) == (int(row[1]), int(row[2]))
                assert w['centre'] == (row[3], row[4])
                assert (w['dp_x'], w['dp_y']) == (row[5], row[6]), (s, st, i)
                assert len(w['target_regions']) == int(row[9])
                assert [tuple(c) for c in w['stage_centres']] == [tuple(c) for c in centres[i, :int(row[7])]]
                assert [tuple(z) for z in w['stage_sizes']] == [tuple(z) for z in sizes[i, :int(row[8])]]


@pytest.mark.parametrize('name', ['g6_cone_scan_config_portal', 'g6_cone_scan_config', 'g9_cone_multipass'])
def test_cone_scans_host_logic(host_only, name):
    im = golden('g6_cone_images')
    run_and_compare(host_only, golden(name), im['left'] / 255.0, im['right'] / 255.0)


def test_window_dicts_match_reference_layout(host_only):
    v = host_only
    rng = np.random.default_rng(0)
    left, right = rng.random((30, 50)), rng.random((30, 50))
    windows = v.whole_image_search_regions(left, right, 12, 10, scheme=(3, 3), scheme_shift_size=(4, 0),
                                           window_overlap=0.5)
    assert list(windows) == list(range(len(windows)))
    w = windows[0]
    assert set(w) == {'row', 'column', 'centre', 'size', 'image_slice', 'target_regions'}
    assert w['centre'] == (6.0, 5.0) and w['size'] == (12, 10)
    assert w['image_slice'].base is left or w['image_slice'].base is left.base
    # centre first, then the grid row by row without the centre
    centres = [r['centre'] for r in w['target_regions']]
    assert centres[0] == (6.0, 5.0)
    assert centres[1:] == [c for c in v.region_pairs(6.0, 12, 5.0, 10, (3, 3), (4, 0))[1:]
                           if v.get_window_boundaries(c[0], c[1], 12, 10, 50, 30) is not None]
    for r in w['target_regions']:
        b = v.get_window_boundaries(r['centre'][0], r['centre'][1], 12, 10, 50, 30)
        assert r['size'] == (b[1] - b[0], b[3] - b[2])
        assert np.shares_memory(r['image_slice'], right)
        assert r['image_slice'].shape == (b[3] - b[2], b[1] - b[0])
    # bottom-right windows are truncated by slicing
    last = windows[len(windows) - 1]
    assert last['image_slice'].shape[0] <= 10 and last['image_slice'].shape[1] <= 12


def test_grid_tables_enumerate_the_reference_pairs(host_only):
    '''The per-axis tables handed to svamd_scan_grid, expanded the way the device
    expands them, give exactly the (template, region) boxes, in exactly the order, of
    the reference's per-window enumeration (region_pairs + get_window_boundaries,
    sv_image_comparison.py:119-191) - including candidates that leave the image, empty
    boxes at the border, truncated windows and schemes whose grid misses the centre.'''
    from helpers import enumerate_grid_pairs
    v = host_only
    rng = np.random.default_rng(11)
    cases = 0
    for _ in range(60):
        h, w = int(rng.integers(20, 48)), int(rng.integers(20, 60))
        ww, wh = int(rng.integers(4, 14)), int(rng.integers(4, 14))
        scheme = (int(rng.choice([1, 3, 5, 9])), int(rng.choice([1, 3, 5])))
        shift = (int(rng.choice([0, 1, 3, 25])), int(rng.choice([0, 1, 2, 30])))
        overlap = float(rng.choice([0, 0.3, 0.5, 0.75]))
        if int(ww * (1 - overlap)) <= 0 or int(wh * (1 - overlap)) <= 0:
            continue
        left, right = rng.random((h, w)), rng.random((h, w))
        grid = v._Stage0Grid(left, right, ww, wh, scheme, shift, overlap)
        tables = grid.tables()
        begin, end = sorted(int(i) for i in rng.integers(0, grid.nrow + 1, size=2))
        desc, is_centre, pair_slot, win_off = enumerate_grid_pairs(
            tables, grid.ncol, scheme[0], grid.nrow, scheme[1], begin, end, w, w)
        p = 0
        for k, window_id in enumerate(range(begin * grid.ncol, end * grid.ncol)):
            row, column = divmod(window_id, grid.ncol)
            x, y = int(grid.ax.starts[column]), int(grid.ay.starts[row])
            centre = (x + ww / 2, y + wh / 2)
            assert win_off[k] == p
            for cx, cy in v.region_pairs(centre[0], ww, centre[1], wh, scheme, shift):
                box = v.get_window_boundaries(cx, cy, ww, wh, w, h)
                if box is None:
                    continue
                x0, x1, y0, y1 = box
                assert tuple(desc[p]) == (y * w + x, w, min(y + wh, h) - y, min(x + ww, w) - x,
                                          y0 * w + x0, w, y1 - y0, x1 - x0)
                assert bool(is_centre[p]) == ((cx, cy) == centre)
                slot = int(pair_slot[p])
                got = centre if slot == 0 else (grid.ax.grid[column, (slot - 1) % scheme[0]],
                                                grid.ay.grid[row, (slot - 1) // scheme[0]])
                assert got == (cx, cy)
                p += 1
            assert grid.pair_count(row, column) == p - win_off[k]
        assert win_off[-1] == p == int(grid.row_first[end] - grid.row_first[begin])
        cases += 1
    assert cases > 30


def test_first_stage_windows_are_built_on_demand(host_only):
    '''sequence_scan's first-stage result behaves as the dict it stands for, builds a
    window when it is read and keeps it.'''
    v = host_only
    rng = np.random.default_rng(3)
    left = rng.random((30, 44))
    right = np.roll(left, 2, axis=1)
    cfg = [{'window_width': 10, 'window_height': 10, 'scheme': [5, 1], 'scheme_shift_size': [1, 0],
            'window_overlap': 0, 'correlation_threshold': 0.5}]
    with contextlib.redirect_stdout(io.StringIO()):
        windows = v.sequence_scan(left, right, cfg)[0]
    assert len(windows) == 15 and list(windows) == list(range(15)) and 3 in windows and 15 not in windows
    assert not windows._built
    w = windows[5]
    assert windows[5] is w and list(windows._built) == [5]
    assert list(w) == ['row', 'column', 'centre', 'size', 'image_slice', 'target_regions', 'stage_centres',
                       'stage_sizes', 'dp_x', 'dp_y']
    assert (w['row'], w['column'], w['centre'], w['size']) == (1, 0, (5.0, 15.0), (10, 10))
    assert w['stage_centres'] == [w['centre']] and w['stage_sizes'] == [(10, 10)]
    assert type(w['dp_x']) is float and np.shares_memory(w['image_slice'], left)
    moved = [i for i, info in windows.items() if info['dp_x'] or info['dp_y']]
    assert moved == windows.moved() and moved
    assert sum(windows[i]['dp_x'] == 2.0 and windows[i]['dp_y'] == 0.0 for i in moved) >= 9
    with pytest.raises(KeyError):
        windows[15]
    assert windows.get(99) is None and dict(windows) == {i: windows[i] for i in range(15)}


def test_get_window_boundaries_edges(host_only):
    v = host_only
    assert v.get_window_boundaries(-6, 5, 12, 10, 50, 30) is None          # x + w/2 <= 0
    assert v.get_window_boundaries(56.5, 5, 12, 10, 50, 30) is None        # x - w/2 > W
    assert v.get_window_boundaries(56, 5, 12, 10, 50, 30) == (50, 50, 0, 10)   # starts exactly at W: empty
    assert v.get_window_boundaries(3.5, 27.5, 12, 10, 50, 30) == (0, 9, 22, 30)
    assert v.region_partition_pairs(10, 8, 6, 4, 2) == [(8.0, 5.0), (12.0, 5.0), (8.0, 7.0), (12.0, 7.0)]


def test_zero_stride_is_rejected(host_only):
    with pytest.raises(ValueError, match='stride of zero'):
        host_only.whole_image_search_regions(np.zeros((40, 40)), np.zeros((40, 40)), 31, 31, (3, 1), (1, 0),
                                             30 / 31)


def test_resident_image_is_a_read_only_array_with_a_twin():
    '''What the loaders return: an ndarray for every numpy purpose, read-only, whose
    device twin does not follow slices, copies or arithmetic.'''
    import utils
    host = np.arange(12.0).reshape(3, 4)
    twin = object()                       # stands for the device tensor
    image = utils.ResidentImage(host, twin)
    assert isinstance(image, np.ndarray) and image.shape == (3, 4) and image.resident is twin
    with pytest.raises(ValueError):
        image[0, 0] = 5.0
    assert image[1:].resident is None and (image * 2).resident is None
    editable = np.array(image)
    editable[0, 0] = 5.0
    assert type(editable) is np.ndarray and host[0, 0] == 0.0
    assert np.array_equal(np.asarray(image), host) and float(image.sum()) == 66.0
    # making it writable gives up the twin first
    image.setflags(write=True)
    assert image.resident is None
    image[0, 0] = 5.0
    assert image[0, 0] == 5.0
    # the flag flipped directly (numpy does not route that through setflags): the twin is
    # still attached but no longer used, the edited pixels are uploaded
    uploads = []
    import types
    twin = types.SimpleNamespace(shape=(3, 4))
    image = utils.ResidentImage(np.arange(12.0).reshape(3, 4), twin)
    assert utils.resident_float64(image) is twin
    image.flags.writeable = True
    image[0, 0] = 7.0
    import _native as nv
    real = nv.to_device
    nv.to_device = lambda a: uploads.append(np.array(a)) or 'uploaded'
    try:
        assert utils.resident_float64(image) == 'uploaded' and uploads[0][0, 0] == 7.0
    finally:
        nv.to_device = real


def test_band_scan_input_levels():
    '''8-bit levels are taken as uint8, as value / 255 and as value * (1 / 255); nothing
    else is (the band kernel's exactness rests on integer pixels).'''
    import band_scan
    levels = np.arange(256, dtype=np.uint8).reshape(16, 16)
    for image in (levels, levels / 255.0, levels * (1.0 / 255), (levels / 255.0).astype(np.float32)):
        assert np.array_equal(band_scan.as_uint8_image(image), levels)
    for bad in (levels / 254.0, levels.astype(np.int32), levels / 255.0 + 1e-9):
        with pytest.raises(TypeError):
            band_scan.as_uint8_image(bad)


def test_dot_grouping_matches_scipy_labelling():
    '''find_peaks groups the surviving pixels itself: same boxes, in the same order, as
    scipy.ndimage.label + find_objects (what the reference calls,
    sv_calibration.py:131-132); and the dot template is the reference's.'''
    from scipy import ndimage
    import sv_calibration as sc
    rng = np.random.default_rng(0)
    for density in (0.02, 0.15, 0.5):
        for _ in range(30):
            m = rng.random((30, 41)) < density
            rows, cols = np.nonzero(m)
            lab, _n = ndimage.label(m)
            want = [(s[0].start, s[0].stop - 1, s[1].start, s[1].stop - 1) for s in ndimage.find_objects(lab)]
            assert sc.group_boxes(rows, cols) == want
    assert sc.group_boxes(np.array([], dtype=int), np.array([], dtype=int)) == []
    xg, yg = np.meshgrid(np.linspace(-5, 5, 101), np.linspace(-5, 5, 101))
    assert np.array_equal(np.exp(-(np.power(xg, 2) + np.power(yg, 2))), sc.dot_template())

