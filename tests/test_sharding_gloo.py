'''The N > 1 path on CPU: two gloo ranks shard the windows of a scan, evaluate
their share and all-gather the per-window results.  The device call is
replaced by the CPU oracle here (there is no GPU), which leaves the sharding,
index shifting, padding and gather logic under test.'''
import io
import contextlib
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import helpers  # noqa: F401
from helpers import oracle_select_local


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, size, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import sharding
        import sv_image_comparison as v
        import _native as nv
        assert sharding.world() == (rank, size)
        # uneven shares: 7 rows over 2 ranks
        b, e = sharding.shard_range(7)
        full = np.arange(21, dtype=np.float64).reshape(7, 3)
        got = sharding.all_gather_concat(full[b:e], 7)
        assert np.array_equal(got, full)

        v._select_local = oracle_select_local
        nv.to_device = lambda a: np.ascontiguousarray(a)     # "resident" images stay numpy here
        rng = np.random.default_rng(7)
        left = rng.random((40, 66))
        right = np.roll(left, 3, axis=1)
        cfg = [{'window_width': 12, 'window_height': 12, 'scheme': [7, 1], 'scheme_shift_size': [1, 0],
                'window_overlap': 0.5, 'correlation_threshold': 0.4},
               {'factor': 2, 'correlation_threshold': 0}]
        with contextlib.redirect_stdout(io.StringIO()):
            res = v.sequence_scan(left, right, cfg)
        np.save(os.path.join(out_dir, f'dp_rank{rank}.npy'),
                np.array([[w['dp_x'], w['dp_y']] for st in res for w in st.values()]))
    finally:
        dist.destroy_process_group()


def test_two_rank_scan_matches_single_process(tmp_path, oracle):
    size = 2
    mp.spawn(worker, args=(size, free_port(), str(tmp_path)), nprocs=size, join=True)
    r0 = np.load(tmp_path / 'dp_rank0.npy')
    r1 = np.load(tmp_path / 'dp_rank1.npy')
    assert np.array_equal(r0, r1)
    rng = np.random.default_rng(7)
    left = rng.random((40, 66))
    right = np.roll(left, 3, axis=1)
    cfg = [{'window_width': 12, 'window_height': 12, 'scheme': [7, 1], 'scheme_shift_size': [1, 0],
            'window_overlap': 0.5, 'correlation_threshold': 0.4},
           {'factor': 2, 'correlation_threshold': 0}]
    want = oracle.sequence_scan(left, right, cfg)
    want = np.array([[w['dp_x'], w['dp_y']] for st in want for w in st.values()])
    assert np.array_equal(r0, want)
    assert (want != 0).any()


def test_shard_range_covers_everything():
    import sharding
    for count in (0, 1, 5, 16, 4066):
        for size in (1, 2, 3, 8):
            spans = [sharding.shard_range(count, r, size) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
