'''The N > 1 path on CPU: two gloo ranks shard the windows of a scan, evaluate
their share and all-gather the per-window results.  The device call is
replaced by the CPU oracle here (there is no GPU), which leaves the sharding,
index shifting, padding and gather logic under test.'''
import io
import contextlib
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers  # noqa: F401
from helpers import oracle_scan_grid_rows, oracle_select_local


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
        v._scan_grid_rows = oracle_scan_grid_rows
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


def band_pair():
    import bench
    left, _ = bench.synthetic_pair(15 + 36, 15 + 80, seed=5)
    return left, np.roll(left, 3, axis=1)


def band_worker(rank, size, port, out_dir):
    '''The benchmark's strong-scaling split on CPU: each rank scans its band of
    window rows from a sub-image with a halo of win - 1 rows (the oracle stands in
    for the band kernel), packs int16 disparities + float32 correlations into an
    equal-sized byte buffer and rank 0 gathers them.'''
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import bench
        import sharding
        from oracle import ncc_oracle
        win, md = 15, 6
        left, right = band_pair()
        nwy, nwx = left.shape[0] - win + 1, left.shape[1] - win + 1
        bands = [sharding.shard_range(nwy, r, size) for r in range(size)]
        counts = [(e - b) * nwx for b, e in bands]
        cap = max(counts)
        b, e = bands[rank]
        h = e - b + win - 1
        disp, val = ncc_oracle.band_scan(left[b:b + h], right[b:b + h], win, win, md, mode='fold', threshold=0.5)
        buf = np.zeros(6 * cap, dtype=np.uint8)
        buf[:2 * cap].view(np.int16)[:counts[rank]] = disp.ravel()
        buf[2 * cap:].view(np.float32)[:counts[rank]] = val.ravel()
        gathered = [torch.empty(6 * cap, dtype=torch.uint8) for _ in range(size)] if rank == 0 else None
        dist.gather(torch.from_numpy(buf), gathered, dst=0)
        if rank == 0:
            parts = [bench.unpack_results(g.numpy(), cap, n) for g, n in zip(gathered, counts)]
            np.save(os.path.join(out_dir, 'disp.npy'), np.concatenate([d for d, _ in parts]).reshape(nwy, nwx))
            np.save(os.path.join(out_dir, 'val.npy'), np.concatenate([v for _, v in parts]).reshape(nwy, nwx))
    finally:
        dist.destroy_process_group()


def test_band_rows_with_halo_gathered_equal_whole_scan(tmp_path, oracle):
    import bench
    size = 2
    mp.spawn(band_worker, args=(size, free_port(), str(tmp_path)), nprocs=size, join=True)
    left, right = band_pair()
    disp, val = oracle.band_scan(left, right, 15, 15, 6, mode='fold', threshold=0.5)
    assert np.array_equal(np.load(tmp_path / 'disp.npy'), disp.astype(np.int16))
    assert np.array_equal(np.load(tmp_path / 'val.npy'), val.astype(np.float32))
    assert (disp != 0).any()
    # evaluation counts of the bands add up to the whole pair's
    nwy = left.shape[0] - 15 + 1
    total = bench.count_evals(left.shape[0], left.shape[1], 15, 6)
    assert sum(bench.count_evals(left.shape[0], left.shape[1], 15, 6, rows=e - b)
               for b, e in [(0, nwy // 2), (nwy // 2, nwy)]) == total


class OracleBandScanner:
    '''band_scan.BandScanner's interface on the CPU, the oracle doing the scan: what
    bench.Sharded needs from a scanner (upload, prepare, run, disp, val).'''

    def __init__(self, height, width, win_w, win_h, max_disp):
        import torch
        self.win, self.max_disp = win_w, max_disp
        self.disp = torch.zeros((height - win_h + 1, width - win_w + 1), dtype=torch.int32)
        self.val = torch.zeros((height - win_h + 1, width - win_w + 1), dtype=torch.float64)

    def upload(self, left, right):
        self.left, self.right = np.ascontiguousarray(left), np.ascontiguousarray(right)

    def prepare(self):
        pass

    def run(self, threshold, prepared=False):
        import torch
        from oracle import ncc_oracle
        disp, val = ncc_oracle.band_scan(self.left, self.right, self.win, self.win, self.max_disp, mode='fold',
                                         threshold=threshold)
        self.disp.copy_(torch.from_numpy(disp))
        self.val.copy_(torch.from_numpy(val))


def bench_worker(rank, size, port, out_dir):
    '''bench.py's own step loop (Sharded.step with its two wire buffers taking turns and
    the asynchronous gather, drain, timed's barriers and max-over-ranks, last_gathered,
    slowest_rank) over gloo, the oracle standing in for the band kernel.'''
    import types
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import bench
        import sharding
        bench.WIN, bench.MAX_DISP = 15, 6
        left, right = band_pair()
        nwy, nwx = left.shape[0] - 15 + 1, left.shape[1] - 15 + 1
        bands = [sharding.shard_range(nwy, r, size) for r in range(size)]
        counts = [(e - b) * nwx for b, e in bands]
        job = bench.Sharded(types.SimpleNamespace(BandScanner=OracleBandScanner), torch, left, right, bands[rank],
                            size, rank, 'cpu', max(counts), device='cpu')
        elapsed = bench.timed(torch, dist, size, 'cpu', job, steps=3, warmup=1, sync=lambda: None)
        assert elapsed > 0 and job.issued == 4 and job.in_flight == [None, None]
        worst = bench.slowest_rank(torch, dist, size, 'cpu', 10.0 + rank, 5.0 - rank, 100 + rank)
        assert worst == (10.0 + size - 1, 5.0 - (size - 1), 100 + size - 1)
        if rank == 0:
            parts = job.last_gathered(counts)
            np.save(os.path.join(out_dir, 'disp.npy'), np.concatenate([d for d, _ in parts]).reshape(nwy, nwx))
            np.save(os.path.join(out_dir, 'val.npy'), np.concatenate([v for _, v in parts]).reshape(nwy, nwx))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('size', [2, 3])
def test_bench_step_loop_gathers_the_whole_map(tmp_path, oracle, size):
    '''37 window rows over 2 and over 3 ranks (uneven shares): after bench.py's timed loop
    the last gathered map equals one process scanning the whole pair, bit for bit.'''
    mp.spawn(bench_worker, args=(size, free_port(), str(tmp_path)), nprocs=size, join=True)
    left, right = band_pair()
    assert (left.shape[0] - 15 + 1) % size != 0
    disp, val = oracle.band_scan(left, right, 15, 15, 6, mode='fold', threshold=0.5)
    assert np.array_equal(np.load(tmp_path / 'disp.npy'), disp.astype(np.int16))
    assert np.array_equal(np.load(tmp_path / 'val.npy'), val.astype(np.float32))
