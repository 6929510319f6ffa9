#!/usr/bin/env python3
'''Benchmark of the dense band scan (the headline workload): a synthetic
4096 x 4096 uint8 stereo pair, a 31 x 31 window at every pixel, 257 horizontal
candidate offsets (-128..+128), image_scan's selection rule fused in.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed over RCCL when N > 1).  The job is ONE
pair per step, sharded over the ranks by bands of window rows (strong scaling):
rank r holds the image rows of its band plus a halo of window_height - 1 rows,
builds the summed-area tables and window statistics of that band only (K1), runs
the band kernel on it (K2) and sends its per-window results (int16 disparity +
float32 correlation, 6 bytes per window) to rank 0.  No other communication.  The
gather of step i is issued asynchronously and travels while step i + 1 computes
(two result buffers take turns); every gather has completed before the clock
stops.  After the timed region rank 0 scans the whole pair alone once and checks
that the gathered map equals it.  For N > 1 a second, untimed-by-the-contract
measurement gives every rank a pair of its own (weak scaling) and is reported as
`scaling_weak_value`.  Rank 0 prints one JSON line.

metric: NCC window evaluations per second, an evaluation being one (window,
offset) score = 2 * 31 * 31 algorithmic float32 flops of cross term.
roofline: the band kernel alone, timed with HIP events on its stream on every rank
(the slowest rank's figures are reported), against the FP32 vector-ALU peak.
cpu_baseline (N = 1 only, measured before the GPU is touched): the C oracle (OpenMP,
every core granted) on a crop, one-thread numpy rows, and the vectorised numpy form
on every granted core through multiprocessing.
'''
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
sys.path.insert(0, ROOT)

WIN = 31
MAX_DISP = 128
THRESHOLD = 0.5
FP32_VALU_PEAK_TFLOPS = 157.3     # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz


def synthetic_pair(height, width, seed):
    '''Textured left image and a right image displaced by a piece-wise constant
    disparity (256-px tiles, |d| <= 100) plus noise, both uint8.'''
    rng = np.random.default_rng(seed)
    pad = 128
    base = rng.integers(0, 256, size=(height, width + 2 * pad), dtype=np.uint8).astype(np.float32)
    # cheap separable smoothing so neighbouring offsets are correlated
    base = (np.roll(base, 1, 1) + 2 * base + np.roll(base, -1, 1)) / 4
    base = (np.roll(base, 1, 0) + 2 * base + np.roll(base, -1, 0)) / 4
    # stretch contrast back to the 8-bit range
    base = np.clip((base - 127.5) * 2.5 + 127.5, 0, 255)
    left = np.rint(base[:, pad:pad + width]).astype(np.uint8)
    right = np.empty_like(left)
    for ty in range(0, height, 256):
        for tx in range(0, width, 256):
            d = int(rng.integers(-100, 101))
            y1, x1 = min(ty + 256, height), min(tx + 256, width)
            right[ty:y1, tx:x1] = np.rint(base[ty:y1, pad + tx - d:pad + x1 - d]).astype(np.uint8)
    noise = rng.integers(-2, 3, size=right.shape)
    right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return left, right


def measured_traffic():
    '''HBM bytes per launch of the band kernel from the last rocprofv3 counter run
    (collected separately: counters cannot be read from inside this process).'''
    path = os.path.join(ROOT, 'profiles', 'band_traffic.json')
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None


def count_evals(height, width, win, max_disp, rows=None):
    '''Number of (window, offset) pairs whose candidate box lies inside the image
    (of `rows` window rows, default all).'''
    nwx, nwy = width - win + 1, (height - win + 1 if rows is None else rows)
    x = np.arange(nwx)
    lo = np.maximum(x - max_disp, 0)
    hi = np.minimum(x + max_disp, nwx - 1)
    return int(nwy) * int((hi - lo + 1).sum())


def numpy_rows(seed):
    """One-thread numpy baselines of the same arithmetic on crops of the same pair:
    (a) the reference's form, one Python-level NCC evaluation per (window, offset)
    with numpy reductions over the 31 x 31 boxes (correlation.py:104-116 evaluated
    at full overlap); (b) a vectorised numpy port (one product image per offset,
    box sums through summed-area tables)."""
    rows = []
    left, right = synthetic_pair(WIN + 39, WIN + 63 + 2 * MAX_DISP, seed)
    lf, rf = left.astype(np.float64), right.astype(np.float64)
    n_eval = 0
    t0 = time.perf_counter()
    for y in range(left.shape[0] - WIN + 1):
        for x in range(MAX_DISP, MAX_DISP + 64):
            t = lf[y:y + WIN, x:x + WIN]
            tc = t - t.mean()
            ts = t.std()
            for d in range(-MAX_DISP, MAX_DISP + 1, 8):           # every 8th offset: bounded sample
                r = rf[y:y + WIN, x + d:x + d + WIN]
                _ = np.sum(tc * (r - r.mean())) / t.size / (ts * r.std())
                n_eval += 1
    dt = time.perf_counter() - t0
    rows.append({'value': n_eval / dt, 'unit': 'NCC window evals/s', 'cores': 1, 'kind': 'port',
                 'sample': f'{n_eval} evaluations in {dt:.2f} s, one numpy NCC per (window, offset) '
                           f'as the reference evaluates them'})
    ch, cw = WIN + 255, WIN + 1023
    left, right = synthetic_pair(ch, cw, seed)
    t0 = time.perf_counter()
    n_eval = numpy_band_scan(left, right)
    dt = time.perf_counter() - t0
    rows.append({'value': n_eval / dt, 'unit': 'NCC window evals/s', 'cores': 1, 'kind': 'port',
                 'sample': f'{ch}x{cw} crop, {n_eval} evaluations in {dt:.2f} s, vectorised numpy '
                           f'(product image per offset + summed-area tables)'})
    return rows


def numpy_band_scan(left, right):
    """The band scan's arithmetic in vectorised numpy (one product image per offset, box
    sums through summed-area tables, running maximum per window); returns the number of
    (window, offset) evaluations made."""
    lf, rf = left.astype(np.float64) - 128.0, right.astype(np.float64) - 128.0

    def box(a):
        sat = np.zeros((a.shape[0] + 1, a.shape[1] + 1))
        np.cumsum(np.cumsum(a, axis=0), axis=1, out=sat[1:, 1:])
        return sat[WIN:, WIN:] - sat[:-WIN, WIN:] - sat[WIN:, :-WIN] + sat[:-WIN, :-WIN]
    n = float(WIN * WIN)
    sl, sr = box(lf), box(rf)
    al, ar = n * box(lf * lf) - sl * sl, n * box(rf * rf) - sr * sr
    nwx = left.shape[1] - WIN + 1
    best = np.full(sl.shape, -np.inf)
    n_eval = 0
    with np.errstate(all='ignore'):
        for d in range(-MAX_DISP, MAX_DISP + 1):
            lo, hi = max(0, -d), min(nwx, nwx - d)                # windows whose box at x + d is inside
            if hi <= lo:
                continue
            prod = np.zeros_like(lf)
            prod[:, lo:hi + WIN - 1] = lf[:, lo:hi + WIN - 1] * rf[:, lo + d:hi + d + WIN - 1]
            score = (n * box(prod)[:, lo:hi] - sl[:, lo:hi] * sr[:, lo + d:hi + d]) / \
                np.sqrt(al[:, lo:hi] * ar[:, lo + d:hi + d])
            np.maximum(best[:, lo:hi], score, out=best[:, lo:hi])
            n_eval += score.size
    return n_eval


def _numpy_band_worker(job):
    seed, rows = job
    left, right = synthetic_pair(WIN + rows - 1, WIN + 1023, seed)
    return numpy_band_scan(left, right)


def numpy_all_cores(seed, cores):
    """The vectorised numpy form on every granted host core: one process per core
    (multiprocessing, forked before this process touches the GPU), each scanning a band
    of 96 window rows of a pair of its own."""
    import multiprocessing as mp
    rows = 96
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_numpy_band_worker, [(seed + 7 * i, 8) for i in range(cores)])        # workers up, numpy warm
        t0 = time.perf_counter()
        counts = pool.map(_numpy_band_worker, [(seed + i, rows) for i in range(cores)])
        dt = time.perf_counter() - t0
    return {'value': sum(counts) / dt, 'unit': 'NCC window evals/s', 'cores': cores, 'kind': 'port',
            'sample': f'{cores} processes x ({rows} window rows x 1024 windows x {2 * MAX_DISP + 1} offsets), '
                      f'{sum(counts)} evaluations in {dt:.2f} s, vectorised numpy via multiprocessing'}


def cpu_baseline(height, width, seed):
    """The C oracle's band scan (same semantics, float64, OpenMP over window rows)
    on a crop of the same synthetic pair; one-thread numpy rows beside it."""
    from oracle import ncc_oracle
    ncc_oracle.build()
    ch, cw = 31 + 255, 31 + 1023        # 256 x 1024 windows: a few seconds on 16 cores
    left, right = synthetic_pair(ch, cw, seed)
    cores = ncc_oracle.max_threads()
    t0 = time.perf_counter()
    ncc_oracle.band_scan(left, right, WIN, WIN, MAX_DISP, mode='fold', threshold=THRESHOLD)
    dt = time.perf_counter() - t0
    evals = count_evals(ch, cw, WIN, MAX_DISP)
    out = {'value': evals / dt, 'unit': 'NCC window evals/s', 'cores': cores, 'kind': 'port',
           'sample': f'{ch}x{cw} crop of the same synthetic pair, {evals} evals in {dt:.2f} s '
                     f'(oracle/ncc_oracle.c orc_band_scan, OpenMP)',
           'numpy_one_thread': numpy_rows(seed),
           'numpy_all_cores': numpy_all_cores(seed, cores)}
    try:
        import numba  # noqa: F401
        out['numba'] = 'importable but not timed: the reference\'s njit kernels are 1D only'
    except ImportError:
        out['numba'] = 'unavailable'
    return out


class Sharded:
    """One rank's share of a scan: a band of window rows of one pair (its image rows
    plus a halo of WIN - 1 rows) with the buffers its results travel in."""

    def __init__(self, bs, torch, left, right, rows, world, rank, wire, wire_windows, device='cuda'):
        begin, end = rows
        self.rows = rows
        h = end - begin + WIN - 1
        self.scanner = bs.BandScanner(h, left.shape[1], WIN, WIN, MAX_DISP)
        self.scanner.upload(left[begin:begin + h], right[begin:begin + h])
        self.nwin = self.scanner.disp.numel()
        self.torch, self.world, self.rank, self.wire = torch, world, rank, wire
        # Results travel as one byte buffer per rank: int16 disparities, then float32
        # correlations (16-bit integers are not a collective dtype).  Every rank's
        # buffer has the size of the largest share (shares differ by one window row at
        # most): a gather wants equal sizes.
        self.cap = cap = wire_windows + (wire_windows & 1)      # the float32 half starts 4-byte aligned
        self.packed, self.disp16, self.val32 = [], [], []
        for _ in range(2 if world > 1 else 0):          # two buffers take turns
            buf = torch.zeros(6 * cap, dtype=torch.uint8, device=device)
            self.packed.append(buf)
            self.disp16.append(buf[:2 * cap].view(torch.int16)[:self.nwin].view(self.scanner.disp.shape))
            self.val32.append(buf[2 * cap:].view(torch.float32)[:self.nwin].view(self.scanner.val.shape))
        self.gathered = None
        if world > 1 and rank == 0:
            self.gathered = [[torch.empty(6 * cap, dtype=torch.uint8, device=wire) for _ in range(world)]
                             for _ in range(2)]
        self.in_flight = [None, None]
        self.issued = 0

    def step(self, dist, events=None):
        sc = self.scanner
        sc.prepare()                                        # K1: SATs + window statistics of the band
        if events:
            events[0].record()
        sc.run(THRESHOLD, prepared=True)                    # K2: band kernel
        if events:
            events[1].record()
        if self.world > 1:                                  # results -> rank 0 over RCCL
            b = self.issued % 2
            self.issued += 1
            if self.in_flight[b] is not None:
                self.in_flight[b].wait()                    # this buffer's previous gather has landed
            self.disp16[b].copy_(sc.disp)
            self.val32[b].copy_(sc.val)
            self.in_flight[b] = dist.gather(self.packed[b].to(self.wire),
                                            self.gathered[b] if self.rank == 0 else None, dst=0,
                                            async_op=True)

    def drain(self):
        for b in range(2):
            if self.in_flight[b] is not None:
                self.in_flight[b].wait()
                self.in_flight[b] = None

    def last_gathered(self, counts):
        """(disparity int16, correlation float32) of every rank's last step, on the
        host, trimmed to that rank's `counts[r]` windows."""
        b = (self.issued - 1) % 2
        return [unpack_results(buf.cpu().numpy(), self.cap, n) for buf, n in zip(self.gathered[b], counts)]


def unpack_results(raw, cap, n):
    """A rank's wire buffer (uint8: `cap` int16 disparities, then `cap` float32
    correlations) -> its first n windows."""
    return raw[:2 * cap].view(np.int16)[:n].copy(), raw[2 * cap:].view(np.float32)[:n].copy()


def timed(torch, dist, world, wire, job, steps, warmup, events=None, sync=None):
    """W untimed steps, then exactly K timed ones between barriers and device
    synchronisations; the maximum over ranks.  `sync` replaces
    torch.cuda.synchronize (the CPU rehearsal of the multi-rank path has no device)."""
    sync = sync or torch.cuda.synchronize
    for _ in range(warmup):
        job.step(dist)
    job.drain()
    sync()
    if world > 1:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    for i in range(steps):
        job.step(dist, (events[0][i], events[1][i]) if events else None)
    job.drain()
    sync()
    if world > 1:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=wire)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def slowest_rank(torch, dist, world, wire, kernel_ms, achieved, evals):
    """The band kernel's figures on the rank where it ran at the lowest rate: (average
    launch duration in ms, TFLOP/s, evaluations per launch).  Every rank times its own
    launches; the roofline entry must not flatter the job with rank 0's device."""
    if world == 1:
        return kernel_ms, achieved, evals
    mine = torch.tensor([kernel_ms, achieved, float(evals)], dtype=torch.float64, device=wire)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    rows = [t.cpu().tolist() for t in every]
    worst = min(rows, key=lambda r: r[1])
    return worst[0], worst[1], int(worst[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--height', type=int, default=4096)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-weak', action='store_true', help='skip the secondary weak-scaling measurement')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # the CPU rows first: their worker processes are forked before this process holds a GPU context
    cpu_rows = cpu_baseline(args.height, args.width, seed=1000) if world == 1 and not args.no_cpu_baseline else None

    import torch
    import torch.distributed as dist
    # SVAMD_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer
    # GPUs than ranks (ranks share devices, results travel through host memory)
    backend = os.environ.get('SVAMD_BENCH_BACKEND', 'nccl')
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device_index))
        else:
            dist.init_process_group(backend)
    wire = 'cuda' if backend == 'nccl' else 'cpu'

    import band_scan as bs
    import sharding

    H, W = args.height, args.width
    nwy, nwx = H - WIN + 1, W - WIN + 1
    left, right = synthetic_pair(H, W, seed=1000)           # the same pair on every rank
    bands = [sharding.shard_range(nwy, r, world) for r in range(world)]
    counts = [(e - b) * nwx for b, e in bands]
    job = Sharded(bs, torch, left, right, bands[rank], world, rank, wire, max(counts))
    evals = count_evals(H, W, WIN, MAX_DISP)
    my_evals = count_evals(H, W, WIN, MAX_DISP, rows=bands[rank][1] - bands[rank][0])

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    elapsed = timed(torch, dist, world, wire, job, args.steps, args.warmup, (ev0, ev1))
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps
    achieved = my_evals * 2.0 * WIN * WIN / (kernel_ms * 1e-3) / 1e12
    kernel_ms, achieved, slowest_evals = slowest_rank(torch, dist, world, wire, kernel_ms, achieved, my_evals)

    # the gathered map against one rank scanning the whole pair alone
    check = None
    found = None
    if rank == 0:
        if world > 1:
            whole = bs.BandScanner(H, W, WIN, WIN, MAX_DISP)
            whole.upload(left, right)
            whole.run(THRESHOLD)
            wd, wv = whole.disp.cpu().numpy(), whole.val.cpu().numpy().astype(np.float32)
            parts = job.last_gathered(counts)
            gd = np.concatenate([d for d, _ in parts]).reshape(nwy, nwx)
            gv = np.concatenate([v for _, v in parts]).reshape(nwy, nwx)
            check = bool(np.array_equal(gd, wd.astype(np.int16)) and np.array_equal(gv, wv))
            found = int((gd != 0).sum())
            del whole
        else:
            found = int((job.scanner.disp != 0).sum().item())

    # secondary: the same scan through the box-sum kernel (identical results by construction and
    # checked here), K1 + kernel per step, no gather
    box = None
    sc = job.scanner
    if sc.box_sums_supported():
        sc.run(THRESHOLD, prepared=False)
        want_d, want_v = sc.disp.clone(), sc.val.clone()
        sc.run(THRESHOLD, method='sums')
        same = bool(torch.equal(sc.disp, want_d) and torch.equal(sc.val.view(torch.int64), want_v.view(torch.int64)))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sc.prepare()
            sc.run(THRESHOLD, prepared=True, method='sums')
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        box_elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([box_elapsed, 0.0 if same else 1.0], dtype=torch.float64, device=wire)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            box_elapsed, same = float(t[0].item()), bool(t[1].item() == 0.0)
        box = {'value': evals * args.steps / box_elapsed, 'unit': 'NCC window evals/s',
               'ms_per_step': box_elapsed / args.steps * 1e3, 'identical_to_the_timed_kernel': same,
               'what': 'the same scan with SVAMD_BAND_BOX_SUMS: cross terms from per-offset summed-area tables '
                       'of the product image in LDS (tables + kernel per step, no gather); not the headline, '
                       'whose metric counts one multiply-add per window pixel'}

    weak = None
    if world > 1 and not args.no_weak:
        # secondary: a pair of its own per rank (weak scaling), same step structure
        del job
        torch.cuda.empty_cache()
        l2, r2 = synthetic_pair(H, W, seed=1000 + rank)
        wjob = Sharded(bs, torch, l2, r2, (0, nwy), world, rank, wire, nwy * nwx)
        wel = timed(torch, dist, world, wire, wjob, args.steps, args.warmup)
        weak = world * evals * args.steps / wel

    if rank == 0:
        flops_per_launch = slowest_evals * 2.0 * WIN * WIN
        line = {
            'metric': 'NCC window evals/sec (whole node)',
            'value': evals * args.steps / elapsed,
            'unit': 'NCC window evals/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': f'dense band scan, ONE synthetic {H}x{W} uint8 stereo pair per step sharded over '
                            f'{world} GPU(s) by bands of window rows, {WIN}x{WIN} window at every pixel, '
                            f'{2 * MAX_DISP + 1} offsets (-{MAX_DISP}..+{MAX_DISP}), selection rule '
                            f'threshold {THRESHOLD}',
                'evals_per_pair': evals,
                'pairs_per_step': 1,
                'window_rows_per_rank': [e - b for b, e in bands],
                'windows_with_nonzero_disparity': found,
            },
            'roofline': {
                'bound': 'valu',
                'kernel': 'band_scan_kernel<31>',
                'achieved': achieved,
                'peak': FP32_VALU_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / FP32_VALU_PEAK_TFLOPS,
                'kernel_ms': kernel_ms,
                'kernel_ms_is': 'average launch of the slowest rank',
                'flops_per_launch': flops_per_launch,
                'traffic': None,
            },
        }
        if check is not None:
            line['config']['gathered_map_equals_single_gpu_scan'] = check
        if box is not None:
            line['box_sum_variant'] = box
        if weak is not None:
            line['scaling_weak_value'] = weak
            line['scaling_weak_note'] = f'{world} pairs per step, one per GPU, results gathered to rank 0'
        traffic = measured_traffic()
        if traffic and (H, W) == (4096, 4096) and world == 1:
            line['roofline']['traffic'] = traffic['bytes_per_launch']
            line['roofline']['traffic_source'] = traffic['source']
        if cpu_rows is not None:
            line['cpu_baseline'] = cpu_rows
        print(json.dumps(line), flush=True)
        if check is False:
            raise SystemExit('the gathered disparity map differs from the single-GPU scan')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
