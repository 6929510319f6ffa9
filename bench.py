#!/usr/bin/env python3
'''Benchmark of the dense band scan (the headline workload): a synthetic
4096 x 4096 uint8 stereo pair, a 31 x 31 window at every pixel, 257 horizontal
candidate offsets (-128..+128), image_scan's selection rule fused in.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed over RCCL when N > 1).  Every rank scans
its own pair per step (weak scaling: the job is a batch of N pairs, i.e. row
bands of an N-times taller strip); only the per-window results (int16 disparity
+ float32 correlation) are gathered to rank 0.  A step = summed-area tables +
window statistics (K1) + the band kernel (K2) + the gather; the gather of step i
is issued asynchronously and travels while step i + 1 computes (two result
buffers take turns), every gather has completed before the clock stops.  Rank 0
prints one JSON line.

metric: NCC window evaluations per second, an evaluation being one (window,
offset) score = 2 * 31 * 31 algorithmic float32 flops of cross term.
roofline: the band kernel alone, timed with HIP events on its stream, against
the FP32 vector-ALU peak.  cpu_baseline: the C oracle (OpenMP) on a crop.
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


def count_evals(height, width, win, max_disp):
    '''Number of (window, offset) pairs whose candidate box lies inside the image.'''
    nwx, nwy = width - win + 1, height - win + 1
    x = np.arange(nwx)
    lo = np.maximum(x - max_disp, 0)
    hi = np.minimum(x + max_disp, nwx - 1)
    return int(nwy) * int((hi - lo + 1).sum())


def cpu_baseline(height, width, seed):
    '''The C oracle's band scan (same semantics, float64, OpenMP over window rows)
    on a crop of the same synthetic pair.'''
    from oracle import ncc_oracle
    ncc_oracle.build()
    ch, cw = 31 + 1023, 31 + 1023       # 1024 x 1024 windows
    left, right = synthetic_pair(ch, cw, seed)
    cores = ncc_oracle.max_threads()
    t0 = time.perf_counter()
    ncc_oracle.band_scan(left, right, WIN, WIN, MAX_DISP, mode='fold', threshold=THRESHOLD)
    dt = time.perf_counter() - t0
    evals = count_evals(ch, cw, WIN, MAX_DISP)
    return {'value': evals / dt, 'unit': 'NCC window evals/s', 'cores': cores, 'kind': 'port',
            'sample': f'{ch}x{cw} crop of the same synthetic pair, {evals} evals in {dt:.2f} s '
                      f'(oracle/ncc_oracle.c orc_band_scan, OpenMP)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--height', type=int, default=4096)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
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

    import band_scan as bs

    H, W = args.height, args.width
    left, right = synthetic_pair(H, W, seed=1000 + rank)
    scanner = bs.BandScanner(H, W, WIN, WIN, MAX_DISP)
    scanner.upload(left, right)
    evals = count_evals(H, W, WIN, MAX_DISP)

    # results travel as one byte buffer per rank: int16 disparities then float32
    # correlations (6 bytes per window; 16-bit integers are not a collective dtype)
    nwin = scanner.disp.numel()
    wire = 'cuda' if backend == 'nccl' else 'cpu'
    packed, disp16, val32, gathered = [], [], [], []
    for _ in range(2 if world > 1 else 0):          # two buffers take turns
        buf = torch.empty(6 * nwin, dtype=torch.uint8, device='cuda')
        packed.append(buf)
        disp16.append(buf[:2 * nwin].view(torch.int16).view(scanner.disp.shape))
        val32.append(buf[2 * nwin:].view(torch.float32).view(scanner.val.shape))
        gathered.append([torch.empty(6 * nwin, dtype=torch.uint8, device=wire) for _ in range(world)]
                        if rank == 0 else None)
    in_flight = [None, None]
    issued = [0]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        scanner.prepare()                                   # K1: SATs + window statistics
        if i is not None:
            ev0[i].record()
        scanner.run(THRESHOLD, prepared=True)               # K2: band kernel
        if i is not None:
            ev1[i].record()
        if world > 1:                                       # results -> rank 0 over RCCL
            b = issued[0] % 2
            issued[0] += 1
            if in_flight[b] is not None:
                in_flight[b].wait()                         # this buffer's previous gather has landed
            disp16[b].copy_(scanner.disp)
            val32[b].copy_(scanner.val)
            in_flight[b] = dist.gather(packed[b].to(wire), gathered[b], dst=0, async_op=True)

    def drain():
        for b in range(2):
            if in_flight[b] is not None:
                in_flight[b].wait()
                in_flight[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=wire)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps

    # sanity: the planted disparities are recovered on this rank's pair
    disp = scanner.disp.cpu().numpy()
    found = int((disp != 0).sum())

    if rank == 0:
        flops_per_launch = evals * 2.0 * WIN * WIN
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        line = {
            'metric': 'NCC window evals/sec (whole node)',
            'value': world * evals * args.steps / elapsed,
            'unit': 'NCC window evals/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': f'dense band scan, synthetic {H}x{W} uint8 stereo pair per GPU, '
                            f'{WIN}x{WIN} window at every pixel, {2 * MAX_DISP + 1} offsets '
                            f'(-{MAX_DISP}..+{MAX_DISP}), selection rule threshold {THRESHOLD}',
                'evals_per_pair': evals,
                'pairs_per_step': world,
                'windows_with_nonzero_disparity_rank0': found,
            },
            'roofline': {
                'bound': 'valu',
                'kernel': 'band_scan_kernel<31>',
                'achieved': achieved,
                'peak': FP32_VALU_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / FP32_VALU_PEAK_TFLOPS,
                'kernel_ms': kernel_ms,
                'flops_per_launch': flops_per_launch,
                'traffic': None,
            },
        }
        traffic = measured_traffic()
        if traffic and (H, W) == (4096, 4096):
            line['roofline']['traffic'] = traffic['bytes_per_launch']
            line['roofline']['traffic_source'] = traffic['source']
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(H, W, seed=1000)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
