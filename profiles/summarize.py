#!/usr/bin/env python3
'''Prints a text summary of the rocprofv3 (rocpd sqlite) outputs that
profiles/run_profile.sh leaves under gpurun_out/prof_<tag>/: per-kernel time
statistics and, per kernel, the averaged hardware counters of each --pmc pass.

    python3 profiles/summarize.py gpurun_out/prof_<tag> > profiles/<tag>_summary.txt
'''
import glob
import os
import sqlite3
import sys


def main(root):
    stats = os.path.join(root, 'stats', 'stats_results.db')
    if os.path.exists(stats):
        db = sqlite3.connect(stats)
        print('== rocprofv3 --kernel-trace --stats (durations in microseconds) ==')
        print(f'{"calls":>6} {"total_us":>14} {"avg_us":>14} {"pct":>7}  kernel')
        for name, calls, total, avg, pct in db.execute(
                'select name, total_calls, total_duration, average, percentage from top_kernels'):
            print(f'{calls:>6} {total:>14.0f} {avg:>14.1f} {pct:>7.2f}  {name[:110]}')
        print()
    for path in sorted(glob.glob(os.path.join(root, 'pmc_*', 'pmc_results.db'))):
        db = sqlite3.connect(path)
        print(f'== counters: {os.path.basename(os.path.dirname(path))} (average per dispatch) ==')
        rows = db.execute(
            'select kernel_name, counter_name, count(*), avg(value), avg(duration) '
            'from counters_collection group by kernel_name, counter_name '
            'order by kernel_name, counter_name').fetchall()
        last = None
        for kernel, counter, n, value, dur in rows:
            if 'svamd' not in kernel:
                continue
            if kernel != last:
                print(f'  {kernel[:110]}  (dispatches: {n}, avg duration {dur:.0f} ns)')
                last = kernel
            print(f'      {counter:<24} {value:>22.1f}')
        print()


if __name__ == '__main__':
    main(sys.argv[1])
