#!/usr/bin/env python3
'''Averages per kernel of the counters of one rocprofv3 --pmc run (rocpd sqlite).

    python3 profiles/pmc_summary.py <results.db> [kernel-name-filter]
'''
import sqlite3
import sys


def main(path, pattern=''):
    db = sqlite3.connect(path)
    rows = db.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                      'group by kernel_name, counter_name order by kernel_name, counter_name').fetchall()
    last = None
    for kernel, counter, n, value, dur in rows:
        if pattern not in kernel:
            continue
        if kernel != last:
            print(f'  {kernel[:100]}  (dispatches: {n}, avg duration {dur / 1e3:.1f} us)')
            last = kernel
        print(f'      {counter:<36} {value:>20.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
