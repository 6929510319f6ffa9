#!/usr/bin/env python3
'''Per-kernel call counts and average / minimum durations (microseconds) of a
rocprofv3 --kernel-trace run (rocpd sqlite output).

    python3 profiles/kernel_avgs.py <results.db> [name-filter]
'''
import sqlite3
import sys


def main(path, pattern=''):
    db = sqlite3.connect(path)
    rows = db.execute('select name, count(*), avg(end - start), min(end - start), sum(end - start) from kernels '
                      'group by name order by sum(end - start) desc').fetchall()
    print(f'{"calls":>6} {"avg_us":>10} {"min_us":>10} {"total_us":>12}  kernel')
    for name, n, avg, mn, tot in rows:
        if pattern in name:
            print(f'{n:>6} {avg / 1e3:>10.1f} {mn / 1e3:>10.1f} {tot / 1e3:>12.1f}  {name[:100]}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
