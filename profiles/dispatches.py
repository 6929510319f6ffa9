#!/usr/bin/env python3
'''Lists the kernel dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite
output) in launch order: start offset, duration, grid and kernel name.

    python3 profiles/dispatches.py <results.db> [name-filter] [context-before] [context-after]
'''
import sqlite3
import sys


def main(path, pattern='', before=0, after=0):
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start').fetchall()
    keep = set()
    for i, r in enumerate(rows):
        if pattern in r[0]:
            keep.update(range(max(0, i - before), min(len(rows), i + after + 1)))
    t0 = rows[0][1] if rows else 0
    for i in sorted(keep):
        r = rows[i]
        print(f'{(r[1] - t0) / 1e3:12.1f} us  +{(r[2] - r[1]) / 1e3:9.1f} us  grid {r[3]}x{r[4]}x{r[5]} / {r[6]}  {r[0][:90]}')


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0], a[1] if len(a) > 1 else '', int(a[2]) if len(a) > 2 else 0, int(a[3]) if len(a) > 3 else 0)
