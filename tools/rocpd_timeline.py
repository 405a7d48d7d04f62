#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST step in a rocprofv3 kernel trace (rocpd SQLite): start offset, duration, gap to the previous kernel.
A step ends with the kernel whose name contains <end_marker> (default: sgd_multi_kernel = the optimizer launch of bench.py --mode train).
    python tools/rocpd_timeline.py gpurun_out/prof/x_results.db [end_marker] > profiles/rNN_train_timeline.txt"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'sgd_multi_kernel'
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = db.execute(f'select {name_col}, start, end from kernels order by start').fetchall()
    names = {}
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(ends) < 2:
        lo, hi = 0, len(rows) - 1
    else:
        lo, hi = ends[-2] + 1, ends[-1]
    t0 = rows[lo][1]
    prev_end = t0
    busy = 0
    print(f'# launches {hi - lo + 1}, wall {(rows[hi][2] - t0) / 1e3:.1f} us')
    print('#   start_us   dur_us   gap_us  kernel')
    for n, s, e in rows[lo:hi + 1]:
        if n not in names:
            names[n] = short(n)
        print(f'{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {names[n]}')
        busy += e - s
        prev_end = e
    print(f'# kernel time {busy / 1e3:.1f} us, gaps {(rows[hi][2] - t0 - busy) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
