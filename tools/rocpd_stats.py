#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 rocpd SQLite result, as CSV.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.csv"""
import re
import sqlite3
import sys


def short(name):
    if name.startswith('_Z'):
        import subprocess
        try:
            name = subprocess.run(['c++filt', name.replace('DF16b', 'u6__bf16')], capture_output=True, text=True, timeout=10).stdout.strip() or name
        except Exception:
            pass
    name = re.sub(r'\[clone[^\]]*\]', '', name)
    name = re.sub(r'^void ', '', name)
    if name.startswith('at::native::'):       # ATen kernels: keep the functor names (what the launch was for)
        return re.sub(r'\(anonymous namespace\)::|at::native::|std::array<char\*,\d+ul> ?', '', name).replace(', ', ',')[:110]
    m = re.match(r'(?:fsvit::)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_:]+(?:<[^(]*>)?)', name)
    return (m.group(1) if m else name).replace(', ', ',')[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    dur = 'duration' if 'duration' in cols else '(end - start)'
    rows = db.execute(f'select {name_col}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name_col} order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    print('kernel,calls,total_ms,avg_us,min_us,max_us,percent,ms_per_step')
    for n, c, s, a, mn, mx in rows:
        print(f'"{short(n)}",{c},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.2f},{s / 1e6 / steps:.3f}')


if __name__ == '__main__':
    main()
