#!/usr/bin/env python3
"""The end_to_end leg of bench.py alone (test_few_shot.evaluate on a synthetic miniImageNet-format pickle), for a kernel trace:
   rocprofv3 --kernel-trace -d /tmp/e2e -o kt -- python3 tools/prof_e2e.py     then     python3 tools/prof_e2e.py --gaps /tmp/e2e/.../kt_results.db"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gaps(dbpath):
    import sqlite3
    db = sqlite3.connect(dbpath)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if 'kernels' in tabs:
        cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
        rows = db.execute('select %s, start, end from kernels order by start' % ('name' if 'name' in cols else 'kernel_name')).fetchall()
    else:
        kd = [t for t in tabs if 'kernel_dispatch' in t][0]
        ks = [t for t in tabs if 'kernel_symbol' in t][0]
        rows = db.execute(f'select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start').fetchall()
    # the last evaluate(): from the last image_transform launch that follows a gap > 50 ms
    starts = [i for i in range(1, len(rows)) if rows[i][1] - rows[i - 1][2] > 50e6]
    lo = starts[-1] if starts else 0
    seg = rows[lo:]
    t0, t1 = seg[0][1], seg[-1][2]
    busy = sum(e - s for _, s, e in seg)
    print(f'last evaluate: {len(seg)} launches, wall {(t1 - t0) / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms')
    big = sorted(((seg[i][1] - seg[i - 1][2], i) for i in range(1, len(seg))), reverse=True)[:12]
    for g, i in big:
        print(f'  gap {g / 1e3:9.1f} us at +{(seg[i][1] - t0) / 1e6:7.1f} ms between {seg[i - 1][0][:40]} and {seg[i][0][:40]}')
    by = {}
    for n, s, e in seg:
        by[n[:50]] = by.get(n[:50], 0) + e - s
    for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print(f'  {v / 1e6:8.2f} ms  {n}')


if len(sys.argv) > 2 and sys.argv[1] == '--gaps':
    gaps(sys.argv[2])
    sys.exit(0)

import torch
import bench
args = bench.parse(['--no-cpu-baseline'])
r = bench.end_to_end_leg(args, torch.device('cuda', 0))
print({k: v for k, v in r.items() if k != 'workload'})
