#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database:  python tools/pmc_summary.py <db> [kernel substring]"""
import sqlite3, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from rocpd_stats import short
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = db.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name').fetchall()
for name, ctr, n, avg in sorted(rows):
    k = short(name)
    if flt in k and not k.startswith(('at::', '__amd')):
        print(f'{k:60s} {ctr:32s} n={n:4d} avg={avg:.4g}')
