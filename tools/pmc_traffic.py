#!/usr/bin/env python3
"""HBM bytes per launch of every kernel from two separate rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE and WRITE_SIZE cannot share a pass; rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request
of a wide coalesced read, so it is doubled; WRITE_SIZE is used as reported).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1
    python tools/pmc_traffic.py <fetch.db> <write.db> "<note>" > profiles/rNN_hbm_traffic.json
"""
import json
import sqlite3
import sys

from csrc_hash import csrc_sha
from rocpd_stats import short


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name', (counter,)).fetchall()
    out = {}
    for name, n, tot in rows:
        k = short(name)
        c, t = out.get(k, (0, 0.0))
        out[k] = (c + n, t + tot)
    return out


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    import re
    m1, m2 = re.search(r'--steps (\d+)', note), re.search(r'--warmup (\d+)', note)
    res = {'_meta': {'csrc_sha': csrc_sha(), 'steps_profiled': (int(m1.group(1)) + int(m2.group(1))) if m1 and m2 else None, 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); ' + note,
                     'units': 'bytes per launch; rocprofv3 reports KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B '
                              'request on wide coalesced reads); WRITE_SIZE uncorrected'}}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        if k not in write or k.startswith('at::') or k.startswith('__amd'):
            continue
        nf, tf = fetch[k]
        nw, tw = write[k]
        f = 2.0 * 1024.0 * tf / nf
        w = 1024.0 * tw / nw
        res[k] = {'launches': nf, 'fetch_bytes_per_launch_corrected': f, 'write_bytes_per_launch': w, 'hbm_bytes_per_launch': f + w}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
