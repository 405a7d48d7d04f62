#!/usr/bin/env python3
"""MFMA / VALU / LDS utilisation of every kernel of the bench from rocprofv3 SQ counter passes (north star: "evidenced by rocprof MFMA
utilisation").  Input: one or two rocpd databases of

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d <dir> -- python3 bench.py --no-cpu-baseline --no-roofline --no-modes --steps 3 --warmup 1
    rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d <dir> -- (same command)

    python tools/pmc_mfma.py <db1> [<db2>] "<note>" > profiles/rNN_mfma_pmc.json        (text table on stderr)

Derived per kernel (means over its launches):
    mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)     fraction of the chip's matrix-pipe cycles in use
                     (SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs: 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16;
                     GRBM_GUI_ACTIVE = shader-clock cycles of the launch SUMMED over the 8 XCDs, as rocprofv3 reports it: a 1.3 ms launch reads ~2e7) - comparable with roofline.frac, which counts only the
                     ALGORITHMIC flops (padded head dims / K slices are MFMA-busy but not algorithmic)
    valu_active    = 4 * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES ... reported as fractions of wave cycles (quad-cycle counters):
    wave_valu, wave_lds, wave_wait_any (parked at s_waitcnt / barrier), wave_wait_inst (issue stalls), lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import json
import sqlite3
import sys

from csrc_hash import csrc_sha
from rocpd_stats import short


def load(db_path, acc):
    db = sqlite3.connect(db_path)
    rows = db.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name').fetchall()
    for name, ctr, n, avg in rows:
        k = short(name)
        if k.startswith(('at::', '__amd', 'void at::')):
            continue
        d = acc.setdefault(k, {})
        if ctr in d:                      # the same counter in both passes (SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE): keep the first, note the launches
            continue
        d[ctr] = avg
        d.setdefault('_launches', n)


def main():
    dbs = [a for a in sys.argv[1:] if a.endswith('.db')]
    note = ' '.join(a for a in sys.argv[1:] if not a.endswith('.db'))
    acc = {}
    for p in dbs:
        load(p, acc)
    out = {'_meta': {'csrc_sha': csrc_sha(), 'source': 'rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE passes (kernels serialised by the profiler); ' + note,
                     'mfma_busy': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)'}}
    rows = []
    for k, d in acc.items():
        gui = d.get('GRBM_GUI_ACTIVE')
        if not gui:
            continue
        wc = d.get('SQ_WAVE_CYCLES', 0.0)
        rec = {'launches': d['_launches'], 'gui_active_cycles': gui,
               'mfma_busy': d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8.0 * 1024.0)}
        for key, ctr in (('wave_valu', 'SQ_ACTIVE_INST_VALU'), ('wave_lds', 'SQ_ACTIVE_INST_LDS'), ('wave_wait_any', 'SQ_WAIT_ANY'),
                         ('wave_wait_inst', 'SQ_WAIT_INST_ANY'), ('wave_wait_inst_lds', 'SQ_WAIT_INST_LDS'), ('wave_active_any', 'SQ_ACTIVE_INST_ANY')):
            if ctr in d and wc:
                rec[key] = d[ctr] / wc
        if d.get('SQ_LDS_IDX_ACTIVE'):
            rec['lds_conflict'] = d.get('SQ_LDS_BANK_CONFLICT', 0.0) / d['SQ_LDS_IDX_ACTIVE']
            rec['lds_busy'] = d['SQ_LDS_IDX_ACTIVE'] / (gui / 8.0 * 256.0)
        for ctr in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT'):
            if ctr in d:
                rec[ctr] = d[ctr]
        out[k] = rec
        rows.append((gui * d['_launches'], k, rec))
    json.dump(out, sys.stdout, indent=1)
    print()
    tot = sum(r[0] for r in rows) or 1.0
    print('%-52s %6s %7s %7s %7s %7s %7s %7s' % ('kernel', 'share', 'mfma', 'valu', 'lds', 'waitany', 'waitins', 'ldsconf'), file=sys.stderr)
    for g, k, r in sorted(rows, reverse=True):
        print('%-52s %5.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%%' % (
            k[:52], 100 * g / tot, 100 * r['mfma_busy'], 100 * r.get('wave_valu', 0), 100 * r.get('wave_lds', 0), 100 * r.get('wave_wait_any', 0),
            100 * r.get('wave_wait_inst', 0), 100 * r.get('lds_conflict', 0)), file=sys.stderr)


if __name__ == '__main__':
    main()
