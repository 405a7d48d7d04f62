#!/usr/bin/env python3
"""Static check of a gfx950 kernel's ISA for the hazard hipcc does not pad around inline-asm MFMAs (cdna_hip_programming.md 5.7 item 2):
(a) a VALU / v_accvgpr_write that writes a register an MFMA reads as SrcA / SrcB / SrcC within the next 2 wait states;
(b) a non-MFMA instruction that reads or writes an MFMA's destination within NEED wait states of its issue - 12 for the 8-pass 32x32x16 shape,
    8 for the 4-pass 16x16x32 one (what hipcc itself pads behind a builtin MFMA); an intervening MFMA counts as its passes, s_nop N as N + 1,
    anything else as 1.
   python3 tools/check_mfma_hazard.py build/X.o kernel_name_pattern
(c) a VALU write to the data registers of a global / buffer store of more than 8 bytes in the very next issue slot (hipcc keeps one instruction between
    them in its own code; an inline-asm store is not padded, cdna_asm_programming.md 4.1 rows 8 / 9).
Also prints, per kernel: instructions issued between consecutive MFMAs (histogram) - a single wave per SIMD issues one instruction per 4 cycles."""
import os
import re
import subprocess
import sys
import tempfile

L = '/opt/rocm/lib/llvm/bin/'
obj, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, 'k.fatbin'), os.path.join(td, 'k.co')
    subprocess.check_call([L + 'llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat])
    subprocess.check_call([L + 'clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
    asm = subprocess.check_output([L + 'llvm-objdump', '-d', '--no-show-raw-insn', co], text=True)


def regs(tok):
    """'v[10:25]' / 'a3' / 'v7' -> set of (file, index)"""
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([va])(\d+)', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


cur, body = None, {}
for line in asm.splitlines():
    m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
    if m:
        cur = m.group(1)
        body[cur] = []
        continue
    if cur and line.startswith('\t'):
        ins = line.split('//')[0].strip()
        if ins:
            body[cur].append(ins)
bad = 0
for name, ins in body.items():
    if not re.search(pat, name):
        continue
    gaps, last = {}, None
    for i, s in enumerate(ins):
        if not s.startswith('v_mfma'):
            continue
        if last is not None:
            g = i - last - 1
            gaps[g] = gaps.get(g, 0) + 1
        last = i
        ops = [o.strip() for o in s.split(None, 1)[1].split(',')]
        reads = set().union(*[regs(o) for o in ops[1:4]])
        states = 0
        for j in range(i - 1, max(i - 4, -1), -1):
            p = ins[j]
            w = p.split()[0]
            if w.startswith('s_nop'):
                states += int(p.split()[1]) + 1
            else:
                if w.startswith('v_') and not w.startswith('v_mfma') and not w.startswith('v_cmp'):
                    dst = [o.strip() for o in p.split(None, 1)[1].split(',')][0]
                    if regs(dst) & reads and states < 2:
                        bad += 1
                        print('HAZARD %s: "%s" -> "%s" with %d wait states' % (name[:40], p, s, states))
                states += 1
            if states >= 2:
                break
    for i, s in enumerate(ins):                      # (c)
        w0 = s.split()[0]
        if not (w0.endswith('store_dwordx4') or w0.endswith('store_dwordx3')) or w0.startswith('ds_') or w0.startswith('scratch_'):
            continue
        ops = [o.strip() for o in s.split(None, 1)[1].split(',')]
        data = regs(ops[1].split(' ')[0]) if w0.startswith('global_') else regs(ops[0].split(' ')[0])
        states = 0
        for j in range(i + 1, min(i + 2, len(ins))):
            q = ins[j]
            w = q.split()[0]
            if w.startswith('s_nop'):
                states += int(q.split()[1]) + 1
            else:
                if w.startswith('v_') and not w.startswith('v_cmp') and len(q.split(None, 1)) > 1:
                    dst = regs([o.strip() for o in q.split(None, 1)[1].split(',')][0])
                    if dst & data and states < 1:
                        bad += 1
                        print('HAZARD(c) %s: "%s" then "%s" after %d wait states' % (name[:40], s, q, states))
                states += 1
            if states >= 2:
                break
    for i, s in enumerate(ins):                      # (b)
        if not s.startswith('v_mfma'):
            continue
        dst = regs([o.strip() for o in s.split(None, 1)[1].split(',')][0])
        need = 12 if '32x32' in s else 8
        states = 0
        for j in range(i + 1, min(i + 14, len(ins))):
            q = ins[j]
            w = q.split()[0]
            if w.startswith('s_nop'):
                states += int(q.split()[1]) + 1
            elif w.startswith('v_mfma'):
                states += 8 if '32x32' in q else 4
            else:
                toks = set()
                for o in (q.split(None, 1)[1].split(',') if len(q.split(None, 1)) > 1 else []):
                    toks |= regs(o.strip().split(' ')[0])
                if toks & dst and states < need and not w.startswith('s_'):
                    bad += 1
                    print('HAZARD(b) %s: "%s" then "%s" after %d wait states' % (name[:40], s, q, states))
                states += 1
            if states >= need:
                break
    print(name[:60], 'MFMAs: %d; instructions between consecutive MFMAs:' % sum(1 for s in ins if s.startswith('v_mfma')),
          ', '.join('%d: %d' % (k, gaps[k]) for k in sorted(gaps)))
sys.exit(1 if bad else 0)
