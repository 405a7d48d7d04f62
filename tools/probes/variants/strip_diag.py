#!/usr/bin/env python3
"""Removes the timing-diagnostic preprocessor branches of a kernel source (macros that are never defined in the shipped build), keeping the code the
default build compiles.  Round 4: used once to take the *_DIAG / *_CLK / *_NO_* switches out of few-shot-vit_amd/csrc; the removed branches live
on as patches next to this script (git diff -R of the strip: `git apply tools/probes/variants/<file>.diag.patch` puts a kernel's switches back for
an A/B with tools/build_variant.sh).
usage: strip_diag.py file.hip MACRO|MACRO=VALUE ...   (rewrites the file in place; MACRO=VALUE: a tuning knob frozen at its shipped default, used in #if lines only)"""
import re
import sys


def strip(text, args):
    fixed = dict(a.split('=') for a in args if '=' in a)
    macros = [a.split('=')[0] for a in args]
    pat = re.compile(r'\b(' + '|'.join(map(re.escape, macros)) + r')\b')

    def value(expr):          # the condition with every diagnostic macro undefined (None: not ours)
        if not pat.search(expr):
            return None
        e = re.sub(r'//.*', '', expr)
        e = re.sub(r'defined\s*\(\s*(\w+)\s*\)', lambda m: ('1' if m.group(1) in fixed else '0') if m.group(1) in macros else m.group(0), e)
        e = pat.sub(lambda m: fixed.get(m.group(1), '0'), e)
        if re.search(r'[A-Za-z_]', e):
            return None
        e = e.replace('&&', ' and ').replace('||', ' or ').replace('!', ' not ')
        return bool(eval(e))

    out, stack = [], []       # stack entries: [ours, taken_before, emitting_now, parent_emitting]
    emitting = True
    for line in text.split('\n'):
        s = line.strip()
        m = re.match(r'#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)', s)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ('ifdef', 'ifndef', 'if'):
            if kind == 'if':
                v = value(rest)
            else:
                name = re.match(r'(\w+)', rest).group(1)
                v = None if name not in macros else ((kind == 'ifndef') != (name in fixed))
            if v is None:
                stack.append([False, False, emitting, emitting])
                if emitting:
                    out.append(line)
            else:
                stack.append([True, v, emitting and v, emitting])
                emitting = emitting and v
        elif kind in ('elif', 'else'):
            top = stack[-1]
            if not top[0]:
                if emitting:
                    out.append(line)
                continue
            if kind == 'else':
                v = not top[1]
            else:
                v = value(rest)
                if v is None:
                    raise SystemExit('mixed #elif: ' + line)
                v = v and not top[1]
            top[1] = top[1] or v
            emitting = top[3] and v
        else:
            top = stack.pop()
            if not top[0]:
                if emitting:
                    out.append(line)
            emitting = top[3]
    assert not stack
    return '\n'.join(out)


if __name__ == '__main__':
    path, macros = sys.argv[1], sys.argv[2:]
    src = open(path).read()
    open(path, 'w').write(strip(src, macros))
