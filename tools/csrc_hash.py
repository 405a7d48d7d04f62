#!/usr/bin/env python3
"""Fingerprint of the kernel sources (few-shot-vit_amd/csrc/*.hip, *.h, *.inc, Makefile): the PMC summary tools store it in `_meta.csrc_sha`
and bench.py compares it with the tree it runs from, so a roofline line never silently quotes counters of older kernels."""
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha(repo=REPO):
    d = os.path.join(repo, 'few-shot-vit_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.inc')) or f == 'Makefile':
            h.update(f.encode())
            with open(os.path.join(d, f), 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_sha())
