#!/usr/bin/env python3
"""Writes a copy of csrc/stage1_w4.hip with s_memtime stamps at the segment boundaries of the pipelined body (the W4_STAMP hooks) and builds
tools/probes/variants/libfsvit_w4clk.so from it; `python tools/bench_stage1.py 12800 tools/probes/variants/libfsvit_w4clk.so` then prints the cycles
per body of workgroups 0 and 100, per wave: S1 | barrier C + staging | S2 | S3 | barrier D | S4.   (round 6; replaces stage1_w4.diag.patch's W4_CLK part)
usage: python tools/probes/variants/w4_clk.py [extra hipcc flags]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
src = os.path.join(R, 'few-shot-vit_amd', 'csrc', 'stage1_w4.hip')
s = open(src).read()
s = s.replace('#define W4_STAMP(i) do { } while (0)\n',
              '  long long ck[6] = {0, 0, 0, 0, 0, 0}, ckt = 0;\n#define W4_STAMP(i) do { const long long n_ = __builtin_readcyclecounter(); ck[i] += n_ - ckt; ckt = n_; } while (0)\n')
s = s.replace('  auto body_pipe = [&](int q) {\n', '  auto body_pipe = [&](int q) {\n    ckt = __builtin_readcyclecounter();\n', 1)
tail = '''  if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && q1 - q0 > 3)
    printf("[w4 wg %d wave %d] bodies %d | S1 %lld  barC %lld  S2 %lld  S3 %lld  barD %lld  S4 %lld  (cycles per body)\\n", (int)blockIdx.x, w, q1 - q0 - 3,
           ck[0] / (q1 - q0 - 3), ck[1] / (q1 - q0 - 3), ck[2] / (q1 - q0 - 3), ck[3] / (q1 - q0 - 3), ck[4] / (q1 - q0 - 3), ck[5] / (q1 - q0 - 3));
}

// The engines' stage-1 kernel'''
assert s.count('}\n\n// The engines\' stage-1 kernel') == 1
s = s.replace("}\n\n// The engines' stage-1 kernel", tail, 1)
csrc = os.path.dirname(src)
tmp = os.path.join(csrc, 'stage1_w4_clksrc.hip')
open(tmp, 'w').write(s)
out = os.path.join(R, 'tools', 'probes', 'variants')
try:
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function', '-Wno-c++20-extensions'] + sys.argv[1:] +
                   ['-c', tmp, '-o', os.path.join(out, 'stage1_w4_clk.o')], check=True, cwd=csrc)
finally:
    os.remove(tmp)
objs = [os.path.join(csrc, 'build', f) for f in sorted(os.listdir(os.path.join(csrc, 'build'))) if f.endswith('.o') and f != 'stage1_w4.o']
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libfsvit_w4clk.so')] + objs + [os.path.join(out, 'stage1_w4_clk.o')], check=True)
print('built', os.path.join(out, 'libfsvit_w4clk.so'))
