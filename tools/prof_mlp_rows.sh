#!/bin/bash
# Kernel-only durations of mlp_rows (the C-ABI op also packs the weight image per call): rocprofv3 kernel trace over tools/bench_mlp_rows.py
# usage (GPU box, repo root): bash tools/prof_mlp_rows.sh [libfsvit variant .so]
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/mrprof
rocprofv3 --kernel-trace -d /tmp/mrprof -o p -- python3 $R/tools/bench_mlp_rows.py ${1:+$R/$1} > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/mrprof/*.db | head -1) | grep -E "mlp_rows|mlp_pack"
