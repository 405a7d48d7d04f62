#!/bin/bash
# Kernel-only duration of the stage-1 block kernels: rocprofv3 kernel trace over tools/pmc_stage1.py
# usage (GPU box, repo root): [FSVIT_STAGE1_ROWS=1] bash tools/prof_stage1.sh [images] [libfsvit variant .so]
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/s1prof
rocprofv3 --kernel-trace -d /tmp/s1prof -o p -- python3 $R/tools/pmc_stage1.py ${1:-6400} ${2:+$R/$2} > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/s1prof/*.db | head -1) | grep -E "stage1"
