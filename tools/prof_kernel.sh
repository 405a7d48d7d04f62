#!/bin/bash
# Kernel-only durations from a rocprofv3 kernel trace:  bash tools/prof_kernel.sh <python script under tools/> <kernel name pattern> [script args...]
# (GPU box, repo root; a relative variant .so path among the args is resolved against the repo root)
R=$PWD; script=$1; pat=$2; shift 2
args=(); for a in "$@"; do [ -f "$R/$a" ] && a="$R/$a"; args+=("$a"); done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/tools/$script "${args[@]}" > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) | grep -E "$pat"
