#!/bin/bash
# kernel trace of an arbitrary bench.py command (GPU box, repo root): per-kernel stats CSV (+ timeline of the last step when an end marker is given)
# usage: bash tools/prof_cmd.sh <tag> <steps+warmup> [end_marker|-] <bench args...>
tag=$1; nsteps=$2; marker=$3; shift 3
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pc_kt
rocprofv3 --kernel-trace --stats -d /tmp/pc_kt -o kt -- python3 $R/bench.py "$@" > $out/bench_profiled.json 2> $out/kt.err
db=$(ls /tmp/pc_kt/*.db | head -1)
python3 $R/tools/rocpd_stats.py $db $nsteps > $out/kernel_stats.csv
[ "$marker" != "-" ] && (cd $R/tools && python3 rocpd_timeline.py $db $marker > $out/timeline.txt)
head -40 $out/kernel_stats.csv
