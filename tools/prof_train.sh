#!/bin/bash
# kernel trace of bench.py --mode train (GPU box, repo root): per-kernel stats CSV + launch-by-launch timeline of the last step -> gpurun_out/<tag>/
# usage: bash tools/prof_train.sh <tag> [extra bench args]
tag=${1:-train}; shift
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pt_kt
rocprofv3 --kernel-trace --stats -d /tmp/pt_kt -o kt -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 3 --warmup 1 $* > $out/bench_profiled.json 2> $out/kt.err
db=$(ls /tmp/pt_kt/*.db | head -1)
python3 $R/tools/rocpd_stats.py $db 4 > $out/kernel_stats.csv
(cd $R/tools && python3 rocpd_timeline.py $db > $out/timeline.txt)
head -30 $out/kernel_stats.csv; tail -1 $out/timeline.txt; head -1 $out/timeline.txt
