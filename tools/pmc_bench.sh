#!/bin/bash
# The round's rocprofv3 evidence for the default bench command (GPU box, repo root):
#   kernel trace (+ stats CSV), two SQ counter passes (MFMA / VALU / LDS utilisation), FETCH_SIZE and WRITE_SIZE passes (HBM traffic; each
#   alone - TCC slots - and never combined with a trace domain).  Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/.
# usage: bash tools/pmc_bench.sh <tag> [extra bench args]
tag=${1:-r02}; shift
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline --no-modes --no-legs --steps 3 --warmup 1 $*"
rm -rf /tmp/pb_*
rocprofv3 --kernel-trace --stats -d /tmp/pb_kt -o kt -- $B > $out/kt_bench.json 2> $out/kt.err
python3 $R/tools/rocpd_stats.py $(ls /tmp/pb_kt/*.db | head -1) 4 > $out/kernel_stats.csv   # 1 warm-up + 3 timed steps
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/pb_s1 -o s1 -- $B > /dev/null 2> $out/s1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pb_s2 -o s2 -- $B > /dev/null 2> $out/s2.err
python3 $R/tools/pmc_mfma.py $(ls /tmp/pb_s1/*.db | head -1) $(ls /tmp/pb_s2/*.db | head -1) "$B" > $out/mfma_pmc.json 2> $out/mfma_pmc.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pb_f -o f -- $B > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pb_w -o w -- $B > /dev/null 2> $out/w.err
python3 $R/tools/pmc_traffic.py $(ls /tmp/pb_f/*.db | head -1) $(ls /tmp/pb_w/*.db | head -1) "$B" > $out/hbm_traffic.json
tail -3 $out/*.err | head -40
cat $out/mfma_pmc.txt
