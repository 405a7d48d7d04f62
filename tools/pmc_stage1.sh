#!/bin/bash
# before / after PMC of the stage-1 block operator: stage1_ring (FSVIT_STAGE1_W4=0) vs stage1_w4, 12 800 images, two SQ counter passes each
# usage (GPU box, repo root): bash tools/pmc_stage1.sh <tag>   -> gpurun_out/<tag>/stage1_pmc.txt
tag=${1:-r04}
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
: > $out/stage1_pmc.txt
for w4 in 0 1; do
  export FSVIT_STAGE1_W4=$w4
  rm -rf /tmp/ps_s1 /tmp/ps_s2 /tmp/ps_kt
  rocprofv3 --kernel-trace --stats -d /tmp/ps_kt -o kt -- python3 $R/tools/pmc_stage1.py 12800 > $out/stage1_w4_$w4.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/ps_s1 -o s1 -- python3 $R/tools/pmc_stage1.py 12800 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/ps_s2 -o s2 -- python3 $R/tools/pmc_stage1.py 12800 > /dev/null 2>&1
  echo "== FSVIT_STAGE1_W4=$w4 ($(grep 'us per launch' $out/stage1_w4_$w4.log | tail -1))" >> $out/stage1_pmc.txt
  python3 $R/tools/rocpd_stats.py $(ls /tmp/ps_kt/*.db | head -1) 1 | grep -i "stage1" >> $out/stage1_pmc.txt
  python3 $R/tools/pmc_mfma.py $(ls /tmp/ps_s1/*.db | head -1) $(ls /tmp/ps_s2/*.db | head -1) "pmc_stage1.py 12800" 2>&1 >/dev/null | grep -i "kernel \|stage1" >> $out/stage1_pmc.txt
done
cat $out/stage1_pmc.txt
