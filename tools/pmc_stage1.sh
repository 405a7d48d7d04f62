#!/bin/bash
# PMC of the stage-1 block operator (stage1_w4.hip), 12 800 images, a kernel trace + two SQ counter passes
# usage (GPU box, repo root): bash tools/pmc_stage1.sh <tag>   -> gpurun_out/<tag>/stage1_pmc.txt
# (until round 5 this compared stage1_ring under FSVIT_STAGE1_W4=0; the switch is retired - tools/probes/variants/dispatch_switches.r06.patch)
tag=${1:-r06}
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
: > $out/stage1_pmc.txt
rm -rf /tmp/ps_s1 /tmp/ps_s2 /tmp/ps_kt
rocprofv3 --kernel-trace --stats -d /tmp/ps_kt -o kt -- python3 $R/tools/pmc_stage1.py 12800 > $out/stage1_w4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/ps_s1 -o s1 -- python3 $R/tools/pmc_stage1.py 12800 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/ps_s2 -o s2 -- python3 $R/tools/pmc_stage1.py 12800 > /dev/null 2>&1
echo "== $(grep 'us per launch' $out/stage1_w4.log | tail -1)" >> $out/stage1_pmc.txt
python3 $R/tools/rocpd_stats.py $(ls /tmp/ps_kt/*.db | head -1) 1 | grep -i "stage1" >> $out/stage1_pmc.txt
python3 $R/tools/pmc_mfma.py $(ls /tmp/ps_s1/*.db | head -1) $(ls /tmp/ps_s2/*.db | head -1) "pmc_stage1.py 12800" 2>&1 >/dev/null | grep -i "kernel \|stage1" >> $out/stage1_pmc.txt
cat $out/stage1_pmc.txt
