#!/bin/bash
# SQ counter passes for the stage-1 block operator alone (tools/pmc_stage1.py): bash tools/pmc_stage1.sh <tag> [B] [lib]
tag=${1:-s1}; B=${2:-6400}; lib=${3:+$PWD/$3}
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ps_*
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/ps_1 -o s1 -- python3 $R/tools/pmc_stage1.py $B $lib > /dev/null 2> $out/s1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/ps_2 -o s2 -- python3 $R/tools/pmc_stage1.py $B $lib > /dev/null 2> $out/s2.err
python3 $R/tools/pmc_mfma.py $(ls /tmp/ps_1/*.db | head -1) $(ls /tmp/ps_2/*.db | head -1) "tools/pmc_stage1.py $B $lib" > $out/mfma_pmc.json 2> $out/mfma_pmc.txt
cat $out/mfma_pmc.txt
