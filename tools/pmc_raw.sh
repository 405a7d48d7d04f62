#!/bin/bash
# raw SQ counters of one operator bench (round 6): bash tools/pmc_raw.sh <tag> <kernel substring> <python script + args ...>
#   -> gpurun_out/<tag>_pmc_raw.txt: per-launch means of three counter passes, for the kernels whose name contains the substring
tag=$1; sub=$2; shift 2
R=$PWD; out=$R/gpurun_out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P3="SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH GRBM_GUI_ACTIVE"
P4="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_LOAD GRBM_GUI_ACTIVE"
: > $out/${tag}_pmc_raw.txt
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pr_$i
  rocprofv3 --pmc $P -d /tmp/pr_$i -o p -- python3 "$@" > /dev/null 2>&1
  python3 - "$(ls /tmp/pr_$i/*.db | head -1)" "$sub" >> $out/${tag}_pmc_raw.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, ctr, n, avg in db.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'):
    if sys.argv[2] in name:
        print('%-44s %-30s n=%-3d %.4g' % (name[:44], ctr, n, avg))
PY
done
cat $out/${tag}_pmc_raw.txt
