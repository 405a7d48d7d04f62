#!/bin/bash
# kernel trace + the two SQ counter passes of ONE python driver (GPU box, repo root): bash tools/pmc_kernel.sh <tag> <script.py> [args]
tag=$1; shift
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python3 $R/$*"
rm -rf /tmp/pk_*
rocprofv3 --kernel-trace --stats -d /tmp/pk_kt -o kt -- $B > /dev/null 2> $out/kt.err
python3 $R/tools/rocpd_stats.py $(ls /tmp/pk_kt/*.db | head -1) 1 > $out/kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/pk_s1 -o s1 -- $B > /dev/null 2> $out/s1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pk_s2 -o s2 -- $B > /dev/null 2> $out/s2.err
python3 $R/tools/pmc_mfma.py $(ls /tmp/pk_s1/*.db | head -1) $(ls /tmp/pk_s2/*.db | head -1) "$B" > $out/mfma_pmc.json 2> $out/mfma_pmc.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pk_f -o f -- $B > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pk_w -o w -- $B > /dev/null 2> $out/w.err
python3 $R/tools/pmc_traffic.py $(ls /tmp/pk_f/*.db | head -1) $(ls /tmp/pk_w/*.db | head -1) "$B" > $out/hbm_traffic.json
grep -E "mlp_pair|kernel," $out/kernel_stats.csv
grep -E "mlp_pair" $out/mfma_pmc.txt
python3 - <<P
import json
t=json.load(open("$out/hbm_traffic.json"))
for k,v in t.items():
    if 'mlp_pair' in k: print(k, {a:round(b/1e6,1) if isinstance(b,float) else b for a,b in v.items()})
P
