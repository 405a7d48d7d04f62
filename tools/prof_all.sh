set -x
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" 
bash tools/pmc_bench.sh r06 > /dev/null 2>&1
bash tools/pmc_bench.sh r06_train --mode train > /dev/null 2>&1
bash tools/pmc_bench.sh r06_distill --mode distill > /dev/null 2>&1
bash tools/pmc_bench.sh r06_deit --model deit_small_patch16_224 > /dev/null 2>&1
bash tools/pmc_stage1.sh r06_s1 > /dev/null 2>&1
ls gpurun_out/r06 gpurun_out/r06_train gpurun_out/r06_distill gpurun_out/r06_deit gpurun_out/r06_s1
cat gpurun_out/r06/mfma_pmc.txt | head -12
