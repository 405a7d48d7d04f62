bash tools/pmc_bench.sh r05 > gpurun_out/pmc_r05.log 2>&1
bash tools/pmc_bench.sh r05_train --mode train > gpurun_out/pmc_r05_train.log 2>&1
bash tools/pmc_bench.sh r05_deit --model deit_small_patch16_224 > gpurun_out/pmc_r05_deit.log 2>&1
bash tools/pmc_bench.sh r05_distill --mode distill > gpurun_out/pmc_r05_distill.log 2>&1
bash tools/prof_train.sh r05_train_tl > gpurun_out/prof_r05_train.log 2>&1
bash tools/prof_train_serial.sh r05_train_serial > gpurun_out/prof_r05_train_serial.log 2>&1
bash tools/pmc_kernel.sh r05_mlp_train tools/pmc_mlp_train.py 3 > gpurun_out/pmc_r05_mlp_train.log 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/e2e; rocprofv3 --kernel-trace -d /tmp/e2e -o kt -- python3 $R/tools/prof_e2e.py > $R/gpurun_out/r05_e2e.txt 2> /dev/null
python3 $R/tools/prof_e2e.py --gaps $(find /tmp/e2e -name "*.db" | head -1) > $R/gpurun_out/r05_e2e_gaps.txt
cd $R; ls gpurun_out/r05 gpurun_out/r05_train gpurun_out/r05_deit gpurun_out/r05_distill | head -50; tail -2 gpurun_out/r05_e2e_gaps.txt
