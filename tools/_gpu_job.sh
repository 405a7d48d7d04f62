timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.txt 2>&1; tail -2 gpurun_out/r05_gpu_tests.txt
bash tools/pmc_bench.sh r05 > gpurun_out/pmc_r05.log 2>&1
bash tools/pmc_bench.sh r05_train --mode train > gpurun_out/pmc_r05_train.log 2>&1
bash tools/pmc_bench.sh r05_deit --model deit_small_patch16_224 > gpurun_out/pmc_r05_deit.log 2>&1
bash tools/pmc_bench.sh r05_distill --mode distill > gpurun_out/pmc_r05_distill.log 2>&1
bash tools/prof_train.sh r05_train_tl > gpurun_out/prof_r05_train.log 2>&1
bash tools/prof_train_serial.sh r05_train_serial > gpurun_out/prof_r05_train_serial.log 2>&1
bash tools/pmc_kernel.sh r05_mlp_train tools/pmc_mlp_train.py 3 > gpurun_out/pmc_r05_mlp_train.log 2>&1
python tools/csrc_hash.py
