set -x
R=$PWD
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -15
python bench.py --mode train --no-cpu-baseline > gpurun_out/r02_train_g1.json 2>gpurun_out/r02_train_g1.err; tail -c 600 gpurun_out/r02_train_g1.json
FSVIT_GCONV3X3=0 python bench.py --mode train --no-cpu-baseline > gpurun_out/r02_train_g0.json 2>/dev/null; tail -c 600 gpurun_out/r02_train_g0.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 7 --warmup 0 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 7 > $R/gpurun_out/r02_train7_kernel_stats.csv
head -30 $R/gpurun_out/r02_train7_kernel_stats.csv
