R=$PWD
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -2
python bench.py --mode train --no-cpu-baseline > gpurun_out/r02_train_h3.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r02_train_h3.json'));print(d['ms_per_step'],d['value'],d['final_loss'])"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 7 --warmup 0 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 7 > $R/gpurun_out/r02_train10_kernel_stats.csv
grep -E "proto_head|elementwise|scale_copy" $R/gpurun_out/r02_train10_kernel_stats.csv | head
