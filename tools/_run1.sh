R=$PWD
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_token_label.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('train ms',d['ms_per_step'])"; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 7 --warmup 0 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 7 > $R/gpurun_out/r02_train13_kernel_stats.csv
grep -E "finalize|batch_sum" $R/gpurun_out/r02_train13_kernel_stats.csv
python -c "
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/r02_train13_kernel_stats.csv')))
print('kernel sum', sum(float(r['ms_per_step']) for r in rows))"
