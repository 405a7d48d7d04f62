R=$PWD
python bench.py > gpurun_out/r02_bench_final.json 2>gpurun_out/r02_bench_final.err; tail -c 300 gpurun_out/r02_bench_final.json
python bench.py --mode train > gpurun_out/r02_train_bench_final.json 2>/dev/null; tail -c 200 gpurun_out/r02_train_bench_final.json
for n in bf16x2 f16x2 parity; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 4 --warmup 2 --layers 2>gpurun_out/x2f_layers_$n.txt | tail -1 > gpurun_out/x2f_bench_$n.json; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 7 --warmup 0 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 7 > $R/gpurun_out/r02_train_final_kernel_stats.csv
rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $R/bench.py --numerics bf16x2 --no-cpu-baseline --no-modes --no-roofline --steps 4 --warmup 0 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 4 > $R/gpurun_out/r02_x2_kernel_stats.csv
