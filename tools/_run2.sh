for n in bf16x2 f16x2 parity; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 6 --warmup 2 --layers 2>gpurun_out/x2f_layers_$n.txt | tail -1 > gpurun_out/x2f_bench_$n.json; done
python bench.py > gpurun_out/r02_bench_final.json 2>/dev/null
python bench.py --model deit_small_patch16_224 --numerics bf16x2 --no-cpu-baseline --no-modes --no-roofline --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/deit_x2.json
python bench.py --model deit_small_patch16_224 --numerics parity --no-cpu-baseline --no-modes --no-roofline --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/deit_parity.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kprof
rocprofv3 --kernel-trace -d /tmp/kprof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --numerics bf16x2 --no-cpu-baseline --no-modes --no-roofline --steps 4 --warmup 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(ls /tmp/kprof/*.db | head -1) 4 > $GRAFT_REPO_ROOT/gpurun_out/r02_x2_kernel_stats.csv
