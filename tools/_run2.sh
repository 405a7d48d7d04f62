timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_visformer.py tests/test_gpu_deit.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -2
for n in bf16x2 f16x2 parity; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 6 --warmup 2 --layers 2>gpurun_out/x2f_layers_$n.txt | tail -1 > gpurun_out/x2f_bench_$n.json; done
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_modes.json 2>/dev/null
