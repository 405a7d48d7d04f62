timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_limb" -s 2>&1 | grep -E "grouped|passed|failed" | tail -5
timeout 600 python -m pytest tests/test_gpu_visformer.py -x -q -m gpu -k "two_limb" -s 2>&1 | grep -E "dlogit|passed|failed" | cut -c1-120
for n in bf16x2 f16x2; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 4 --warmup 2 --layers 2>gpurun_out/x2i_layers_$n.txt | tail -1 | cut -c1-200; done
