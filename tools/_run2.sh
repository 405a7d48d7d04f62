timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_limb or conv_gemm" -s 2>&1 | grep -E "g256|passed|failed|Error" | tail -20
timeout 600 python -m pytest tests/test_gpu_visformer.py tests/test_gpu_deit.py -x -q -m gpu -k "two_limb or x2 or golden" -s 2>&1 | grep -E "dlogit|dfeat|passed|failed"
for n in bf16x2 f16x2; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 4 --warmup 2 --layers 2>gpurun_out/x2g_layers_$n.txt | tail -1 | cut -c1-200; done
