timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_limb or conv_gemm or gemm256 or soak" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_visformer.py tests/test_gpu_deit.py -x -q -m gpu -k "two_limb or x2" -s 2>&1 | grep -E "dlogit|dfeat|passed|failed" | cut -c1-110
for n in bf16x2 parity; do python bench.py --numerics $n --no-cpu-baseline --no-modes --steps 6 --warmup 2 --layers 2>gpurun_out/x2j_layers_$n.txt | tail -1 | cut -c1-200; done
