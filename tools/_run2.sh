python bench.py --model deit_small_patch16_224 --no-cpu-baseline --no-modes --steps 3 --warmup 1 --layers 2>gpurun_out/deit_layers_a.txt | tail -1 | cut -c1-160
FSVIT_GEMM256_MIN_AI=100 python bench.py --model deit_small_patch16_224 --no-cpu-baseline --no-modes --steps 3 --warmup 1 --layers 2>gpurun_out/deit_layers_b.txt | tail -1 | cut -c1-160
