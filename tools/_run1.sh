timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_deit.py tests/test_gpu_visformer.py -x -q -m gpu 2>&1 | tail -2
python bench.py --model deit_small_patch16_224 --no-cpu-baseline --no-modes --steps 3 --warmup 1 --layers 2>gpurun_out/deit_layers_d.txt | tail -1 | cut -c1-130
grep "attn.core" gpurun_out/deit_layers_d.txt
python bench.py --no-cpu-baseline --no-modes --steps 20 --warmup 3 --layers 2>gpurun_out/vis_layers_d.txt | tail -1 | cut -c1-130
grep "attn" gpurun_out/vis_layers_d.txt
python bench.py --numerics bf16x2 --no-cpu-baseline --no-modes --steps 4 --warmup 2 --layers 2>gpurun_out/x2l_layers.txt | tail -1 | cut -c1-130
grep "attn.core" gpurun_out/x2l_layers.txt
