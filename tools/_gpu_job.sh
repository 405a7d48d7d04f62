mkdir -p gpurun_out/r05b
timeout 600 python -m pytest tests/test_gpu_mlp_train.py -x -q > gpurun_out/r05b/ops.txt 2>&1; tail -5 gpurun_out/r05b/ops.txt
python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r05b/train_fused.json 2> gpurun_out/r05b/train_fused.err
FSVIT_MLP_TRAIN_FUSED=0 python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r05b/train_unfused.json 2> gpurun_out/r05b/train_unfused.err
python - <<'P'
import json
for n in ("fused","unfused"):
    try: print(n, json.load(open("gpurun_out/r05b/train_%s.json"%n))["ms_per_step"])
    except Exception as e: print(n, e)
P
timeout 1500 python -m pytest tests/test_gpu_train.py -q > gpurun_out/r05b/train_tests.txt 2>&1; tail -8 gpurun_out/r05b/train_tests.txt
timeout 900 python -m pytest tests/test_gpu_unfused_paths.py -x -q -k "training_dispatch or weight_rounding" -s > gpurun_out/r05b/switch_tests.txt 2>&1; tail -12 gpurun_out/r05b/switch_tests.txt
bash tools/prof_train_serial.sh r05b_train_serial > gpurun_out/r05b/prof.log 2>&1; tail -3 gpurun_out/r05b/prof.log
