timeout 900 python -m pytest tests/test_gpu_visformer.py -x -q -m gpu -k "image_size" -s 2>&1 | grep -E "64 x 64|passed|failed|Error" | tail -5
FSVIT_STAGE1_RING=1 timeout 900 python -m pytest tests/test_gpu_visformer.py tests/test_gpu_soak.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -2
