for i in 1 2 3; do for v in few-shot-vit_amd/libfsvit.so ${1:-tools/probes/variants/libfsvit_pre.so}; do python tools/bench_variant.py $v --steps 20 --warmup 5 --no-legs --no-modes --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
