for s in 12800 6400 3200 1600 3200 12800; do FSVIT_STEM_SLICE=$s python bench.py --steps 20 --warmup 5 --no-legs --no-modes --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slice $s', round(d['value'],1), round(d['ms_per_step'],3))"; done
