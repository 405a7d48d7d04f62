timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('train ms',d['ms_per_step'])"; done
python -c "
import __graft_entry__ as g
g.smoke()
" 2>&1 | tail -8
