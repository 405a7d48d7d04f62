export FSVIT_STAGE1_RING=1
for v in ilp maxilp noslp u2 u1; do echo $v; python tools/bench_stage1.py 12800 tools/probes/variants/libfsvit_$v.so; done
python tools/bench_stage1.py 12800
