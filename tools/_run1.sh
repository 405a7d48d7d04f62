export FSVIT_STAGE1_RING=1
python tools/bench_stage1.py 12800
python tools/bench_stage1.py 12800 tools/probes/variants/libfsvit_nogelu.so
python tools/bench_stage1.py 12800 tools/probes/variants/libfsvit_nomask.so
python tools/bench_stage1.py 12800 tools/probes/variants/libfsvit_nogelumask.so
