R=$PWD; mkdir -p gpurun_out/r05m
timeout 900 python -m pytest tests/test_gpu_transform.py -q -x > gpurun_out/r05m/tr.txt 2>&1; tail -3 gpurun_out/r05m/tr.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/e2e; rocprofv3 --kernel-trace -d /tmp/e2e -o kt -- python3 $R/tools/prof_e2e.py > $R/gpurun_out/r05m/e2e2.txt 2> $R/gpurun_out/r05m/e2e2.err; tail -1 $R/gpurun_out/r05m/e2e2.txt
python3 $R/tools/prof_e2e.py --gaps $(find /tmp/e2e -name "*.db" | head -1) | grep -E "last evaluate|transform"
python3 $R/tools/prof_e2e.py 2>/dev/null | tail -1
