timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "not uvt" > gpurun_out/t17_tests.log 2>&1; rc=$?; tail -2 gpurun_out/t17_tests.log
test $rc -eq 0 || { grep -n "Error\|assert\|FAILED" gpurun_out/t17_tests.log | head -20; exit 1; }
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline']['kernel_us_event_pairs']['avg'], d['roofline']['launch_period_us'])"; done
