# Collects the round's measurement evidence on the MI355X box into gpurun_out/evidence/ (copied to profiles/ afterwards).
set -e
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence
rm -rf $E && mkdir -p $E
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $E/gpu_tests.log 2>&1 || { tail -30 $E/gpu_tests.log; exit 1; }
tail -1 $E/gpu_tests.log
timeout -k 10 300 python bench.py > $E/bench_c2_final_run.json 2> $E/bench_c2_final_run.err
cut -c1-400 $E/bench_c2_final_run.json
timeout -k 10 300 python bench.py --dp-mode native --no-cpu-baseline > $E/bench_c2_dp_native_one_rank.json 2>/dev/null
cut -c1-200 $E/bench_c2_dp_native_one_rank.json
timeout -k 10 600 python bench.py --workload C4 --no-cpu-baseline --steps 2000 --warmup 200 > $E/bench_c4_single.json 2>/dev/null
timeout -k 10 600 python bench.py --workload C4 --dp-mode native --no-cpu-baseline --steps 2000 --warmup 200 > $E/bench_c4_dp_native_one_rank.json 2>/dev/null
cut -c1-160 $E/bench_c4_single.json $E/bench_c4_dp_native_one_rank.json
timeout -k 10 300 python tools/bench_uvt.py > $E/uvt_pass_roofline.txt 2>&1
cat $E/uvt_pass_roofline.txt | cut -c1-230
timeout -k 10 300 python tools/bench_metrics.py > $E/metric_functions_c2.txt 2>&1 || true
cat $E/metric_functions_c2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $E/bench_c2_under_rocprof.json 2>/dev/null
MFCD_SKIP_TORCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_uvt -- python3 $R/tools/bench_uvt.py C2 C3 C5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_dp -- python3 $R/bench.py --dp-mode native --no-cpu-baseline --steps 2098 --warmup 1049 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $E/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2098 --warmup 1049 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $E/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2098 --warmup 1049 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $E/pmc_fetch $E/pmc_write $E/pmc_traffic.json | tail -12
find $E -name "*kernel_stats.csv" | head
# keep only the stats summaries (the traces are large)
find $E -name "*kernel_trace.csv" -size +8M -delete || true
