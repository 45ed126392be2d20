# Collects round-2 measurement evidence on the MI355X box into gpurun_out/ev2/ (tools/copy_evidence_r02.py -> profiles/).
# usage: round2_evidence.sh bench | prof | pmc      (three calls: each stays well under the 20-minute gpurun limit)
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/ev2
mkdir -p $E
cd $R
if [ "$1" = "bench" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $E/gpu_tests.log 2>&1; tail -1 $E/gpu_tests.log
timeout -k 10 300 python bench.py > $E/bench_c2_default.json 2> $E/bench_c2_default.err; cut -c1-220 $E/bench_c2_default.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench_c2_driver_cmd.json 2>/dev/null; cut -c1-220 $E/bench_c2_driver_cmd.json
timeout -k 10 300 python bench.py --workload C3 --steps 3356 --warmup 1678 --no-cpu-baseline > $E/bench_c3_f32.json 2>/dev/null; cut -c1-200 $E/bench_c3_f32.json
timeout -k 10 300 python bench.py --workload C3 --factor-dtype bf16 --steps 3356 --warmup 1678 --no-cpu-baseline > $E/bench_c3_bf16.json 2>/dev/null; cut -c1-200 $E/bench_c3_bf16.json
timeout -k 10 300 python bench.py --workload C4 --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > $E/bench_c4_single.json 2>/dev/null; cut -c1-200 $E/bench_c4_single.json
timeout -k 10 300 python bench.py --dp-mode native --no-cpu-baseline --steps 2098 --warmup 1049 > $E/bench_c2_dp_native_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c2_dp_native_one_rank.json
timeout -k 10 300 python bench.py --dp-mode shard --no-cpu-baseline --steps 2098 --warmup 1049 > $E/bench_c2_shard_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c2_shard_one_rank.json
timeout -k 10 300 python bench.py --workload C4 --dp-mode shard --no-cpu-baseline --steps 1000 --warmup 100 > $E/bench_c4_shard_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c4_shard_one_rank.json
timeout -k 10 300 python tools/bench_uvt.py > $E/uvt_pass_roofline.txt 2>&1; cat $E/uvt_pass_roofline.txt | cut -c1-200
timeout -k 10 300 python tools/bench_metrics.py > $E/metric_functions_c2.txt 2>&1; cat $E/metric_functions_c2.txt
timeout -k 10 300 python tools/bench_sizes.py > $E/step_period_by_size.txt 2>&1; cat $E/step_period_by_size.txt | cut -c1-200
timeout -k 10 300 python tools/diag_short_calls.py 20 > $E/short_call_breakdown.txt 2>&1; tail -5 $E/short_call_breakdown.txt
timeout -k 10 300 python tools/bench_forms_tiny.py > $E/tiny_problem_forms.txt 2>&1; tail -8 $E/tiny_problem_forms.txt
timeout -k 10 300 python tools/diag_common_path.py > $E/resident_common_path.txt 2>&1; timeout -k 10 300 python tools/diag_common_path.py C3 >> $E/resident_common_path.txt 2>&1; grep "B=" $E/resident_common_path.txt
timeout -k 10 300 python tools/exp_uvt_sustained.py > $E/uvt_pass_vs_load_history.txt 2>&1; grep -v amdgpu.ids $E/uvt_pass_vs_load_history.txt
timeout -k 10 300 python tools/exp_driver_call.py 20 > $E/driver_call_event_pair_cost.txt 2>&1; grep record $E/driver_call_event_pair_cost.txt
timeout -k 10 120 tools/microbench/valu_rate > $E/valu_issue_microbench.txt 2>&1; head -3 $E/valu_issue_microbench.txt
fi
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
if [ "$1" = "prof" ]; then
$P --stats -d $E/prof_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $E/bench_c2_driver_cmd_under_rocprof.json 2>/dev/null
$P --stats -d $E/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $E/bench_c2_under_rocprof.json 2>/dev/null
$P --stats -d $E/prof_c3 -- python3 $R/bench.py --workload C3 --steps 3356 --warmup 1678 --no-cpu-baseline --no-extras > /dev/null 2>&1
MFCD_SKIP_TORCH=1 UVT_BENCH_SECONDS=0.04 $P --stats -d $E/prof_uvt -- python3 $R/tools/bench_uvt.py C2 C3 C5 > /dev/null 2>&1
fi
if [ "$1" = "profuvt" ]; then
MFCD_SKIP_TORCH=1 UVT_BENCH_SECONDS=0.04 $P --stats -d $E/prof_uvt -- python3 $R/tools/bench_uvt.py C2 C3 C5 > /dev/null 2>&1
fi
if [ "$1" = "pmc" ]; then
B="python3 $R/bench.py --no-cpu-baseline --no-extras --clock-ramp 0 --steps 2098 --warmup 1049"
$P --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH -d $E/pmc_c2_a -- $B > /dev/null 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $E/pmc_c2_b -- $B > /dev/null 2>&1
$P --pmc FETCH_SIZE -d $E/pmc_c2_fetch -- $B > /dev/null 2>&1
$P --pmc WRITE_SIZE -d $E/pmc_c2_write -- $B > /dev/null 2>&1
B3="python3 $R/bench.py --workload C3 --no-cpu-baseline --no-extras --clock-ramp 0 --steps 3356 --warmup 1678"
$P --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH -d $E/pmc_c3_a -- $B3 > /dev/null 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $E/pmc_c3_b -- $B3 > /dev/null 2>&1
for S in C4 C5; do
  $P --pmc FETCH_SIZE -d $E/pmc_${S}_fetch -- python3 $R/tools/stream_step.py $S 40 > /dev/null 2>&1
  $P --pmc WRITE_SIZE -d $E/pmc_${S}_write -- python3 $R/tools/stream_step.py $S 40 > /dev/null 2>&1
  $P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $E/pmc_${S}_sq -- python3 $R/tools/stream_step.py $S 40 > /dev/null 2>&1
  $P --pmc TCC_HIT_sum TCC_MISS_sum -d $E/pmc_${S}_tcc -- python3 $R/tools/stream_step.py $S 40 > /dev/null 2>&1
done
cd $R
for tag in c2 c3; do python3 tools/pmc_summary.py $E/pmc_${tag}_a resident_train_kernel > $E/resident_pmc_${tag}.txt; python3 tools/pmc_summary.py $E/pmc_${tag}_b resident_train_kernel >> $E/resident_pmc_${tag}.txt; done
for S in C4 C5; do for k in fetch write sq tcc; do python3 tools/pmc_summary.py $E/pmc_${S}_$k train_step_kernel; done > $E/streaming_pmc_$S.txt; done
cat $E/resident_pmc_c2.txt $E/streaming_pmc_C5.txt
fi
find $E -name "*kernel_trace.csv" -size +6M -delete || true
find $E -name "*counter_collection.csv" -size +6M -delete || true
du -sh $E
