# Collects round-3 measurement evidence on the MI355X box into gpurun_out/ev3/ (tools/copy_evidence_r03.py -> profiles/).
# usage: round3_evidence.sh bench | prof | pmc | fuzz      (three calls: each stays well under the 20-minute gpurun limit)
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/ev3
mkdir -p $E
cd $R
if [ "$1" = "bench" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $E/gpu_tests.log 2>&1; tail -1 $E/gpu_tests.log
timeout -k 10 400 python bench.py > $E/bench_c2_default.json 2> $E/bench_c2_default.err; cut -c1-220 $E/bench_c2_default.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench_c2_driver_cmd.json 2>/dev/null; cut -c1-220 $E/bench_c2_driver_cmd.json
timeout -k 10 300 python bench.py --workload C3 --steps 3356 --warmup 1678 --no-cpu-baseline > $E/bench_c3_f32.json 2>/dev/null; cut -c1-200 $E/bench_c3_f32.json
timeout -k 10 300 python bench.py --workload C3 --factor-dtype bf16 --steps 3356 --warmup 1678 --no-cpu-baseline > $E/bench_c3_bf16.json 2>/dev/null; cut -c1-200 $E/bench_c3_bf16.json
timeout -k 10 300 python bench.py --workload C4 --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > $E/bench_c4_single.json 2>/dev/null; cut -c1-200 $E/bench_c4_single.json
timeout -k 10 300 python bench.py --dp-mode native --no-cpu-baseline --steps 2098 --warmup 1049 > $E/bench_c2_dp_native_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c2_dp_native_one_rank.json
timeout -k 10 300 python bench.py --dp-mode shard --no-cpu-baseline --steps 2098 --warmup 1049 > $E/bench_c2_shard_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c2_shard_one_rank.json
timeout -k 10 300 python bench.py --workload C4 --dp-mode shard --no-cpu-baseline --no-extras --steps 2000 --warmup 200 > $E/bench_c4_shard_one_rank.json 2>/dev/null; cut -c1-200 $E/bench_c4_shard_one_rank.json
timeout -k 10 300 python bench.py --workload C4 --dp-mode shard --tune shard_pipeline=0 --no-cpu-baseline --no-extras --steps 2000 --warmup 200 > $E/bench_c4_shard_one_rank_strict.json 2>/dev/null; cut -c1-200 $E/bench_c4_shard_one_rank_strict.json
timeout -k 10 300 python tools/bench_samplers.py > $E/samplers.txt 2>&1; tail -6 $E/samplers.txt
timeout -k 10 300 python tools/bench_big.py 2000 > $E/big_resident.txt 2>&1; tail -4 $E/big_resident.txt
timeout -k 10 300 python tools/exp_chain_depth.py > $E/chain_depth_experiment.txt 2>&1; tail -3 $E/chain_depth_experiment.txt
timeout -k 10 300 python tools/bench_uvt.py > $E/uvt_pass_roofline.txt 2>&1; cat $E/uvt_pass_roofline.txt | cut -c1-240
timeout -k 10 300 python tools/bench_metrics.py > $E/metric_functions_c2.txt 2>&1; cat $E/metric_functions_c2.txt
timeout -k 10 400 python tools/bench_metrics_c5.py > $E/metric_functions_c5.txt 2>&1; tail -6 $E/metric_functions_c5.txt
timeout -k 10 300 python tools/diag_short_calls.py 20 > $E/short_call_breakdown.txt 2>&1; tail -5 $E/short_call_breakdown.txt
timeout -k 10 300 bash tools/exp_short_call_sweep.sh >> $E/short_call_breakdown.txt 2>&1; tail -7 $E/short_call_breakdown.txt
timeout -k 10 300 python tools/diag_common_path.py > $E/resident_common_path.txt 2>&1; timeout -k 10 300 python tools/diag_common_path.py C3 >> $E/resident_common_path.txt 2>&1; grep "B=" $E/resident_common_path.txt
MFCD_LIB=$R/matrix-factorization-with-comparison-data_amd/libmfcd_hip_diag.so timeout -k 10 300 python tools/diag_resident_stats.py > $E/resident_wave_accounting.txt 2>&1; head -8 $E/resident_wave_accounting.txt
MFCD_LIB=$R/matrix-factorization-with-comparison-data_amd/libmfcd_hip_stamps.so timeout -k 10 300 python tools/diag_short_call_stamps.py 20 > $E/short_call_wave_timeline.txt 2>&1; MFCD_LIB=$R/matrix-factorization-with-comparison-data_amd/libmfcd_hip_stamps.so timeout -k 10 300 python tools/diag_short_call_stamps.py 100 >> $E/short_call_wave_timeline.txt 2>&1; python tools/sim_chain_depth.py >> $E/short_call_wave_timeline.txt 2>&1; tail -12 $E/short_call_wave_timeline.txt
timeout -k 10 300 python tools/bench_forms_tiny.py > $E/tiny_problem_forms.txt 2>&1; tail -8 $E/tiny_problem_forms.txt
fi
if [ "$1" = "fuzz" ]; then
timeout -k 10 500 python tools/fuzz_parity.py 1500 31 > $E/fuzz_parity.txt 2>&1; tail -3 $E/fuzz_parity.txt
timeout -k 10 300 python tools/fuzz_bf16.py 600 32 > $E/fuzz_bf16.txt 2>&1; tail -3 $E/fuzz_bf16.txt
timeout -k 10 300 python tools/fuzz_multi.py 1000 33 > $E/fuzz_multi_gpu_rehearsal.txt 2>&1; tail -3 $E/fuzz_multi_gpu_rehearsal.txt
timeout -k 10 300 python tools/fuzz_uvt.py 800 34 > $E/fuzz_uvt.txt 2>&1; tail -3 $E/fuzz_uvt.txt
fi
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
if [ "$1" = "prof" ]; then
$P --stats -d $E/prof_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $E/bench_c2_driver_cmd_under_rocprof.json 2>/dev/null
$P --stats -d $E/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $E/bench_c2_under_rocprof.json 2>/dev/null
$P --stats -d $E/prof_c3 -- python3 $R/bench.py --workload C3 --steps 3356 --warmup 1678 --no-cpu-baseline --no-extras > /dev/null 2>&1
$P --stats -d $E/prof_c4 -- python3 $R/bench.py --workload C4 --steps 4000 --warmup 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
MFCD_SKIP_TORCH=1 UVT_BENCH_SECONDS=0.01 $P --stats -d $E/prof_uvt -- python3 $R/tools/bench_uvt.py C2 C3 C5 > /dev/null 2>&1
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
U="python3 $R/tools/bench_uvt.py C3 C5"
MFCD_SKIP_TORCH=1 UVT_BENCH_SECONDS=0.04 $P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $E/pmc_uvt_sq -- $U > /dev/null 2>&1
MFCD_SKIP_TORCH=1 UVT_BENCH_SECONDS=0.04 $P --pmc FETCH_SIZE -d $E/pmc_uvt_fetch -- $U > /dev/null 2>&1
cd $R
for tag in c2 c3; do python3 tools/pmc_summary.py $E/pmc_${tag}_a resident_train_kernel > $E/resident_pmc_${tag}.txt; python3 tools/pmc_summary.py $E/pmc_${tag}_b resident_train_kernel >> $E/resident_pmc_${tag}.txt; done
python3 tools/pmc_summary.py $E/pmc_uvt_sq uvt_tiled_kernel > $E/uvt_pmc.txt; python3 tools/pmc_summary.py $E/pmc_uvt_fetch uvt_tiled_kernel >> $E/uvt_pmc.txt
cat $E/resident_pmc_c2.txt $E/uvt_pmc.txt
fi
find $E -name "*kernel_trace.csv" -size +6M -not -path "*prof_uvt*" -delete || true
find $E -name "*counter_collection.csv" -size +6M -delete || true
du -sh $E
