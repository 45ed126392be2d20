# PMC passes over the resident step kernel (bench.py, 2 epochs timed): instruction mix and wave-time split
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/respmc
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/respmc/a -- python3 $R/bench.py --no-cpu-baseline --steps 2098 --warmup 1049 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/respmc/b -- python3 $R/bench.py --no-cpu-baseline --steps 2098 --warmup 1049 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/respmc resident_train_kernel
