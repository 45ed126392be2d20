# PMC passes over the UV^T pass (one config per call): clock, MFMA busy, wave stall split
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-C5}
cd /tmp && export TMPDIR=/tmp
export MFCD_SKIP_TORCH=1
rm -rf $R/gpurun_out/uvtpmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/uvtpmc/a -- python3 $R/tools/bench_uvt.py $CFG > $R/gpurun_out/uvtpmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/uvtpmc/b -- python3 $R/tools/bench_uvt.py $CFG > $R/gpurun_out/uvtpmc_b.log 2>&1
ls -R $R/gpurun_out/uvtpmc | head -30
