# uvt pass: parity tests, timing, in-kernel stamps (diagnostic build libmfcd_exp1.so = -DMFCD_UVT_STAMPS=1)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "uvt or metric or golden or e2e" > gpurun_out/uvt_tests.log 2>&1 || { tail -30 gpurun_out/uvt_tests.log; exit 1; }
tail -2 gpurun_out/uvt_tests.log
export MFCD_SKIP_TORCH=1
python3 tools/bench_uvt.py C2 C3 C5 2>&1 | grep uvt_stats | cut -c1-150
MFCD_LIB=$R/matrix-factorization-with-comparison-data_amd/libmfcd_exp1.so python3 tools/diag_uvt_stamps.py 2>&1 | grep tiles
