# uvt pass: parity tests, kernel trace, and the diagnostic EXP builds (no X loads / no epilogue sums / no MFMA)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "uvt or metric or golden or e2e" > gpurun_out/uvt_tests.log 2>&1 || { tail -30 gpurun_out/uvt_tests.log; exit 1; }
tail -2 gpurun_out/uvt_tests.log
cd /tmp && export TMPDIR=/tmp
export MFCD_SKIP_TORCH=1
rm -rf $R/gpurun_out/uvtprof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/uvtprof -- python3 $R/tools/bench_uvt.py C2 C3 C5 > $R/gpurun_out/uvtprof.log 2>&1
grep uvt_stats $R/gpurun_out/uvtprof.log | cut -c1-150
for e in 1 2 3; do echo "EXP $e"; MFCD_LIB=$R/matrix-factorization-with-comparison-data_amd/libmfcd_exp$e.so python3 $R/tools/bench_uvt.py C2 C3 C5 2>&1 | grep uvt_stats | cut -c1-110; done
