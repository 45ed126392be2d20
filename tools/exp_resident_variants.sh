# Timing experiments on the resident kernel's look-ahead loop (C2): variants built with `make -C csrc exp NAME=.. EXPFLAGS=..`
P=$PWD/matrix-factorization-with-comparison-data_amd
show() { python -c "
import json,sys
d=json.load(open('$1'))
print('$1', d['value'], d['ms_per_step'], d['roofline'].get('kernel_only'))
"; }
for la in 2 3 4 6 8 12 16; do python bench.py --no-cpu-baseline --no-extras --tune resident_lookahead=$la > gpurun_out/r3_b7_la$la.json 2>> gpurun_out/r3_b7.err; show gpurun_out/r3_b7_la$la.json; done
for v in nopf poll1 sleep8 sleep32; do [ -f $P/libmfcd_hip_$v.so ] || continue; MFCD_LIB=$P/libmfcd_hip_$v.so python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3_b7_$v.json 2>> gpurun_out/r3_b7.err; show gpurun_out/r3_b7_$v.json; done
MFCD_LIB=$P/libmfcd_hip_poll1.so python bench.py --no-cpu-baseline --no-extras --tune resident_lookahead=8 > gpurun_out/r3_b7_poll1_la8.json 2>> gpurun_out/r3_b7.err; show gpurun_out/r3_b7_poll1_la8.json
MFCD_LIB=$P/libmfcd_hip_diag.so python tools/diag_resident_stats.py resident_lookahead=8 > gpurun_out/r3_stats5_la8.txt 2>&1; cat gpurun_out/r3_stats5_la8.txt
