# kernel-only time of a resident launch of K steps (mfcd_train_steps_timed), K = 3 .. 160: intercept and slope
for K in 3 5 10 20 40 80 160; do python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-extras --clock-ramp 0.1 > gpurun_out/r3_sweepK_$K.json 2>> gpurun_out/r3_sweepK.err; python -c "
import json,sys
d=json.load(open('gpurun_out/r3_sweepK_$K.json'))
print($K, d['ms_per_step']*1e3*$K, d['roofline'].get('kernel_only'))
"; done
