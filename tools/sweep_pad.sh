#!/bin/bash
# Diagnostic: effect of capping workgroups per CU (unused LDS allocation) on the resident kernel at C2.
for rep in 1 2; do for pad in 0 57344 40960; do
  echo "rep=$rep LDS_PAD=$pad: $(python bench.py --no-extras --tune resident_lds_pad=$pad --no-cpu-baseline --steps 5245 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print(round(d["value"]/1e6,2), "M upd/s", round(d["ms_per_step"]*1e3,3), "us/step")')"
done; done
