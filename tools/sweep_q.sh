#!/bin/bash
# Diagnostic: slice size / waves per CU at C2 (not the product).
for rep in 1 2; do for cfg in "4 8" "2 16" "4 16"; do set -- $cfg
  echo "rep=$rep Q=$1 WPC=$2: $(python bench.py --no-extras --tune resident_q=$1 --tune resident_wpc=$2 --no-cpu-baseline --steps 5245 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print(round(d["value"]/1e6,2), "M upd/s", round(d["ms_per_step"]*1e3,3), "us/step")')"
done; done
