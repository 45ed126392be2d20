#!/bin/bash
# Diagnostic sweep of the resident kernel's slice size / wave count at C2 (not the product).
for cfg in "4 8" "8 8" "16 8" "2 16" "4 16" "1 16"; do
  set -- $cfg
  echo "Q=$1 WPC=$2: $(python bench.py --no-extras --tune resident_q=$1 --tune resident_wpc=$2 --no-cpu-baseline --steps 5245 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print(d["value"], "upd/s", d["ms_per_step"]*1e3, "us/step")')"
done
