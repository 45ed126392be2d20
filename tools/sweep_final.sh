#!/bin/bash
# Diagnostic: look-ahead depth after the SGPR-spill fix (not the product).
for rep in 1 2; do for w in 0 4 8; do
  echo "rep=$rep LOOK=$w: $(python bench.py --no-extras --tune resident_lookahead=$w --no-cpu-baseline --steps 5245 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print(round(d["value"]/1e6,2), "M upd/s", round(d["ms_per_step"]*1e3,3), "us/step", d["roofline"]["launch_period_us"])')"
done; done
