#!/bin/bash
# pruned inner list: refresh cadence x inner skin, steps/s on the live loop (GPU box only)
# usage: scripts/inner_sweep.sh <workload> <steps per window>
wl=${1:-c4}; st=${2:-6000}
for every in 24 32 48 64; do for skin in 0.08 0.10 0.12 0.15; do
  echo -n "every=$every skin=$skin  "
  CHX_PRUNE_EVERY=$every CHX_INNER_SKIN=$skin python scripts/steps_rate.py $wl $st 3 2>/dev/null | tail -1
done; done
