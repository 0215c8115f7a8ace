#!/bin/bash
# How often does a fresh process land in the HIP runtime's slow launch regime?
# usage: scripts/regime_sweep.sh <processes per setting>     (GPU box only)
n=${1:-8}
for setting in "4 sleep" "2 sleep" "4 spin" "2 spin" "0 sleep"; do
  set -- $setting
  echo "== CHX_PIN_CPUS=$1 CHX_POLL=$2"
  for i in $(seq $n); do
    CHX_PIN_CPUS=$1 CHX_POLL=$2 python scripts/steps_rate.py c3 20000 3 2>/dev/null | tail -1
  done
done
