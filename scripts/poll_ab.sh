#!/bin/bash
# sleeping against spinning launch window, fresh processes alternating on one box (GPU box only)
n=${1:-20}
for i in $(seq $n); do
  for pol in sleep spin; do
    echo -n "$pol " ; CHX_POLL=$pol python scripts/steps_rate.py c3 20000 3 2>/dev/null | tail -1
  done
done
