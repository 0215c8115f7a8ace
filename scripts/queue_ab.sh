#!/bin/bash
# Fresh processes alternating between plain HIP launches (CHX_AQL=0) and the step queue (GPU box only):
#   scripts/queue_ab.sh [processes per mode] [workload] [steps] -> one line per process
n=${1:-12}; wl=${2:-c3}; steps=${3:-20000}
for i in $(seq 1 "$n"); do
  echo -n "hip   "; CHX_AQL=0 timeout -k 10 120 python scripts/steps_rate.py "$wl" "$steps" 3 2>/dev/null
  echo -n "queue "; timeout -k 10 120 python scripts/steps_rate.py "$wl" "$steps" 3 2>/dev/null
done
