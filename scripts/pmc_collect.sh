#!/bin/bash
# Collect PMC counters for the hot kernels, one counter group per pass (GPU box only; --pmc never together with a
# trace domain other than --kernel-trace).
# usage: scripts/pmc_collect.sh <outdir> <workloads...>
# The kernels are profiled on the fluid EQUILIBRATED for $KB_WARM steps (default 20000 = 20 ps) with a list built
# from that state -- the state bench.py quotes its roofline entries on.  The equilibration runs once, unprofiled, and
# leaves a snapshot every pass loads; the pass logs carry the state's P_nl (scripts/pmc_to_json.py stores it, bench.py
# withholds `traffic` when its own list differs by more than 2 %).  Only the reduced tables are kept: the raw pass
# directories are deleted (gpurun copies back at most 64 MiB).
export TMPDIR=/tmp
export KB_WARM=${KB_WARM:-20000}
out=$1; shift
mkdir -p "$out"
export KB_SNAPSHOT_DIR=/tmp/kb_snapshots_$$
echo "{\"equilibration_steps\": $KB_WARM}" > "$out/meta.json"
GS=auto KB_REPS=2 python3 scripts/kernel_bench.py "$@" > "$out/equilibrate.log" 2>&1 || exit 1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "TA_BUSY_avr GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  GS=auto timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pass$i" -- python3 scripts/kernel_bench.py "$@" > "$out/pass$i.log" 2>&1 || exit 1
done
python3 scripts/pmc_summary.py "$out" > "$out/summary.txt"
python3 scripts/pmc_to_json.py "$out" "$out/pmc_traffic.json" "$out/pmc_hot_kernels.md" > /dev/null || exit 1
rm -rf "$out"/pass[0-9] "$KB_SNAPSHOT_DIR"
