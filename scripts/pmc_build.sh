#!/bin/bash
# PMC counters of the cell-build kernels, one counter group per pass (GPU box only).
# usage: scripts/pmc_build.sh <outdir> <workloads...>
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pass$i" -- python3 scripts/exp_build.py "$@" > "$out/pass$i.log" 2>&1
done
python3 scripts/pmc_summary.py "$out"
