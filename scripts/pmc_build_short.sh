export TMPDIR=/tmp
out=gpurun_out/pmc_build_r2b; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pass$i" -- python3 scripts/exp_build.py c4 > "$out/pass$i.log" 2>&1
  echo pass $i rc=$?
done
python3 scripts/pmc_summary.py $out > $out/summary.txt
find $out -name "*kernel_trace*" -delete
grep -A 14 "cellwise\|k_cell_scan" $out/summary.txt
