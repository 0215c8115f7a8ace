# PMC passes over the row kernels of the integrator's rebuilds on the equilibrated C3 fluid (one counter group per pass)
# usage: bash scripts/pmc_rows.sh "CHX_BUILD_SUB=0" "CHX_BUILD_SUB_U=2" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/pmc_rows/$v; mkdir -p $out
  export $v
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    EXP_WARM=20000 EXP_HALF_ROWS=0 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$out/pass$i" -- python3 scripts/exp_build.py c3 > "$out/pass$i.log" 2>&1
    echo "$v pass $i rc=$?"
  done
  unset CHX_BUILD_SUB CHX_BUILD_SUB_U
  python3 scripts/pmc_summary.py $out > $out/summary.txt
  echo "== $v"; grep -A 20 "k_nl_build_sub_fl\|k_nl_build_cell_fl<false>" $out/summary.txt | head -24
  find $out -name "*counter_collection*" -delete
done
