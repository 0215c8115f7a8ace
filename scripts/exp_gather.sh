set -e
cd $GRAFT_REPO_ROOT
for v in "-DCHX_EXP_NO_MATH" "-DCHX_EXP_NO_INDEX" "-DCHX_EXP_NO_INDEX -DCHX_EXP_SAME_LINE_GATHER" "-DCHX_EXP_NO_MATH -DCHX_EXP_SAME_LINE_GATHER"; do
  echo "== variant: $v"
  bash scripts/build_variant.sh $v
  GS=auto python scripts/kernel_bench.py c3 c4 2>&1 | grep -v amdgpu.ids
done
