# what bounds the list kernels: variants without the pair arithmetic / without the position gathers (GPU box)
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
set -e
cd $GRAFT_REPO_ROOT
for v in "" "-DCHX_EXP_NO_MATH" "-DCHX_EXP_NO_GATHER" "-DCHX_EXP_NO_MATH -DCHX_EXP_NO_GATHER"; do
  echo "== variant: $v"
  bash scripts/build_variant.sh $v
  GS=auto python scripts/kernel_bench.py c3 c4 2>&1 | grep -v amdgpu.ids
  GS=1,4 python scripts/kernel_bench.py c4 2>&1 | grep -v amdgpu.ids
done
