# launch-to-launch floor of the step kernel (no force loop, no epilogue arithmetic, no control read) vs lanes per particle
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
set -e
cd $GRAFT_REPO_ROOT
for v in "-DCHX_EXP_NOFORCE -DCHX_EXP_NOEPI -DCHX_EXP_NOCTRL" ""; do
  echo "== variant: $v"
  bash scripts/build_variant.sh $v
  GS=1,2,4,8,16 python scripts/kernel_bench.py c3 2>&1 | grep -v amdgpu.ids
done
