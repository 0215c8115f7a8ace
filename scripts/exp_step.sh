# decomposition of the fused step kernel's launch-to-launch time (GPU box): each variant removes one part
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
set -e
cd $GRAFT_REPO_ROOT
for v in "" "-DCHX_EXP_NOCTRL" "-DCHX_EXP_NOFORCE" "-DCHX_EXP_NOEPI" "-DCHX_EXP_NOFORCE -DCHX_EXP_NOEPI" "-DCHX_EXP_NOFORCE -DCHX_EXP_NOEPI -DCHX_EXP_NOCTRL"; do
  echo "== variant: $v"
  bash scripts/build_variant.sh $v
  GS=auto python scripts/kernel_bench.py c3 c4 2>&1 | grep -v amdgpu.ids
done
