# A/B of the row kernels of the integrator's rebuilds (CHX_BUILD_SUB=0: one wavefront per particle; default: sixteen lanes
# per particle behind a filter) on the equilibrated fluid: event time of the whole build chain, then per-kernel averages
# under rocprofv3.   usage: bash scripts/sub_ab.sh [workload ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/sub_ab
for wl in ${@:-c3}; do
for rep in 1 2; do
for v in "CHX_BUILD_SUB=0" "CHX_BUILD_SUB=1"; do
  env CHX_AQL=0 $v EXP_WARM=20000 EXP_HALF_ROWS=0 python3 scripts/exp_build.py $wl 2>&1 | grep "us per call" | sed "s/^/$v /"
done; done
for v in "CHX_BUILD_SUB=0" "CHX_BUILD_SUB=1"; do
  export $v
  EXP_WARM=20000 EXP_HALF_ROWS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sub_ab/$wl.$v -- python3 scripts/exp_build.py $wl > gpurun_out/sub_ab/$wl.$v.log 2>&1
  echo "== $wl $v"; python3 scripts/kernel_stats.py gpurun_out/sub_ab/$wl.$v 2>/dev/null | grep "k_nl_build\|k_cell_\|k_build_epi"
done
done
find gpurun_out/sub_ab -name "*kernel_trace*" -delete
