# GPU test suite under the tuning overrides (step queue off / with HIP's fence scopes -- every other row runs with the
# default: steady-state packets without cache fences --, lanes per particle, pruning cadence, launch queue and its poll policy, build
# kernel / candidate lists / id slots / half rows, list carry-over and its device check, Monte Carlo inner list): every
# combination must pass (tests that exercise a feature the override switches off skip themselves)
# usage: bash scripts/test_matrix.sh [first] [count]     -> gpurun_out/test_matrix_<first>.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
first=${1:-0}; count=${2:-100}; k=0
log=gpurun_out/test_matrix_$first.log
: > $log
# (round 6: the queue's narrower modes -- steps only, gate at the first step, plain join, no overlap --, the parameter-cache
# switch and the lane counts 16 / 64 of the fused kernels are gone with their switches; what is left is what a user or a
# maintainer can still reach)
for env in "CHX_AQL=0" "CHX_AQL_FENCE=agent" "CHX_QUEUE_DEBUG=250" "CHX_AQL=0 CHX_LAUNCH_WINDOW=3" "CHX_SYNC_REBUILD=1" "CHX_LAUNCH_WINDOW=1" \
           "CHX_LAUNCH_WINDOW=500" "CHX_FL_G=1" "CHX_FL_G=2 CHX_LAUNCH_WINDOW=3" "CHX_FL_G=8" "CHX_PRUNE_EVERY=0" "CHX_PRUNE_EVERY=1" \
           "CHX_PRUNE_EVERY=7 CHX_INNER_SKIN=0.02" "CHX_PRUNE_EVERY=50 CHX_INNER_SKIN=0.3" "CHX_POLL=spin" "CHX_PIN_CPUS=0" \
           "CHX_LIST_CARRY=0" "CHX_LIST_CARRY_MAX_STEPS=100000" "CHX_BUILD_KERNEL=cell" "CHX_BUILD_LISTS=0" "CHX_BUILD_SUB=0" \
           "CHX_RUN_HALF_ROWS=1 CHX_BUILD_SLOTS=0" "CHX_MC_INNER_SKIN=0" "CHX_CARRY_VALIDATE=0"; do
  k=$((k + 1))
  if [ $k -le $first ] || [ $k -gt $((first + count)) ]; then continue; fi
  echo "== $env" | tee -a $log
  env $env timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/test_matrix_last.log 2>&1
  tail -1 gpurun_out/test_matrix_last.log | tee -a $log
  grep -q "failed" gpurun_out/test_matrix_last.log && grep -n "^E \|^FAILED\|Error" gpurun_out/test_matrix_last.log | head -20 | tee -a $log
done
