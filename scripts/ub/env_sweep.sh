# Does any HIP runtime environment flag move a bare launch loop out of (or into) the slow-process regime?
# (GPU box only; scripts/ub/launch_mode_probe.hip, six fresh processes per setting, third of five samples each)
cd $GRAFT_REPO_ROOT/scripts/ub
hipcc --offload-arch=gfx950 -O3 -o /tmp/lmp launch_mode_probe.hip || exit 1
for e in "X=0" "DEBUG_CLR_MAX_BATCH_SIZE=8" "DEBUG_CLR_MAX_BATCH_SIZE=100000" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=100000" \
         "HSA_KERNARG_POOL_SIZE=16777216" "DEBUG_HIP_KERNARG_COPY_OPT=0" "AMD_OPT_FLUSH=0" "ROC_AQL_QUEUE_SIZE=256"; do
  echo "== $e"
  for i in 1 2 3 4 5 6; do
    env $e /tmp/lmp 50000 | sed -n 3p | sed -e 's/.*launches: //'
  done
done
