# interleaved fresh processes of the bare launch probe: default kernarg pool vs HSA_KERNARG_POOL_SIZE (GPU box only)
cd $GRAFT_REPO_ROOT/scripts/ub
hipcc --offload-arch=gfx950 -O3 -o /tmp/lmp launch_mode_probe.hip || exit 1
for i in $(seq 1 ${1:-12}); do
  for e in "X=0" "HSA_KERNARG_POOL_SIZE=16777216" "HSA_KERNARG_POOL_SIZE=67108864"; do
    echo "$e: $(env $e /tmp/lmp 50000 | sed -n 3p | sed -e 's/.*launches: //')"
  done
done
