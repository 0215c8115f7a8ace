# second sweep for the slow-process regime of a bare launch loop: CPU placement of the process's threads and the
# runtime's signal flags (GPU box only; eight fresh processes per setting, third of five samples each)
cd $GRAFT_REPO_ROOT/scripts/ub
hipcc --offload-arch=gfx950 -O3 -o /tmp/lmp launch_mode_probe.hip || exit 1
nproc
run() { echo "== $*"; for i in 1 2 3 4 5 6 7 8; do "$@" /tmp/lmp 50000 | sed -n 3p | sed -e 's/.*launches: //'; done; }
run env X=0
run taskset -c 0-1
run taskset -c 0-7
run env ROC_CPU_WAIT_FOR_SIGNAL=0
# run env ROC_SYSTEM_SCOPE_SIGNAL=0   (hangs: the probe never returns)
run env DEBUG_HIP_DYNAMIC_QUEUES=0
run nice -n 10
