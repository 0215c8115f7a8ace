# A/B on one box: scripts/ab.sh VAR "cmd" reps   -> alternates VAR=0 / VAR=1
cd $GRAFT_REPO_ROOT
VAR=$1; CMD=$2; REPS=${3:-3}
for i in $(seq $REPS); do for v in 0 1; do echo -n "$VAR=$v: "; env $VAR=$v $CMD 2>&1 | grep -v amdgpu.ids | head -1; done; done
