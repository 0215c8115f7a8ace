# steps/s of the bench workloads as a function of the lanes-per-particle override
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
cd $GRAFT_REPO_ROOT
for wl in c3 c4; do
for g in 2 4 8 16; do
  echo "== $wl G=$g"
  CHX_FL_G=$g python bench.py --workload $wl --cpu-steps 0 --extra-c4 0 --steps 2000 --warmup 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['config'].get('list_rebuilds_in_timed_region'))"
done; done
