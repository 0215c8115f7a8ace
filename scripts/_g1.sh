cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2 3; do for h in 0 1; do CHX_BUILD_SLOTS=$h timeout -k 10 200 python scripts/run_phases.py 20 c3 0 > gpurun_out/phases_s$h.$r.log 2>&1; echo "slots=$h $(grep 'device idle' gpurun_out/phases_s$h.$r.log) $(grep 'kick done' gpurun_out/phases_s$h.$r.log)"; done; done
