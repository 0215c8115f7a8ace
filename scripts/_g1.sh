cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; tail -5 gpurun_out/t_all.log
CHX_LAUNCH_WINDOW=500 timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t_w500.log 2>&1; tail -40 gpurun_out/t_w500.log | grep -n "^E \|FAILED\|passed\|failed"
CHX_MC_DECIDE_LAUNCH=1 timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t_mcd.log 2>&1; tail -40 gpurun_out/t_mcd.log | grep -n "^E \|FAILED\|passed\|failed"
for h in 0 1; do CHX_RUN_HALF_ROWS=$h timeout -k 10 200 python scripts/run_phases.py 20 c3 0 > gpurun_out/phases_h$h.log 2>&1; grep "device idle\|run() returned" gpurun_out/phases_h$h.log; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_h0.log 2>&1; tail -1 gpurun_out/bench_h0.log | cut -c1-400
