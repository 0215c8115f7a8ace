cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== loop: $(timeout -k 10 300 python scripts/_diag_ms.py 300 2>&1 | grep 'mismatching runs')"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log; grep -n "^E " gpurun_out/t_all.log | head -20
for c in 1 0; do timeout -k 10 200 python scripts/run_phases.py 20 c3 $c > gpurun_out/phases_c$c.log 2>&1; echo "carry=$c $(grep 'device idle' gpurun_out/phases_c$c.log) $(grep 'first steps enq' gpurun_out/phases_c$c.log)"; done
