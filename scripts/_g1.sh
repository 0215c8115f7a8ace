cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mc or MC or mcmc" > gpurun_out/t_mc.log 2>&1; tail -5 gpurun_out/t_mc.log; grep -n "^E " gpurun_out/t_mc.log | head -20
for s in 0 0.1 0.05; do echo "inner skin $s"; CHX_MC_INNER_SKIN=$s timeout -k 10 200 python scripts/mc_rate.py 2>&1 | tail -3; done
