#!/bin/bash
# Is the run-to-run spread of the C3 figure (137 k .. 164 k steps/s on one box) a clock state?  (GPU box only.)
# Runs the bench several times with a long timed region and samples rocm-smi clocks in the middle of each.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
  python bench.py --workload c3 --cpu-steps 0 --extra-c4 0 --steps 400000 --warmup 2000 --steady-steps 0 --kernel-reps 64 > /tmp/b_$rep.log 2>/dev/null &
  pid=$!
  sleep 4.0
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk" | head -8 > /tmp/clk_$rep.txt
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2 >> /tmp/clk_$rep.txt
  wait $pid
  python - <<PY
import json
for l in open("/tmp/b_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); print("rep $rep:", d["value"], "steps/s  kernel_us", d["roofline"]["kernel_us"])
PY
  sed 's/^/    /' /tmp/clk_$rep.txt
done
