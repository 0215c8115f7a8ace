"""Feasibility probe: pair-force kernel time as a function of the list skin (GPU box only)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for skin in (0.5, 0.2, 0.12, 0.08):
    bench.SKIN_NM = skin
    print(f"--- skin {skin}", flush=True)
    sys.argv = ["kernel_bench.py", "c3", "c4"]
    os.environ["GS"] = "auto"
    exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
