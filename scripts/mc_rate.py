"""Monte Carlo displacement trials/s at a workload (bench.py's mc_block alone).  GPU box only.
usage: python scripts/mc_rate.py [workload]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench
args = types.SimpleNamespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", chunk=32, build="cell")
for _ in range(2):
    print(bench.mc_block(args), flush=True)
