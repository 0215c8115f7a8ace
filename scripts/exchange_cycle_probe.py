"""Host time line of one replica-exchange cycle at world size 1 (RCCL group of one): run(1000) | energy | exchange |
next run() up to its first step launch.  GPU box only.  usage: python scripts/exchange_cycle_probe.py [cycles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I, ops, unit
from chiron_amd.multistate import ReplicaExchange
from chiron_amd.rng import PRNGKey
from chiron_amd.states import ThermodynamicState

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
rex = ReplicaExchange([ThermodynamicState(pot, temperature=300.0 * unit.kelvin)], PRNGKey(2024, 7), rank=0, world_size=1)
marks = {}
orig = I._RunState._enqueue_window


def enq(self, s0, n, *a, **k):
    if s0 == 0:
        marks["first_launch"] = time.perf_counter()
    return orig(self, s0, n, *a, **k)


I._RunState._enqueue_window = enq
state, nbr = integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
rex.exchange(pot.compute_energy(state.positions, nbr))
rows = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(cycles):
    ta = time.perf_counter()
    state, nbr = integ.run(state, rex.my_state, number_of_steps=1000, nbr_list=nbr)
    tb = time.perf_counter()
    e = pot.compute_energy(state.positions, nbr)
    tc = time.perf_counter()
    rex.exchange(e)
    td = time.perf_counter()
    rows.append((marks["first_launch"] - ta, tb - marks["first_launch"], tc - tb, td - tc))
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
r = 1e6 * np.median(np.array(rows), axis=0)
print(f"{cycles} cycles of 1000 steps: {1e6 * t_all / cycles:.0f} us per cycle = {1000 * cycles / t_all:.0f} steps/s")
print(f"  run(): entry -> first step launch enqueued  {r[0]:7.1f} us")
print(f"  run(): first launch -> return               {r[1]:7.1f} us  (1000 steps + rebuilds + final kick + state copy)")
print(f"  compute_energy (enqueue)                    {r[2]:7.1f} us")
print(f"  exchange (gather + decisions)               {r[3]:7.1f} us")
