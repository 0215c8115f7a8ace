import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench
from chiron_amd import unit
from chiron_amd.multistate import ReplicaExchange
from chiron_amd.rng import PRNGKey
from chiron_amd.states import ThermodynamicState
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
rex = ReplicaExchange([ThermodynamicState(pot, temperature=300.0 * unit.kelvin)], PRNGKey(2024, 7), rank=0, world_size=1)
state, nbr = integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
rex.exchange(pot.compute_energy(state.positions, nbr))
def cycles(k):
    global state, nbr
    for _ in range(k):
        state, nbr = integ.run(state, rex.my_state, number_of_steps=1000, nbr_list=nbr)
        e = pot.compute_energy(state.positions, nbr)
        rex.exchange(e)
torch.cuda.synchronize(); t0 = time.perf_counter(); cycles(100); torch.cuda.synchronize()
print("us per cycle:", 1e4 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); cycles(200); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
dist.destroy_process_group()
