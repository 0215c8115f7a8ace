"""Aggregate steps/s of MultiStateSampler on ONE card, replicas propagated one after the other vs two / three at a time
(concurrent_replicas).  GPU box only.  usage: python scripts/multistate_concurrency.py [replicas] [steps per segment]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import unit as u
from chiron_amd.mcmc import LangevinDynamicsMove, MCMCSampler, MoveSchedule
from chiron_amd.multistate import MultiStateSampler
from chiron_amd.states import ThermodynamicState
from chiron_amd.rng import PRNGKey

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fluid, pot, state0, thermo0, nbr0, integ0 = bench.make_system("c3", 300.0, 32, "cell")
import copy
for workers in (1, 2, 3, 1, 2):
    thermo = [ThermodynamicState(pot, temperature=300.0 * 1.05 ** k * u.kelvin) for k in range(K)]
    states, lists = [], []
    for k in range(K):
        st = copy.deepcopy(state0)
        st.current_PRNG_key = PRNGKey(99, k)
        states.append(st)
        lists.append(copy.deepcopy(nbr0))
    move = LangevinDynamicsMove(timestep=1.0 * u.femtoseconds, number_of_steps=steps)
    ms = MultiStateSampler(MCMCSampler(MoveSchedule([("LangevinDynamicsMove", move)])), swap="neighbors", concurrent_replicas=workers)
    ms.create(thermo, states, lists)
    ms.run(n_iterations=2, n_mcmc_iterations=1)  # warm-up (lists, scratch, clocks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ms.run(n_iterations=8, n_mcmc_iterations=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{K} replicas of N=32768, {steps} steps per segment, concurrent_replicas={workers}: {6 * K * steps / dt:9.0f} aggregate steps/s", flush=True)
