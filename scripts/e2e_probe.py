"""End-to-end steps/s of LangevinIntegrator.run at C3 outside bench.py (GPU box only): is the spread a property of
the process?  usage: python scripts/e2e_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("E2E_THREADS1", "1") == "1":
    torch.set_num_threads(1)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
out = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize()
    out.append(steps / (time.perf_counter() - t0))
t_step, t_force, p_nl, fl = bench.time_kernels(fluid, pot, state, nbr, integ, 200)
print("e2e windows:", [round(v) for v in out], " time_kernels step %.2f us force %.2f us" % (t_step * 1e6, t_force * 1e6))
fl = nbr.force_list
print("ptrs: fl %x deg %x pos %x" % (fl.fl.data_ptr(), fl.deg.data_ptr(), state.positions.data_ptr()))
