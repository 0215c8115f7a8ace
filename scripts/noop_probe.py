import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench
for wl in ("c3", "c4"):
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
    integ.noop_launches = 0
    nreb = 0
    t0 = time.perf_counter()
    n = 20000 if wl == "c3" else 8000
    state, nbr = integ.run(state, thermo, number_of_steps=n, nbr_list=nbr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(wl, f"{n/dt:.0f} steps/s, rebuilds {integ.n_rebuilds}, no-op launches {integ.noop_launches} ({integ.noop_launches/max(1,integ.n_rebuilds):.1f} per rebuild)", flush=True)
