"""Wall-clock phases of the Langevin path (diagnostic, GPU box only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def sync():
    torch.cuda.synchronize()

def t(label, fn, reps=1):
    sync(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync(); dt = (time.perf_counter() - t0) / reps
    print(f"{label:45s} {dt*1e3:10.3f} ms")
    return out

for wl in sys.argv[1:] or ["c3", "c4"]:
    print("==", wl)
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    t("build_from_state (first, incl. lazy loads)", lambda: nbr.build_from_state(state))
    t("build_from_state (second)", lambda: nbr.build_from_state(state))
    t("force_list (first)", lambda: nbr.force_list)
    def again():
        nbr._force_list = None
        return nbr.force_list
    t("force_list (second)", again)
    t("run 1 step", lambda: integ.run(state, thermo, number_of_steps=1, nbr_list=nbr))
    t("run 1 step again", lambda: integ.run(state, thermo, number_of_steps=1, nbr_list=nbr))
    out = t("run 200 steps", lambda: integ.run(state, thermo, number_of_steps=200, nbr_list=nbr))
    print("   rebuilds:", integ.n_rebuilds, "rebuild wall ms:", 1e3 * getattr(integ, "rebuild_seconds", 0.0))
    integ.rebuild_seconds = 0.0
    st2 = out[0]
    out = t("run 1000 steps (continued)", lambda: integ.run(st2, thermo, number_of_steps=1000, nbr_list=nbr))
    print("   rebuilds:", integ.n_rebuilds, "rebuild wall ms:", 1e3 * getattr(integ, "rebuild_seconds", 0.0))
