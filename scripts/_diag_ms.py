import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import chiron_amd as ca
import chiron_amd.neighbors, chiron_amd.states, chiron_amd.integrators, chiron_amd.mcmc, chiron_amd.multistate, chiron_amd.unit
from oracle import chiron_oracle as O
from chiron_amd.mcmc import LangevinDynamicsMove, MCMCSampler, MoveSchedule
from chiron_amd.multistate import MultiStateSampler
from chiron_amd.rng import PRNGKey
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
u = ca.unit
SIGMA, EPS, CUTOFF = 0.34, 0.238 * 4.184, 1.02
K, n = 5, 2500
x, box = O.lj_lattice(n, 0.3, SIGMA, seed=7)
from chiron_amd.potential import LJPotential
import test_gpu_parity as T
pot = T.lj_pot(ca, n)
temps = [300.0 * 1.08 ** k for k in range(K)]
def run(workers):
    thermo = [ca.states.ThermodynamicState(pot, temperature=t * u.kelvin) for t in temps]
    states, lists = [], []
    for k in range(K):
        st = T.make_state(ca, x, box)
        st.current_PRNG_key = PRNGKey(4321, k)
        states.append(st)
        lists.append(ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                                   skin=0.06 * u.nanometer, n_max_neighbors=180))
    move = LangevinDynamicsMove(timestep=4.0 * u.femtoseconds, number_of_steps=120)
    ms = MultiStateSampler(MCMCSampler(MoveSchedule([("LangevinDynamicsMove", move)])), swap="neighbors", concurrent_replicas=workers)
    ms.create(thermo, states, lists)
    ms.run(n_iterations=4, n_mcmc_iterations=1)
    torch.cuda.synchronize()
    return ms
import threading, zlib
from chiron_amd import integrators as I
LOG = []
_orig_run = I.LangevinIntegrator.run
def h(t):
    if t is None:
        return 0
    t = getattr(t, "_value", t)
    if not isinstance(t, torch.Tensor):
        return zlib.crc32(np.asarray(t).tobytes())
    return zlib.crc32(t.detach().cpu().numpy().tobytes())
def logged_run(self, sampler_state, thermodynamic_state, number_of_steps=5000, nbr_list=None, **kw):
    kin = float(thermodynamic_state.temperature.value_in_unit(u.kelvin)) if hasattr(thermodynamic_state.temperature, "value_in_unit") else 0.0
    ent = (h(sampler_state.positions), h(sampler_state._velocities), h(nbr_list.ref_positions), tuple(int(v) for v in sampler_state._current_PRNG_key))
    out, nb = _orig_run(self, sampler_state, thermodynamic_state, number_of_steps, nbr_list, **kw)
    LOG.append((round(kin, 3), ent, tuple(self.rebuild_steps), h(out.positions), h(out.velocities), threading.current_thread().name))
    return out, nb
I.LangevinIntegrator.run = logged_run
from chiron_amd import neighbors as N
_orig_confirm = N.NeighborListNsqrd._confirm_build
STALE = []
def confirm(self, wait=True):
    spec = getattr(self, "_spec", None)
    before = self._status_host.tolist() if spec is not None else None
    ok = _orig_confirm(self, wait)
    if spec is not None and not wait:
        torch.cuda.current_stream().synchronize()
        after = self._status_host.tolist()
        if after != before:
            STALE.append((before, after, threading.current_thread().name))
            print("   STALE STATUS READ", before, "->", after, threading.current_thread().name, flush=True)
    return ok
#N.NeighborListNsqrd._confirm_build = confirm
MODE = os.environ.get("DIAG_MODE", "")
_orig_spec = N.NeighborListNsqrd._build_speculative
def spec(self, positions, publish=True, pack=None, ctrl=None, ctrl_keep=0, copy_positions=False, half_rows=True):
    if MODE == "noriders":
        pack = ctrl = None
    if MODE == "syncbefore":
        torch.cuda.current_stream().synchronize()
    r = _orig_spec(self, positions, publish=publish, pack=pack, ctrl=ctrl, ctrl_keep=ctrl_keep, copy_positions=copy_positions, half_rows=half_rows)
    if MODE == "syncafter":
        torch.cuda.current_stream().synchronize()
    return r
N.NeighborListNsqrd._build_speculative = spec
_orig_load = I._RunState._load
def load(self, x_user, v_user, step=0, owner=None, ref_user=None, ctrl_keep=0, packed=False):
    if MODE == "noriders":
        packed = False
    return _orig_load(self, x_user, v_user, step, owner=owner, ref_user=ref_user, ctrl_keep=ctrl_keep, packed=packed)
I._RunState._load = load
def snap(ms):
    integs = [m.move.move_schedule[0][1].integrator for m in ms._mcmc_sampler]
    return dict(u=ms._reporter.get_property("u_kn"), x=[s.positions.cpu().numpy() for s in ms.sampler_states],
                v=[s.velocities.cpu().numpy() for s in ms.sampler_states], assign=list(ms._replica_thermodynamic_states),
                nb=[int(l.n_builds) for l in ms._nbr_lists], ref=[l.ref_positions.cpu().numpy() for l in ms._nbr_lists])
ref = snap(run(1)); REFLOG = sorted(LOG, key=lambda r: (r[1][3], r[0])); LOG.clear()
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    w = 2 if i % 4 else 1
    LOG.clear()
    cur = snap(run(w))
    curlog = sorted(LOG, key=lambda r: (r[1][3], r[0]))
    du = np.argwhere(cur["u"] != ref["u"])
    if len(du) or cur["assign"] != ref["assign"]:
        bad += 1
        if bad <= 3:
            refd = {(r[1][3], r[0]): r for r in REFLOG}
            for r in curlog:
                q = refd.get((r[1][3], r[0]))
                if q is None:
                    print("   no ref entry for key/temp", r[1][3], r[0], r[5]); continue
                if r[:5] != q[:5]:
                    print("   T", r[0], "key", r[1][3], r[5], "entry x/v/ref same:", r[1][0] == q[1][0], r[1][1] == q[1][1], r[1][2] == q[1][2],
                          "rebuilds", r[2], "REF", q[2], "out same", r[3] == q[3], r[4] == q[4], flush=True)
        print(i, "workers", w, "u_kn diffs", du.tolist()[:10], "x differ", [int(not np.array_equal(a, b)) for a, b in zip(cur["x"], ref["x"])],
              "v differ", [int(not np.array_equal(a, b)) for a, b in zip(cur["v"], ref["v"])], "assign", cur["assign"], ref["assign"],
              "builds", cur["nb"], ref["nb"], "ref differ", [int(not np.array_equal(a, b)) for a, b in zip(cur["ref"], ref["ref"])], flush=True)
print("mismatching runs:", bad, flush=True)
