"""BAOAB Langevin integrator behind the reference's ``LangevinIntegrator`` API.

Reference: chiron/integrators.py:13-286.  The Python loop of the reference (one
``jax.grad`` + ~12 XLA dispatches + one host sync per step, :174-205) becomes ONE HIP
launch per step on the neighbor-list path: ``k_lj_step`` evaluates the pair force and,
in the same kernel, applies the half-kicks that consume it followed by A-O-A, the
periodic wrap and the skin check of the *next* step (the trailing kick :195 of step
s-1 is merged with the leading kick :181 of step s; the arithmetic and its order are
unchanged).  Steps are enqueued from C behind a bounded launch queue (``launch_window``
launches ahead of the device, ``chx_langevin_fused_run``).  The skin check that the
reference turns into a Python ``if`` every step (:241) records the step at which it
fires in device memory and in a host-mapped copy the C loop polls; the few launches
still queued turn into no-ops, and the host rebuilds the neighbor list from exactly the
positions the reference would have rebuilt it from, then resumes.
"""
from __future__ import annotations

import collections
import copy
import logging
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops, unit
from ._device import as_tensor
from ._lib import CTRL_WORDS, Box, LangevinParams
from .rng import JaxKey, PRNGKey, fold_in, split
from .states import SamplerState, ThermodynamicState

log = logging.getLogger("chiron_amd")

_INT_MAX = 2**31 - 1
_TRACE_RUN = __import__("os").environ.get("CHX_TRACE_RUN") == "1"
_SPECULATIVE_REBUILD = __import__("os").environ.get("CHX_SYNC_REBUILD") != "1"
# The list builds inside run() write the force list (what the step kernels read) and only COUNT the reference-format
# half rows (the counts feed the reference's capacity checks, chiron/neighbors.py:709-727); the rows themselves are a
# function of the list's reference positions and are written when somebody reads `neighbor_list` (NeighborListNsqrd.
# __getattr__).  CHX_RUN_HALF_ROWS=1: write them with every build, as the reference does (A/B, tests).
_RUN_HALF_ROWS = __import__("os").environ.get("CHX_RUN_HALF_ROWS") == "1"
_CARRY_VALIDATE = __import__("os").environ.get("CHX_CARRY_VALIDATE", "1") != "0"
_EV_LOG_CAP = 4096
# List carry-over (LangevinIntegrator(list_carry=...)): a short run() that continues the trajectory of the previous one
# (same integrator, same list object, the very position tensor that run returned, untouched) starts on the force list its
# predecessor ended on instead of building one -- the reference's run-start build (chiron/integrators.py:169) shrinks to
# "reference positions := x0" and the caller's list is rebuilt from x0 only if somebody reads it.
# The environment only supplies the DEFAULTS of the two constructor arguments (CHX_LIST_CARRY=0 / CHX_LIST_CARRY_MAX_STEPS):
# runs longer than the limit keep the build on entry, because a carried list is older, runs out of its guarantee sooner,
# and from ~100 steps on the renewals cost what the build would have (measured, round 3).
_CARRY_DEFAULT = __import__("os").environ.get("CHX_LIST_CARRY", "1") != "0"
_CARRY_MAX_STEPS_DEFAULT = int(__import__("os").environ.get("CHX_LIST_CARRY_MAX_STEPS", "64"))

_MODS = None


def _mods():
    """(NeighborListNsqrd, PairListNsqrd, LJPotential, get_list_of_mass, initialize_velocities): imported on first use
    (the modules import each other) and kept -- a ``from . import`` statement inside run() costs over a microsecond per
    call, on the path to the first launch of every short run()."""
    global _MODS
    if _MODS is None:
        from .neighbors import NeighborListNsqrd, PairListNsqrd
        from .potential import LJPotential
        from .utils import get_list_of_mass, initialize_velocities

        _MODS = (NeighborListNsqrd, PairListNsqrd, LJPotential, get_list_of_mass, initialize_velocities)
    return _MODS


_ENV_DATA = getattr(os.environ, "_data", None)
if not (isinstance(_ENV_DATA, dict) and all(isinstance(k, bytes) for k in list(_ENV_DATA)[:4])):
    _ENV_DATA = None  # (not CPython on POSIX: the mapping behind os.environ is not the bytes-keyed dict assumed below)


def _env(name: str, name_b: bytes):
    """``os.environ.get(name)`` for the two tuning overrides that are read on the way to a run()'s first step launch and
    must be read at call time (tests set them in-process): straight from the dict behind ``os.environ`` -- its ``get``
    encodes the key and decodes the value, 0.75 us a call -- where that dict is what CPython keeps on POSIX."""
    if _ENV_DATA is None:
        return os.environ.get(name)
    v = _ENV_DATA.get(name_b)
    return None if v is None else os.fsdecode(v)


def _wait_for_steps_and_stream(device_wide: bool = False) -> None:
    """Host wait for everything this thread has enqueued: the step launches (they run on the step queue paired with the
    stream, include/chx.h: chx_steps_join -- a wait for the stream alone does not cover them) and then the stream."""
    ops.steps_join()
    if device_wide:
        torch.cuda.synchronize()
    else:
        torch.cuda.current_stream().synchronize()


def _skip_join_hook() -> bool:
    """Test hook (tests/test_gpu_step_queue.py): a deliberately MISSING join at the end of run(), which the step queue's
    debug mode (ops.queue_debug) has to turn into poison in the returned state every time."""
    if ops._debug_skip_final_join > 0:
        ops._debug_skip_final_join -= 1
        return True
    return False


def langevin_constants(timestep, collision_rate):
    """(half_dt, a, b) in fp32 exactly as chiron/integrators.py:132-137 forms them."""
    dt = float(timestep.value_in_unit_system(unit.md_unit_system))
    gamma = float(collision_rate.value_in_unit_system(unit.md_unit_system))
    half_dt = np.float32(dt * 0.5)
    a = np.exp(np.float32(-gamma * dt))
    b = np.sqrt(np.float32(1.0) - np.exp(np.float32(-2.0 * gamma * dt)))
    return float(half_dt), float(a), float(b)


class LangevinIntegrator:
    """Langevin dynamics with the BAOAB splitting (Leimkuhler & Matthews 2013).

    Same constructor and ``run`` signature as chiron/integrators.py:25-105.
    Additions (none changes the steps a trajectory takes): ``chunk_size`` (steps enqueued per C call),
    ``launch_window`` (launches kept queued ahead of the device) and ``list_carry`` / ``list_carry_max_steps``: a
    ``run()`` of at most that many steps that continues the previous call's trajectory starts on the force list that
    call ended on instead of building one; the caller's list object is then brought up to the reference's state (rebuilt
    from the run's last reference positions) lazily, the first time one of its arrays is read.  ``list_carry=False``
    builds on every entry, exactly like chiron/integrators.py:169.
    """

    def __init__(
        self,
        timestep=1.0 * unit.femtoseconds,
        collision_rate=1.0 / unit.picoseconds,
        refresh_velocities: bool = False,
        report_interval: int = 100,
        reporter=None,
        save_traj_in_memory: bool = False,
        chunk_size: int = 32,
        launch_window: Optional[int] = None,
        list_carry: Optional[bool] = None,
        list_carry_max_steps: Optional[int] = None,
    ) -> None:
        self.kB = unit.BOLTZMANN_CONSTANT_kB * unit.AVOGADRO_CONSTANT_NA
        self.timestep = timestep
        self.collision_rate = collision_rate
        if reporter:
            self.reporter = reporter
        self.report_interval = report_interval
        self.velocities = None
        self.save_traj_in_memory = save_traj_in_memory
        self.traj = []
        self.refresh_velocities = refresh_velocities
        self._move_iteration = 0
        self.chunk_size = max(1, int(chunk_size))
        # step launches the host keeps queued ahead of the device on the neighbor-list path.  After the skin check fires
        # everything still queued is dead time in front of the rebuild.
        if launch_window is None:
            launch_window = int(os.environ.get("CHX_LAUNCH_WINDOW", "12"))
        self.launch_window = max(1, int(launch_window))
        self.list_carry = _CARRY_DEFAULT if list_carry is None else bool(list_carry)
        self.list_carry_max_steps = _CARRY_MAX_STEPS_DEFAULT if list_carry_max_steps is None else max(0, int(list_carry_max_steps))
        self.n_rebuilds = 0  # neighbor-list rebuilds triggered inside the last run()

    # Per-process device resources a run() leaves on the integrator for the next one: host-mapped pinned slots the
    # step kernel stores into, device scratch, cached unit arithmetic keyed by object identity.  A copy must not
    # share them (two integrators polling one slot) and cannot clone them (a deep-copied pinned tensor is pageable
    # memory, and its numpy view no longer aliases it): it starts without and allocates its own on first use.
    # MultiStateSampler deep-copies its sampler, chiron/multistate.py:266-268.
    _PER_PROCESS_STATE = ("_gen_ring", "_gen_next", "_pinned_ring", "_ring_next", "_dev_scratch", "_inner_scratch",
                          "_run_cache", "_defer_scratch", "_rebuild_log", "_carry", "_params_cache", "_mass_q", "_graveyard", "_run_fast")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._PER_PROCESS_STATE}

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__dict__.update({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return new

    # ------------------------------------------------------------------
    def run(
        self,
        sampler_state: SamplerState,
        thermodynamic_state: ThermodynamicState,
        number_of_steps: int = 5_000,
        nbr_list=None,
        progress_bar: bool = False,
        noise: Optional[torch.Tensor] = None,
    ) -> Tuple[SamplerState, object]:
        """chiron/integrators.py:75-218.  ``noise`` (number_of_steps, N, 3) optionally
        replaces the device PRNG in the O step (deterministic parity tests)."""
        ctx = self._begin(sampler_state, thermodynamic_state, number_of_steps, nbr_list, progress_bar, noise)
        st, potential, mode = ctx.st, ctx.potential, ctx.mode

        bar = None
        if progress_bar:
            try:
                from tqdm import tqdm

                bar = tqdm(total=number_of_steps)
            except Exception:  # pragma: no cover
                bar = None
        has_reporter = hasattr(self, "reporter") and self.reporter is not None
        wants_reports = has_reporter or self.save_traj_in_memory
        base = self._move_iteration * number_of_steps

        def next_stop(step: int) -> int:
            """First step count at which the host has to look at the state again."""
            if not wants_reports:
                return number_of_steps
            to_report = (-(step + base)) % self.report_interval  # steps until the next reporting step
            return min(number_of_steps, step + to_report + 1)

        done = 0
        while done < number_of_steps:
            stop = next_stop(done)
            before = done
            done = st.advance_to(done, stop, final=(stop == number_of_steps))
            if bar is not None:
                bar.update(done - before)
            last = done - 1
            elapsed_step = last + base
            if done == stop and elapsed_step % self.report_interval == 0:  # :197-205
                if has_reporter:
                    # (carried-over list: the list the steps run on is the one that is valid for these positions; the
                    # caller's list is brought up to the reference's last rebuild when the run ends)
                    self._report(st.positions_user(), potential, st.report_list(), last, self._move_iteration, elapsed_step)
                if self.save_traj_in_memory:
                    self.traj.append(st.positions_user())
        if bar is not None:
            bar.close()
        if number_of_steps > 0:
            st.final_kick(number_of_steps)
        elif mode == "fused":
            st._settle()
        return self._end(ctx)

    def _begin(self, sampler_state, thermodynamic_state, number_of_steps, nbr_list, progress_bar=False, noise=None,
               allow_carry: bool = True) -> "_RunContext":
        """Everything of ``run()`` in front of its first step: velocity rules, key, the list build of
        chiron/integrators.py:169 (enqueued first), the device-resident run state.  ``run_batch`` calls it once per replica."""
        NeighborListNsqrd, PairListNsqrd, LJPotential, get_list_of_mass, initialize_velocities = _MODS or _mods()

        potential = thermodynamic_state.potential
        topo = potential.topology
        raw = topo.masses_amu() if hasattr(topo, "masses_amu") else None
        self.box_vectors = sampler_state.box_vectors
        self.progress_bar = progress_bar
        temperature = thermodynamic_state.temperature
        x0 = sampler_state.positions
        # Everything below up to the velocity rules is a function of six objects that rarely change between run() calls --
        # the potential, its topology's mass array, the temperature, the time step, the collision rate -- and of the
        # device: one tuple of them (references held, so an id cannot be recycled) stands for the five caches behind it.
        cache = self.__dict__.setdefault("_run_cache", {})
        fast = self.__dict__.get("_run_fast")
        if (fast is not None and raw is not None and fast[0] is potential and fast[1] is raw and fast[2] is temperature
                and fast[3] is self.timestep and fast[4] is self.collision_rate and fast[5] == x0.device):
            mass, kbT_unitless, mass_unitless, self._m_min, half_dt, a, b = fast[6]
        else:
            # (the Quantity around the topology's cached mass array costs 3.5 us to make: kept while the array is the same object)
            held = self.__dict__.get("_mass_q")
            if raw is not None and held is not None and held[0] is raw:
                mass = held[1]
            else:
                mass = get_list_of_mass(topo)
                self._mass_q = (raw, mass) if raw is not None else None

            # unit arithmetic is slow Python (exact rationals) and the mass upload a host->device copy:
            # both are functions of objects that rarely change between run() calls
            kT_key = ("kT", id(temperature))
            if cache.get(kT_key, (None,))[0] is not temperature:
                stale = [k for k in cache if k[0] == "kT"]
                if len(stale) >= 16:  # a temperature ladder fits; anything longer-lived must not grow without bound
                    for k in stale:
                        del cache[k]
                cache[kT_key] = (temperature, float((self.kB * temperature).value_in_unit_system(unit.md_unit_system)))
            kbT_unitless = cache[kT_key][1]
            m_key = ("mass", id(potential.topology), x0.device)  # (a torch.device hashes and compares; str() of it costs 1 us)
            if cache.get(m_key, (None,))[0] is not mass._value:
                cache[m_key] = (mass._value, as_tensor(mass.value_in_unit_system(unit.md_unit_system)))
            mass_unitless = cache[m_key][1]
            if len(cache[m_key]) < 3:  # lightest particle: the speed scale of the pruned inner list's skin
                cache[m_key] = cache[m_key] + (float(np.min(np.asarray(mass.value_in_unit_system(unit.md_unit_system)))),)
            self._m_min = cache[m_key][2]
            c_key = ("dt",)
            if cache.get(c_key, (None, None))[:2] != (id(self.timestep), id(self.collision_rate)):
                cache[c_key] = (id(self.timestep), id(self.collision_rate), langevin_constants(self.timestep, self.collision_rate),
                                self.timestep, self.collision_rate)
            half_dt, a, b = cache[c_key][2]
            self._run_fast = (potential, raw, temperature, self.timestep, self.collision_rate, x0.device,
                              (mass, kbT_unitless, mass_unitless, self._m_min, half_dt, a, b))
        # velocity (re)initialisation rules, :140-160
        needs_velocities = (self.refresh_velocities or sampler_state._velocities is None
                            or sampler_state._velocities.shape[0] != sampler_state.positions.shape[0])

        # The list build of chiron/integrators.py:169 goes to the device FIRST: nothing below depends on it until the
        # first step is launched, and that launch has to wait for the build's ~65 us of device time -- every microsecond
        # of host work done before it is enqueued is a microsecond the device idles at the start of a run().  A list this
        # integrator has built before for the same box is rebuilt without waiting for its status block (validated after
        # the first launches, see _RunState._settle), and the build's own launches copy x0 for the list, pack the state
        # into the integrator's buffers and reset the control block (chx_build_riders_t) -- when the velocities are there
        # already; a run that has to draw them first packs behind the build instead.
        is_lj = isinstance(potential, LJPotential)
        spec0 = packed0 = False
        carried = self._take_carry(sampler_state, nbr_list, number_of_steps) if (is_lj and allow_carry) else None
        if not allow_carry:
            self._carry = None
        if carried is not None:
            # (no build: _RunState validates the carried force list against x0 on the device -- its pack launch rides on
            # the step queue with the steps, behind one gate)
            if isinstance(x0, torch.Tensor) and x0.is_cuda:
                _RunState.device_buffers(self, x0.shape[0], x0.device)  # (allocations before the gate: see below)
                ops.steps_open()
        elif (nbr_list is not None and is_lj and _SPECULATIVE_REBUILD and isinstance(nbr_list, NeighborListNsqrd)
                and getattr(nbr_list, "_box_token", None) is not None
                and nbr_list._box_token is getattr(sampler_state, "_box_token", None)
                and isinstance(x0, torch.Tensor) and x0.is_cuda and x0.dtype == torch.float32):
            pack = ctrl = None
            v_in = None if needs_velocities else sampler_state.velocities
            if (v_in is not None and v_in.is_cuda and v_in.dtype == torch.float32 and v_in.is_contiguous()
                    and mass_unitless.device == x0.device):
                buf = _RunState.device_buffers(self, x0.shape[0], x0.device)
                pack = (v_in, mass_unitless, None, buf["x_a"], buf["v"], buf["mass"], buf["ref4"])
                ctrl = buf["ctrl"]
            # (the build rides on the step queue with the steps: one gate for the whole run(), here -- behind whatever
            # produced x0 and the velocities on the stream, and behind the allocation of the integrator's buffers above)
            ops.steps_open()
            spec0 = nbr_list._build_speculative(x0, publish=False, pack=pack, ctrl=ctrl, copy_positions=True,
                                                half_rows=_RUN_HALF_ROWS)
            packed0 = bool(spec0) and pack is not None

        key = sampler_state.new_PRNG_key  # :124

        if needs_velocities:
            sampler_state.velocities = initialize_velocities(temperature, potential.topology, key)
        v0 = sampler_state.velocities

        if nbr_list is not None and not spec0 and carried is None:  # :169
            nbr_list.build_from_state(sampler_state)
        if is_lj and isinstance(nbr_list, NeighborListNsqrd):
            mode = "fused"
        elif is_lj and (nbr_list is None or isinstance(nbr_list, PairListNsqrd)):
            mode = "pairs"
        else:
            mode = "external"
        if mode != "external" and nbr_list is not None:
            # same validation as compute_energy (chiron/potential.py:263-270)
            if not nbr_list.is_built:
                raise ValueError("Neighborlist must be built before use")
            lc = nbr_list.cutoff
            lc_key = ("list_cutoff",)
            if lc is not None and cache.get(lc_key, (None,))[0] is not lc:  # (unit arithmetic: once per cutoff object)
                cache[lc_key] = (lc, lc.value_in_unit_system(unit.md_unit_system))
            if lc is not None and cache[lc_key][1] != potential.cutoff:
                raise ValueError(
                    f"Neighborlist cutoff ({nbr_list.cutoff}) must be the same as the potential cutoff ({potential.cutoff})"
                )

        chain_key = key
        if isinstance(key, JaxKey) and noise is None and number_of_steps > 0:
            # JAX-compatible stream: `key, subkey = random.split(key)`; `random.normal(subkey, x.shape)` per step
            # (chiron/integrators.py:179,185), generated on the device and fed through the injected-noise path
            n_atoms = x0.shape[0]
            if number_of_steps * n_atoms * 12 > (1 << 30):
                raise ValueError("the threefry-compatible stream materialises the noise of a whole run() call "
                                 f"({number_of_steps} x {n_atoms} x 3 floats): split the run or use the Philox stream")
            noise = torch.empty((number_of_steps, n_atoms, 3), dtype=torch.float32, device=x0.device)
            for s_ in range(number_of_steps):
                chain_key, subkey = split(chain_key)
                ops.jax_normal(subkey, (n_atoms, 3), out=noise[s_])
        st = _RunState(self, potential, nbr_list, x0, v0, mass_unitless, kbT_unitless, half_dt, a, b, key, noise, mode,
                       carry=carried, packed=packed0 and mode == "fused")
        if spec0:
            st._unsettled = (nbr_list.ref_positions, v0, 0, 1)

        # Host work that does not depend on the steps' results -- the key the reference returns (:216: the one its loop
        # ended with; the counter-based stream folds the step count in) and the copy of the box for the returned state --
        # is done while the last launches are still queued, right before the host waits for them (_enqueue_kick); behind
        # the wait the device would idle through it, in front of the first launch as well.
        tail = {}

        def host_tail():
            tail["key"] = chain_key if isinstance(key, JaxKey) else fold_in(key, number_of_steps)
            tail["box"] = sampler_state._box_copy()
            # (and the two arrays the NEXT run()'s list build needs before its first launch can go out)
            if mode == "fused" and spec0:
                nbr_list._prepare_next_build(x0)

        def before_last_wait():
            # (_RunState._enqueue_kick, between the run's last launch and the wait for it: the state run() returns, around
            # the arrays that launch is writing -- made here, not behind the wait)
            fs = st.__dict__.get("_final_state")
            if fs is not None and tail:
                tail["xv"] = fs
                tail["state"] = sampler_state._copy_with(*fs, box_copy=tail["box"])

        st._host_tail = host_tail
        st._before_last_wait = before_last_wait
        return _RunContext(st, potential, mode, nbr_list, sampler_state, number_of_steps, tail, host_tail)

    def _end(self, ctx: "_RunContext") -> Tuple[SamplerState, object]:
        """Everything of ``run()`` behind its last launch: the state that is returned (chiron/integrators.py:210-218)."""
        st, mode, sampler_state, tail, host_tail = ctx.st, ctx.mode, ctx.sampler_state, ctx.tail, ctx.host_tail
        st.finish()
        if not tail:
            host_tail()
        final_xv = st.final_state_user()
        st._host_tail = st._before_last_wait = None  # (closures over `st`: no cycle may outlive the call)
        ready = tail.get("state")
        if ready is not None and tail["xv"][0] is not final_xv[0]:
            ready = None  # (the last launch turned out a no-op -- the skin check fired in the very last step -- and was redone)
        if mode != "fused":
            # (the launches of the paths without a Verlet list ride on the step queue too -- k_ap_step, the copy into user
            # order just made: a run() returns with the queue drained, whatever the path; the fused path has waited in
            # _enqueue_kick)
            if not _skip_join_hook():
                ops.steps_join()
        updated_sampler_state = ready if ready is not None else sampler_state._copy_with(*final_xv, box_copy=tail["box"])  # :210-218
        st.offer_carry(updated_sampler_state)
        updated_sampler_state.current_PRNG_key = tail["key"]
        self.n_rebuilds = st.n_rebuilds
        self._rebuild_log = st.rebuild_log
        if getattr(self, "keep_run_state", False):
            # (tests / audits: the device-resident state the last launch of the run saw -- force list, pruned inner
            # list, control block, positions in internal order.  Off by default: it pins the run's buffers.)
            self._last_run_state = st
        return updated_sampler_state, st.nbr_list

    def _take_carry(self, sampler_state, nbr_list, number_of_steps: int):
        """The private list object whose force list this run may start on (list carry-over), or None.

        Offered by the previous ``run()`` (``_RunState.offer_carry``) and taken only when this call continues that
        trajectory: same list object, same box values, same cutoff and skin, and ``sampler_state.positions`` IS the
        tensor that run returned, not written to since -- positions from anywhere else (a Monte Carlo move, another
        replica) take the ordinary path.  Such positions are covered by the list -- the run that returned them checked
        them against it in its last launch and renewed it if need be (``_RunState.final_kick``) -- unless somebody wrote
        to the tensor behind the version stamp's back: the device checks again (chx_validate_snap)."""
        NeighborListNsqrd = (_MODS or _mods())[0]

        offer, self._carry = self.__dict__.get("_carry"), None
        if offer is None or not self.list_carry or not (0 < number_of_steps <= self.list_carry_max_steps):
            return None
        if not isinstance(nbr_list, NeighborListNsqrd):
            return None
        if (hasattr(self, "reporter") and self.reporter is not None) or self.save_traj_in_memory:
            return None
        x0 = sampler_state.positions
        if not (offer["public"] is nbr_list and x0 is offer["x"] and x0._version == offer["x_version"]
                and nbr_list.is_built and nbr_list._cs_md is not None and nbr_list._cs_md == offer["cs_md"]
                and nbr_list._skin is offer["skin"]
                and getattr(nbr_list, "_box_token", None) is not None
                and nbr_list._box_token is getattr(sampler_state, "_box_token", None)
                and nbr_list._box_token is offer["box_token"]):
            return None
        owner = offer["owner"]
        if owner is nbr_list:
            owner = nbr_list._detach_for_carry()
            if owner is None:
                return None
        max_n = getattr(owner, "_last_max_n", None)
        if max_n is None or max_n > 0.85 * int(nbr_list.n_max_neighbors):  # (the reference's growth rule needs to see
            return None                                                   # every build's row counts)
        return owner

    @property
    def rebuild_steps(self):
        """Step indices (within the last ``run``) at which the reference rebuilt its neighbor list
        (chiron/integrators.py:241-242), in order: the ones the host carried out and the ones the step kernel took
        on the device (carried-over force list)."""
        host_events, dev_log = getattr(self, "_rebuild_log", ([], None))
        dev = [] if dev_log is None else [int(v) for v in dev_log.cpu().tolist()]
        return sorted(list(host_events) + dev)

    def _wrap_and_rebuild_neighborlist(self, x, nbr_list):
        """chiron/integrators.py:220-243 (kept for API parity; ``run`` fuses this on the device)."""
        x = nbr_list.space.wrap(x, self.box_vectors)
        if nbr_list.check(x):
            nbr_list.build(x, self.box_vectors)
        return x, nbr_list

    def _report(self, x, potential, nbr_list, step: int, iteration: int, elapsed_step: int):
        """chiron/integrators.py:245-286"""
        d = {
            "positions": x,
            "potential_energy": potential.compute_energy(x, nbr_list),
            "step": step,
            "iteration": iteration,
            "elapsed_step": elapsed_step,
        }
        if nbr_list is not None:
            d["box_vectors"] = nbr_list.box_vectors
        self.reporter.report(d)


# ---------------------------------------------------------------------------------------------------------------------
# Several replicas side by side: one launch per step for all of them
# ---------------------------------------------------------------------------------------------------------------------
def batch_applies(integrators, sampler_states, thermodynamic_states, nbr_lists, number_of_steps: int) -> bool:
    """Can ``run_batch`` step these replicas in one launch per step?  LJ potential on a Verlet list on the GPU, fp32
    positions, one integrator per replica (its device buffers live on it), nothing that wants to look at the state
    between steps (reporter, trajectory in memory)."""
    NeighborListNsqrd, _, LJPotential = (_MODS or _mods())[:3]
    if number_of_steps <= 0 or len(integrators) < 2 or len({id(i) for i in integrators}) != len(integrators):
        return False
    if len(integrators) > 64:  # (chx_langevin_batch_run takes up to 64 items a call)
        return False
    if len({id(l) for l in nbr_lists}) != len(nbr_lists):
        return False
    for integ, ss, ts, nl in zip(integrators, sampler_states, thermodynamic_states, nbr_lists):
        x = ss.positions
        if not (isinstance(ts.potential, LJPotential) and isinstance(nl, NeighborListNsqrd) and isinstance(x, torch.Tensor)
                and x.is_cuda and x.dtype == torch.float32 and x.device == sampler_states[0].positions.device):
            return False
        if (getattr(integ, "reporter", None) is not None) or integ.save_traj_in_memory:
            return False
    return True


def run_batch(integrators, sampler_states, thermodynamic_states, number_of_steps: int, nbr_lists):
    """``integrators[r].run(sampler_states[r], thermodynamic_states[r], number_of_steps, nbr_lists[r])`` for every r, with ONE
    launch per step for all replicas instead of one per replica and step (include/chx.h: chx_langevin_batch_run).  The
    reference's multistate loop is serial (chiron/multistate.py:497-510); a step of an N = 32 768 replica is a latency
    chain that occupies a quarter of an MI355X, eight of them in one grid fill it.  Each replica keeps its own box,
    temperature, key, lists, control block and rebuild schedule; its launches are the one-replica launches instruction for
    instruction, so the returned states are bit-identical to the serial calls.  Falls back to those when the batch does
    not apply (``batch_applies``).  Returns [(SamplerState, nbr_list)] in order."""
    R = len(integrators)
    if not (len(sampler_states) == len(thermodynamic_states) == len(nbr_lists) == R):
        raise ValueError("run_batch: one integrator, sampler state, thermodynamic state and neighbor list per replica")
    if not batch_applies(integrators, sampler_states, thermodynamic_states, nbr_lists, number_of_steps):
        return [i.run(s, t, number_of_steps=number_of_steps, nbr_list=n)
                for i, s, t, n in zip(integrators, sampler_states, thermodynamic_states, nbr_lists)]
    ctxs = [i._begin(s, t, number_of_steps, n, allow_carry=False)
            for i, s, t, n in zip(integrators, sampler_states, thermodynamic_states, nbr_lists)]
    if any(c.mode != "fused" for c in ctxs):  # (cannot happen behind batch_applies)
        raise RuntimeError("run_batch: a replica is not on the fused neighbor-list path")
    _advance_batch([c.st for c in ctxs], number_of_steps, integrators[0].launch_window, integrators[0].chunk_size)
    return [i._end(c) for i, c in zip(integrators, ctxs)]


def _advance_batch(sts, n_steps: int, window: int, chunk: int) -> None:
    """Steps [0, n_steps) and the trailing kick of every run state in ``sts``, batched.  The per-replica logic is
    ``_RunState._advance_windowed`` / ``final_kick`` -- halt word, rebuild behind the queued no-ops, speculative builds
    settled later -- with the launches of all replicas that are not waiting for a rebuild going out together; a replica
    that has halted is a no-op in the launches still queued, its rebuild goes out behind them on the same queue, and it
    continues from its halt step (the replicas drift apart by the launches they sat out and finish a few launches after
    each other).  (Rebuilding a halted replica's list on a second queue BESIDE the others' steps was built and measured
    in round 6: the step launch of eight replicas saturates the card, the build's kernels slowed it by as much as they
    gained -- 415 k against 440 k replica-steps/s -- and the scheme is gone.)"""
    R = len(sts)
    enq = [0] * R        # first step not enqueued yet, per replica
    kicked = [False] * R
    window = max(1, int(window))
    while True:
        # ---- enqueue until every replica has its steps out, or a halt word is seen ----
        saw_halt = False
        while not saw_halt:
            todo = [r for r in range(R) if enq[r] < n_steps]
            if not todo:
                break
            entries = []
            for r in todo:
                st = sts[r]
                even, odd = (st.x_a, st.x_b) if st.parity == 0 else (st.x_b, st.x_a)
                entries.append((st.params, even, odd, enq[r], min(chunk, n_steps - enq[r]), st.kicks_next))
            # (the last launches go out unthrottled, as in _advance_windowed)
            tail_now = all(n_steps - enq[r] <= chunk for r in todo)
            sent = ops.langevin_batch_run(entries, (1 << 20) if tail_now else window)
            for r, e, k in zip(todo, entries, sent):
                st = sts[r]
                if k:
                    st.parity ^= k & 1
                    st.kicks_next = 2
                    enq[r] += k
                if k < e[4]:
                    saw_halt = True
            for st in sts:  # host work of the runs' END that does not depend on the steps, under the queued launches
                work = getattr(st, "_host_tail", None)
                if work is not None:
                    st._host_tail = None
                    work()
            for r in todo:
                st = sts[r]
                pend = getattr(st, "_unsettled", None)
                if pend is not None and enq[r] - pend[2] >= max(chunk, window + 1):
                    if not st._settle():
                        enq[r] = st._unsettled_step
        all_out = all(e == n_steps for e in enq)
        if all_out:
            need = [r for r in range(R) if not kicked[r]]
            if need:
                entries = []
                for r in need:
                    st = sts[r]
                    f32 = dict(dtype=torch.float32, device=st.dev)
                    x3, v3 = ops.fresh((st.n, 3), **f32), ops.fresh((st.n, 3), **f32)
                    st.params.final_x3, st.params.final_v3, st.params.final_ref3 = x3.data_ptr(), v3.data_ptr(), None
                    st._final_state = (x3, v3)
                    st._final_ref = None
                    even, odd = (st.x_a, st.x_b) if st.parity == 0 else (st.x_b, st.x_a)
                    entries.append((st.params, even, odd, n_steps, 1, 1))
                try:
                    ops.langevin_batch_kick(entries)
                finally:
                    for r in need:
                        sts[r].params.final_x3 = sts[r].params.final_v3 = sts[r].params.final_ref3 = None
                for r in need:
                    kicked[r] = True
                    st = sts[r]
                    work, st._host_tail = getattr(st, "_host_tail", None), None
                    if work is not None:
                        work()
                    pre, st._before_last_wait = st.__dict__.get("_before_last_wait"), None
                    if pre is not None:
                        pre()
            if not _skip_join_hook():
                ops.steps_join()
                if not ops.step_queue_on():
                    # (plain HIP launches, CHX_AQL=0: the launches are on the stream, and what follows reads the halt words
                    # on the host -- _RunState._wait_for_launches does the same)
                    torch.cuda.current_stream().synchronize()
            for st in sts:
                st.integ._graveyard = None
        # ---- what the device found: rejected speculative builds, halt words ----
        again = False
        for r, st in enumerate(sts):
            halt_seen = int(st._gen.host[3]) <= enq[r]
            if not (all_out or halt_seen):
                continue
            if not st._settle():
                enq[r] = st._unsettled_step
                kicked[r] = False
                st._final_state = None
                again = True
                continue
            halt = int(st._gen.host[3])
            if halt > n_steps or halt > enq[r]:
                continue
            # the skin check fired while producing x_halt: steps < halt ran, what is queued behind is no-ops, and the
            # rebuild goes out behind them (chiron/integrators.py:239-242)
            st._final_state = None
            kicked[r] = False
            st.integ.noop_launches = getattr(st.integ, "noop_launches", 0) + max(0, enq[r] - halt)
            st.parity = (st._gen_parity0 + halt - st._gen_step0) & 1
            st._close_generation()
            st._halt_rebuild(halt, speculative=True)
            enq[r] = halt
            again = True
        if all_out and not again:
            break
    for st in sts:
        st._kick_sent = False
        st._close_generation()
        st._retire_all()


class _RunContext:
    """What ``LangevinIntegrator._begin`` hands to the stepping loop and to ``_end``."""

    __slots__ = ("st", "potential", "mode", "nbr_list", "sampler_state", "number_of_steps", "tail", "host_tail")

    def __init__(self, st, potential, mode, nbr_list, sampler_state, number_of_steps, tail, host_tail):
        self.st, self.potential, self.mode, self.nbr_list = st, potential, mode, nbr_list
        self.sampler_state, self.number_of_steps, self.tail, self.host_tail = sampler_state, number_of_steps, tail, host_tail


class _Decoupled:
    """Bookkeeping of a run whose force list was built from OTHER positions than the reference's reference positions
    (list carry-over, ``_RunState._setup_carry``): the private list builders that take turns when the force list is
    renewed, and the number of renewals."""

    __slots__ = ("builders", "next", "n_internal_halts")

    def __init__(self):
        self.n_internal_halts = 0


class _Slot:
    """Host-mapped copy of the control block for one generation of step launches (_RunState._open_generation)."""

    __slots__ = ("tensor", "host", "open", "dirty", "step0")

    def __init__(self):
        self.tensor = torch.empty((CTRL_WORDS,), dtype=torch.int32, pin_memory=True)
        self.host = self.tensor.numpy()  # (a view: reading an element costs ~0.1 us instead of a tensor op)
        self.open = self.dirty = False
        self.step0 = 0


class _RunState:
    """Device-resident state of one ``run`` call in the kernels' internal layout."""

    @staticmethod
    def device_buffers(integ, n: int, dev) -> dict:
        """Device arrays kept on the integrator between run() calls (every launch of a run is enqueued behind the
        previous run's on the same stream): the control block, the rebuild-event log, the energy workspace and the
        Langevin state in internal order -- positions (two buffers: a step reads one and writes the other), velocities,
        masses, reference positions.  Everything a run() returns is a fresh copy, never one of these."""
        keep = integ.__dict__.setdefault("_dev_scratch", {})
        sk = (n, dev)
        if keep.get("key") != sk:
            keep.clear()
            keep["key"] = sk
            f32 = dict(dtype=torch.float32, device=dev)
            keep["ctrl"] = torch.zeros((CTRL_WORDS,), dtype=torch.int32, device=dev)
            keep["sync2"] = torch.zeros((2,), dtype=torch.int32, device=dev)  # (chx_pack_validate_langevin's ticket and flag)
            keep["ev_log"] = torch.zeros((_EV_LOG_CAP,), dtype=torch.int32, device=dev)
            keep["ws"] = torch.empty((int(ops._lib.load().chx_lj_ws_count(n)),), dtype=torch.float64, device=dev)
            keep["x_a"], keep["x_b"] = torch.empty((n, 4), **f32), torch.empty((n, 4), **f32)
            keep["v"], keep["mass"], keep["ref4"] = torch.empty((n, 3), **f32), torch.empty((n,), **f32), torch.empty((n, 4), **f32)
        return keep

    def __init__(self, integ, potential, nbr_list, x0, v0, mass, kT, half_dt, a, b, key: PRNGKey, noise, mode: str,
                 carry=None, packed: bool = False):
        """``packed``: the list build that preceded this call (``NeighborListNsqrd._build_speculative`` with the pack
        rider) has already written the state into the integrator's buffers in the new list's order and reset the
        control block."""
        NeighborListNsqrd, PairListNsqrd = (_MODS or _mods())[:2]

        self.integ = integ
        self.potential = potential
        self.nbr_list = nbr_list
        self.kT, self.half_dt, self.a, self.b = kT, half_dt, a, b
        self.key = key
        self.mode = mode
        self.n = x0.shape[0]
        self.dev = x0.device
        self.noise = None if noise is None else as_tensor(noise)
        if self.noise is not None and (self.noise.dim() != 3 or tuple(self.noise.shape[1:]) != (self.n, 3)):
            raise ValueError(f"noise must have shape (steps, {self.n}, 3), got {tuple(self.noise.shape)}")
        self.mass_user = mass
        keep = self.device_buffers(integ, self.n, self.dev)
        self.ctrl, self.ws, self.ev_log, self.sync2 = keep["ctrl"], keep["ws"], keep["ev_log"], keep["sync2"]
        self.is_verlet = isinstance(nbr_list, NeighborListNsqrd)
        self.is_pairlist = isinstance(nbr_list, PairListNsqrd)
        self.x_a, self.x_b, self.v, self.mass = keep["x_a"], keep["x_b"], keep["v"], keep["mass"]
        self.ref4 = keep["ref4"] if self.is_verlet else None  # (the skin check's reference positions: Verlet lists only)
        self.n_rebuilds = 0
        self.kicks_next = 1  # the first step of a run applies only its leading kick
        self._defer = None
        self._final_state = None
        self.rebuild_log = ([], None)
        self._snap_is_ref = False
        self._host_events = []  # steps of the reference rebuilds the HOST carried out (halting scheme)
        self._carried = carry if (carry is not None and mode == "fused" and self.is_verlet) else None
        if self._carried is not None:
            self._setup_carry()
            # the reference's run-start build, as far as the steps can tell: reference positions := x0
            # Does the carried list cover x0?  By the host's bookkeeping it does: x0 IS what the previous run returned
            # (_take_carry: same tensor, same version), that run's last launch checked exactly these positions against the
            # list's own (the halt word for step n_steps) and final_kick renewed the list if the check fired.  But a
            # tensor's version stamp does not see writes through .data or a raw pointer, and the price of being wrong is a
            # silently wrong force: the device checks again -- inside the pack launch (round 3: a launch of its own in
            # front of the first step, 13 us of a 185 us run(20)).  CHX_CARRY_VALIDATE=0 leaves the check out.
            self._load(x0, v0, 0, owner=self._carried, ref_user=x0, validate=_CARRY_VALIDATE)
            integ.n_carried_runs = getattr(integ, "n_carried_runs", 0) + 1
        else:
            self._load(x0, v0, packed=packed)
        if self.mode == "external":
            self._external_force()

    def pack_rider(self, v_user: torch.Tensor, ref_user=None):
        """(v3, mass, ref3, x4_out, v_out, mass_out, ref4_out) for ``NeighborListNsqrd._build_speculative(pack=...)``:
        the rebuild's own launches write the state into the buffers the steps read (no pack launch behind the build)."""
        return (v_user, self.mass_user, ref_user, self.x_a, self.v, self.mass, self.ref4)

    # -- list carry-over -------------------------------------------------------
    def _setup_carry(self):
        """List carry-over: the run starts on a force list built from OTHER positions than its reference positions
        (x0), so it runs on the decoupled step kernel (``snap4`` set: include/chx.h).  When the force list runs out of
        its own guarantee the halt word fires and only the force list is renewed (``_rebuild_internal``); the
        reference's rebuilds are events the step kernel takes itself."""
        d = self._defer = _Decoupled()
        keep = self.integ.__dict__.setdefault("_defer_scratch", {})
        if keep.get("key") != (self.n, self.dev):
            keep.clear()
            keep["key"] = (self.n, self.dev)
            keep["builders"] = [None, None]
        d.builders = keep["builders"]
        d.next = 1 if d.builders[0] is self._carried else 0  # (a rebuild must not write into the list being walked)

    def _next_builder(self):
        d = self._defer
        b = d.builders[d.next]
        src = self.nbr_list
        if b is None or b._fl_geo != src.__dict__.get("_fl_geo", b._fl_geo) or b._box_h is not src._box_h:
            nb = src._clone_builder()
            if b is not None:
                nb._scratch = b._scratch  # (work arrays: sizes are functions of n and the grid)
            b = d.builders[d.next] = nb
        b.n_max_neighbors = src.n_max_neighbors
        d.next ^= 1
        return b

    def _bury(self, lst):
        """Keep the arrays of the list that is about to be replaced alive until the NEXT rebuild (or the run's end).
        The rebuild rides on the step queue behind the launches of the generation that just halted (no join): the
        halting launch itself may still be reading the old force list when the host drops it, and the allocator would
        hand its memory to whoever asks next -- possibly a torch kernel on the stream, which the queue does not order.
        One rebuild later every launch of that generation has long run."""
        d = lst.__dict__
        self.integ._graveyard = (d.get("_force_list"), d.get("n_neighbors"), d.get("ref_positions"), d.get("_perm"),
                                 d.get("_nl"), d.get("_first"), getattr(self, "fl", None), getattr(self, "perm", None))

    def _rebuild_internal(self, step: int, speculative: bool = True):
        """The carried force list ran out of its guarantee (the halt word fired at ``step`` on the list's own
        positions): build the next one here and now, from the halted state.  No reference event is involved; one that
        is booked for this very step stays booked (``ctrl_keep``)."""
        d = self._defer
        x_user, v_user = self.state_user()
        ref_user = self._unpack(self.ref4)
        b = self._next_builder()
        self._bury(b)
        # (like the halting scheme's rebuild: enqueue the build, carry on, read its status block once it has arrived --
        # a builder that has built before for this box; its first build goes the synchronous way)
        # (the builder keeps a copy of the positions -- made by the binning launch, which rides on the step queue with the
        # rest of the rebuild; a torch clone would run on the stream, beside the queue)
        spec = (speculative and _SPECULATIVE_REBUILD and b.__dict__.get("_perm") is not None
                and b._build_speculative(x_user, publish=False, pack=self.pack_rider(v_user, ref_user), ctrl=self.ctrl,
                                         ctrl_keep=3, half_rows=_RUN_HALF_ROWS, copy_positions=True))
        if not spec:
            ops.steps_join()  # (the synchronous build reads its status on the host: nothing may be pending on the queue)
            b._adopt_positions = True
            try:
                b.build(x_user.clone(), b.box_vectors)
            finally:
                b._adopt_positions = False
        self._load(x_user, v_user, step, owner=b, ref_user=ref_user, ctrl_keep=3, packed=bool(spec))
        self._unsettled = (x_user, v_user, step, self.kicks_next, b, ref_user) if spec else None
        self._snap_is_ref = False
        d.n_internal_halts += 1

    # -- layout ------------------------------------------------------------
    def _pack(self, src, stride, perm):
        out = torch.empty((self.n, stride), dtype=torch.float32, device=self.dev)
        ops.check(
            ops._lib.load().chx_permute_pack(ops._ptr(ops._f32(src)), ops._ptr(perm), self.n, ops._ptr(out), stride, ops._stream()),
            "chx_permute_pack",
        )
        return out

    def _inner_list_settings(self, skin: float):
        """(inner_skin [nm], prune_every [steps]) of the dynamically pruned list, (0, 0) = off.

        The reference's skin (0.5 nm in its examples) keeps ~3x more pairs listed than can
        interact; between rebuilds the step kernel walks a copy pruned to cutoff + inner_skin
        that is refreshed every ``prune_every`` steps (include/chx.h: chx_langevin_t).  Pure
        speed knobs: the interacting pairs are the reference's in every case.
        """
        env_skin, env_every = _env("CHX_INNER_SKIN", b"CHX_INNER_SKIN"), _env("CHX_PRUNE_EVERY", b"CHX_PRUNE_EVERY")
        inner = float(env_skin) if env_skin is not None else getattr(self.integ, "inner_skin_nm", None)
        # Refresh cadence: where the step is a latency chain (N < 2^17) the refreshing launch (+4 us) is worth spreading
        # over 32 steps; where it is throughput-bound the table should be short instead -- refreshed more often, with a
        # thinner inner skin (round-3 sweep on the equilibrated fluid, fresh processes alternating: N = 262 144
        # 58.0 -> 60.1 k steps/s at 20 steps / 0.06 nm against 32 / 0.10; N = 131 072 +1 % at 24 / 0.072)
        # (dense rows -- rho* = 0.8: 300 listed partners -- make the refreshing launch itself expensive: 32 steps there
        # at every size; N = 262 144 at rho* = 0.8: 16.3 k steps/s at 32 against 16.1 k at 20)
        dense = self.fl is not None and int(getattr(self.fl, "maxdeg", 0) or 0) > 128
        default_every = 32 if (self.n < (1 << 17) or dense) else (24 if self.n < (1 << 18) else 20)
        every = int(env_every) if env_every is not None else getattr(self.integ, "prune_every", default_every)
        if inner is None:
            # little to prune under a thin skin; tiny systems are pure launch latency.  Otherwise the inner skin is what
            # the fastest particles (5 sigma_v) travel between two refreshes, on both sides of a pair, with 20 % to
            # spare: 2.4 * every * dt * 5 sqrt(kT / m_min) -- 0.096 nm for argon at 300 K, dt = 1 fs, every = 32 (round
            # 2 had tuned 0.12 / 0.10 nm by hand for exactly that system; 0.096 measures the same at N = 32 768 and
            # +1.3 % at N = 65 536), and the right length again for a hotter replica, a lighter particle, a longer step
            m_min = getattr(self.integ, "_m_min", None)
            if m_min is None and skin >= 0.3 and self.n >= 2048:  # (a state built outside run(): benches, probes)
                m_min = self.integ._m_min = float(self.mass_user.min().item())
            if skin >= 0.3 and self.n >= 2048 and m_min and every > 0:
                inner = 2.4 * every * (2.0 * self.half_dt) * 5.0 * float(np.sqrt(self.kT / m_min))
                inner = min(max(inner, 0.02), 0.6 * skin)
            else:
                inner = 0.0
        if inner <= 0.0 or every <= 0 or inner >= skin:
            return 0.0, 0
        return inner, every

    def _load(self, x_user: torch.Tensor, v_user: torch.Tensor, step: int = 0, owner=None, ref_user=None, ctrl_keep: int = 0,
              packed: bool = False, validate: bool = False):
        """(Re)pack user-order arrays into the internal order of the current force list (``packed``: the build's own
        launches have done that, and reset the control block) and point the launch parameters at that list.

        ``owner``: the list object whose force list the steps run on from here (default: the caller's ``nbr_list``;
        on a carried-over list one of the integrator's private builders).  ``ref_user``: the reference's reference
        positions when they are not the owner's (list carry-over: the public list's positions travel separately).
        ``ctrl_keep``: what survives the control-block reset (ops.pack_langevin).  ``validate`` (a run that starts on a
        carried-over list): the pack launch also checks x_user against the positions the force list was built from and
        leaves the verdict in the halt word it resets (chx_pack_validate_langevin)."""
        owner = self.nbr_list if owner is None else owner
        self._owner = owner
        self.fl = owner.force_list if self.mode == "fused" else None
        perm = self.fl.perm if self.fl is not None else None
        self.perm = perm
        n = self.n
        box = Box(0, 0.0, 0.0, 0.0)
        wrap = 0
        skin = 0.0
        if self.nbr_list is not None:
            box = self.nbr_list._box_h
            wrap = 1
            if self.is_verlet:
                sk = getattr(self, "_skin_md", None)  # (unit arithmetic is slow Python: once per skin object)
                if sk is None or sk[0] is not self.nbr_list.skin:
                    sk = self._skin_md = (self.nbr_list.skin, float(self.nbr_list.skin.value_in_unit_system(unit.md_unit_system)))
                skin = sk[1]
        check = None
        self.parity = 0  # (the pack below writes the buffer that is current at parity 0)
        if validate and not packed and self.mode == "fused":
            # (the verdict also goes to the generation's host-mapped slot: the slot has to be the new one already)
            self._open_generation(step)
            scratch = getattr(owner, "_scratch", None)
            xs4 = scratch["arrays"][4] if scratch and scratch.get("arrays") is not None else None
            if xs4 is None or xs4.shape[0] != n:
                raise RuntimeError("list carry-over: the force list's owner holds no cell-ordered positions")
            check = (xs4, box, skin, step, self._gen.tensor, self.sync2)
        if not packed:
            # one launch: positions (the buffer that is current at parity 0), velocities, masses, reference positions,
            # written into the integrator's buffers (a rebuild has copied the state out first), and the control-block reset
            if ref_user is None and self.is_verlet:
                ref_user = owner.ref_positions
            ops.pack_langevin(x_user, v_user, self.mass_user, ref_user if self.is_verlet else None, perm, self.is_verlet,
                              out=(self.x_a, self.v, self.mass, self.ref4), ctrl=self.ctrl, ctrl_keep=ctrl_keep, validate=check)
        if self.mode == "external":
            self.f = torch.zeros((n, 3), dtype=torch.float32, device=self.dev)
        else:
            self.f = None
        if self.mode == "fused" and check is None:
            self._open_generation(step)
        # The parameter block is kept on the integrator between loads: most of its ~40 fields are functions of objects
        # that outlive a run() -- the integrator's own device buffers, the box, the potential, the thermostat's constants --
        # and filling a ctypes structure costs a fifth of a microsecond per field on the path to the first launch.  The
        # cached block is reused when every one of those is the same object / value as when it was filled (the cache
        # holds references, so an id cannot be recycled); what belongs to the list, the key and the generation is set
        # every time.
        pot = self.potential if self.mode != "external" else None
        lj = (pot.sigma, pot.epsilon, pot.cutoff) if pot is not None else None
        objs = (self.x_a, self.v, self.f, self.ref4, self.mass, self.ctrl, self.ws, box, self.noise)
        vals = (n, self.mode, wrap, skin, self.half_dt, self.a, self.b, self.kT, lj,
                self.is_pairlist and self.nbr_list.cutoff is None)
        cached = self.integ.__dict__.get("_params_cache")
        if (cached is not None and cached[1] == vals and len(cached[0]) == len(objs)
                and all(u is w for u, w in zip(cached[0], objs))):
            p = cached[2]
            # (what _point_params_at_list and _enqueue_kick may have left from the previous load)
            p.fl_inner = p.deg_inner = p.snap4 = p.ev_log = p.final_x3 = p.final_v3 = p.final_ref3 = None
            p.inner_skin, p.prune_every, p.prune_phase, p.ev_log_cap = 0.0, 0, 0, 0
        else:
            p = LangevinParams()
            p.n = n
            p.x4 = self.x_a.data_ptr()
            p.v = self.v.data_ptr()
            p.f = self.f.data_ptr() if self.f is not None else None
            p.ref4 = self.ref4.data_ptr() if self.ref4 is not None else None
            p.mass = self.mass.data_ptr()
            p.box = box
            p.wrap = wrap
            p.skin = skin
            p.half_dt, p.a, p.b, p.kT = self.half_dt, self.a, self.b, self.kT
            if self.mode != "external":
                p.sigma, p.epsilon, p.cutoff = lj
                if self.is_pairlist and self.nbr_list.cutoff is None:
                    p.cutoff = -1.0
            p.noise = self.noise.data_ptr() if self.noise is not None else None
            p.noise_step0 = 0
            p.ctrl = self.ctrl.data_ptr()
            p.ws = self.ws.data_ptr()
            p.external_force = 1 if self.mode == "external" else 0
            self.integ._params_cache = (objs, vals, p)
        p.g = self.fl.g if self.fl is not None else 1
        p.fl_chunks = self.fl.chunks if self.fl is not None else 0
        p.perm = perm.data_ptr() if perm is not None else None
        p.fl = self.fl.fl.data_ptr() if self.fl is not None else None
        p.deg = self.fl.deg.data_ptr() if self.fl is not None else None
        p.k0, p.k1 = self.key[0], self.key[1]
        if self.mode == "fused":
            p.host_ctrl = self._gen.tensor.data_ptr()
        self._skin_now = skin
        self._point_params_at_list(p, owner, step)
        self.params = p

    def _point_params_at_list(self, p, owner, step: int):
        """The fields of the launch parameters that belong to the force list the steps run on: the pruned inner list
        (restarted at ``step``) and, on a carried-over list, the positions the list was built from."""
        n, skin = self.n, self._skin_now
        if self.fl is not None and self.is_verlet:
            inner, every = self._inner_list_settings(skin)
            if every > 0:
                # (kept on the integrator between run() calls: every launch of a run is enqueued behind the previous
                # run's on the same stream, and the first launch after a load refreshes the table)
                keep = self.integ.__dict__.setdefault("_inner_scratch", {})
                if keep.get("shape") != tuple(self.fl.fl.shape) or keep["fl"].device != self.fl.fl.device:
                    keep["shape"] = tuple(self.fl.fl.shape)
                    keep["fl"] = torch.empty_like(self.fl.fl)
                    keep["deg"] = torch.empty_like(self.fl.deg)
                self.fl_inner, self.deg_inner = keep["fl"], keep["deg"]
                p.fl_inner, p.deg_inner = self.fl_inner.data_ptr(), self.deg_inner.data_ptr()
                p.inner_skin, p.prune_every, p.prune_phase = inner, every, step
        d = self._defer
        if d is not None and self.fl is not None:
            # the positions the force list was built from, in its own order: the build's cell-ordered copy
            scratch = getattr(owner, "_scratch", None)
            xs4 = scratch["arrays"][4] if scratch and scratch.get("arrays") is not None else None
            if xs4 is None or xs4.shape[0] != n:  # (cannot happen: a list is only carried over from a cell build)
                raise RuntimeError("list carry-over: the force list's owner holds no cell-ordered positions")
            self.snap4 = xs4
            p.snap4 = xs4.data_ptr()
            p.ev_log, p.ev_log_cap = self.ev_log.data_ptr(), _EV_LOG_CAP
            if owner is self.nbr_list:
                self._snap_is_ref = True  # (a load on the public list: its positions are the reference's)

    # -- host-visible control block of the launch window ---------------------------------------
    def _open_generation(self, step: int):
        """A *generation* = the step launches between two loads of the state (list rebuilds).  All of them copy
        the control block into one host-mapped slot (chx_langevin_t::host_ctrl), and no launch of another
        generation writes there -- the no-op launches still in flight when the host starts a rebuild keep storing
        the OLD halt word, so the next generation needs a slot of its own.  Slots come from a ring of four; one is
        reused (and its counters banked) only once its launches are known to have run: a later generation has
        begun a launch of its own, which the stream orders behind them -- no event, no wait in the common case."""
        cur = getattr(self, "_gen", None)
        if cur is not None and cur.open:
            self._close_generation()
        ring = self.integ.__dict__.setdefault("_gen_ring", [])
        if not ring:
            ring.extend(_Slot() for _ in range(4))
            self.integ._gen_next = 0
        slot = ring[self.integ._gen_next % len(ring)]
        self.integ._gen_next += 1
        if slot.open or slot.dirty:
            if not (cur is not None and cur is not slot and int(cur.host[0]) >= cur.step0):
                _wait_for_steps_and_stream()  # (a run that ended in an exception, a zero-step run)
            self._retire(slot)
        # what chx_ctrl_reset writes, with word 0 = "the launch before `step` has begun": nothing is queued
        slot.host[:] = (step - 1, 0, 0, _INT_MAX, -2, -2, 0, 0, 0, -1, -1, -1, _INT_MAX, 0, 0, 0)
        slot.open, slot.dirty, slot.step0 = True, False, step
        self._gen, self._gen_step0, self._gen_parity0 = slot, step, self.parity

    def _close_generation(self):
        """Every launch of the current generation is enqueued."""
        self._gen.open, self._gen.dirty = False, True

    def _retire(self, slot):
        """The launches that wrote to ``slot`` have run: fold its counters (launches on the full list, refreshes of
        the inner list) into the integrator's totals."""
        self.integ.full_list_launches = getattr(self.integ, "full_list_launches", 0) + int(slot.host[6])
        self.integ.prune_passes = getattr(self.integ, "prune_passes", 0) + int(slot.host[7])
        slot.open = slot.dirty = False

    def _retire_all(self):
        """After a wait for the stream: every closed generation has run."""
        for slot in self.integ.__dict__.get("_gen_ring", []):
            if slot.dirty and not slot.open:
                self._retire(slot)

    def _x_cur(self) -> torch.Tensor:
        return self.x_a if self.parity == 0 else self.x_b

    def _unpack(self, t: torch.Tensor) -> torch.Tensor:
        out = ops.fresh((self.n, 3), dtype=torch.float32, device=self.dev)
        ops.check(
            ops._lib.load().chx_permute_unpack(ops._ptr(t), t.shape[1], ops._ptr(self.perm), self.n, ops._ptr(out), ops._stream()),
            "chx_permute_unpack",
        )
        return out

    def state_user(self):
        """(positions, velocities) in user order, one launch."""
        x3 = ops.fresh((self.n, 3), dtype=torch.float32, device=self.dev)
        v3 = ops.fresh((self.n, 3), dtype=torch.float32, device=self.dev)
        ops.check(
            ops._lib.load().chx_unpack_langevin(ops._ptr(self._x_cur()), ops._ptr(self.v), ops._ptr(self.perm), self.n,
                                                ops._ptr(x3), ops._ptr(v3), ops._stream()),
            "chx_unpack_langevin",
        )
        return x3, v3

    def final_state_user(self):
        """(positions, velocities) a finished run() returns: the copy enqueued behind the final kick when there is one."""
        cached, self._final_state = getattr(self, "_final_state", None), None
        return cached if cached is not None else self.state_user()

    def positions_user(self) -> torch.Tensor:
        return self._unpack(self._x_cur())

    def velocities_user(self) -> torch.Tensor:
        return self._unpack(self.v)

    def _rebuild(self, step: int = 0, speculative: bool = False):
        """The host half of chiron/integrators.py:241-242 once the device check has fired.

        ``speculative``: do not wait for the build's status block (the reference's sync at
        chiron/neighbors.py:709); the caller keeps enqueuing steps on the new list and calls
        ``_settle`` before it trusts anything computed with it."""
        import os, time

        prof = os.environ.get("CHX_PROFILE_REBUILD") == "1"
        if prof:
            _wait_for_steps_and_stream(device_wide=True)
            t0 = time.perf_counter()
        x_user, v_user = self.state_user()
        keep_events = 1 if self._defer is not None else 0
        self._bury(self.nbr_list)
        spec = speculative and _SPECULATIVE_REBUILD and self.nbr_list._build_speculative(
            x_user, publish=False, pack=self.pack_rider(v_user), ctrl=self.ctrl, ctrl_keep=keep_events,
            half_rows=_RUN_HALF_ROWS)
        if not spec:
            ops.steps_join()  # (the synchronous build reads its status on the host)
            self.nbr_list._adopt_positions = True  # x_user is a fresh tensor nobody else holds
            try:
                self.nbr_list.build(x_user, self.nbr_list.box_vectors)
            finally:
                self.nbr_list._adopt_positions = False
        # (carried-over list: the device's event counters carry on; an event it booked for this very step is the one
        # being carried out here)
        self._load(x_user, v_user, step, ctrl_keep=keep_events, packed=bool(spec))
        self._unsettled = (x_user, v_user, step, self.kicks_next) if spec else None
        self.n_rebuilds += 1
        self._host_events.append(step)
        if prof:
            _wait_for_steps_and_stream(device_wide=True)
            self.integ.rebuild_seconds = getattr(self.integ, "rebuild_seconds", 0.0) + time.perf_counter() - t0

    def _settle(self) -> bool:
        """Validate the last speculative rebuild.  False: its capacity assumptions did not hold; the
        list has been rebuilt the synchronous way and the state reloaded at the rebuild step, and
        whatever was enqueued since must be redone from there (``self._unsettled_step``)."""
        pend = getattr(self, "_unsettled", None)
        if pend is None:
            return True
        self._unsettled = None
        # (the list that was built: the caller's, or -- force list renewed on its own -- one of the private builders)
        owner = pend[4] if len(pend) > 4 else self.nbr_list
        if self.mode == "fused":
            # the build's status block is published by the build's summary launch; the SECOND step launch of the
            # generation behind the build has begun => it is there.  Otherwise wait for the queue and the stream.
            if int(self._gen.host[0]) < self._gen_step0 + 1:  # (+ 1: the generation's first launch may run BESIDE the
                # launch that publishes the block: chx_aql_allow_overlap.)  Not there yet by that token: wait for the
                # block itself -- the build's last store -- rather than for everything queued behind the build (round 6:
                # a full join here also hid, under the queue's debug delays, a join that was missing elsewhere)
                if not owner._wait_for_status():
                    _wait_for_steps_and_stream()
            ok = owner._confirm_build(wait=False)
        else:
            ok = owner._confirm_build()
        if ok:
            return True
        x_user, v_user, step, kicks = pend[:4]
        _wait_for_steps_and_stream(device_wide=True)
        if self._defer is not None:
            # whatever ran on the rejected list is void, its reference events included: the device log tells how many
            # rebuilds the reference had made BEFORE this step (the host-made ones are not in it)
            self._restore_event_count(step)
        if len(pend) > 4:
            # (a renewal of the force list alone: the reference's bookkeeping was none of its business.  The counters are
            # back where they were (above); an event the launch before `step` had booked for `step` -- its criterion,
            # chiron/integrators.py:241, on the very arrays saved here -- is booked again)
            self._load(x_user, v_user, step, owner=owner, ref_user=pend[5], ctrl_keep=1)
            if step > 0 and bool(ops.nl_check(x_user, pend[5], self.nbr_list._box_h, self._skin_now).item()):
                self.ctrl[10 + (step & 1)].fill_(step)
        else:
            self._load(x_user, v_user, step, ctrl_keep=1 if self._defer is not None else 0)
        self.kicks_next = kicks
        self._unsettled_step = step
        return False

    def _restore_event_count(self, step: int):
        """Rewind the device's reference-event bookkeeping to the state before launch ``step`` (stream idle)."""
        host = self.ctrl.cpu()
        count = int(host[8])
        if count > _EV_LOG_CAP:
            raise RuntimeError("cannot rewind the rebuild-event log: more events than it holds")
        log = self.ev_log[:count].cpu().tolist()
        kept = [t for t in log if t < step]
        fix = torch.tensor([len(kept), kept[-1] if kept else -1], dtype=torch.int32, device=self.dev)
        self.ctrl[8:10].copy_(fix)

    # -- fused neighbor-list path, bounded launch queue --------------------------------------------
    def _enqueue_window(self, s0: int, n: int, unthrottled: bool = False) -> int:
        even, odd = (self.x_a, self.x_b) if self.parity == 0 else (self.x_b, self.x_a)
        window = (1 << 20) if unthrottled else max(1, int(self.integ.launch_window))
        sent = ops.langevin_fused_run(self.params, even, odd, s0, n, self.kicks_next, window)
        if sent:
            self.parity ^= sent & 1
            self.kicks_next = 2
        return sent

    def _enqueue_kick(self, step: int, with_state: bool = False):
        """Force + trailing half-kick (chiron/integrators.py:195) as launch ``step``: a no-op if the skin check
        has fired by then, like every launch of the generation.  ``with_state``: the launch also writes positions and
        velocities in user order -- what run() returns -- instead of a copy-out launch behind it (void, and taken
        again, in the rare case that the kick turns out to have been a no-op)."""
        even, odd = (self.x_a, self.x_b) if self.parity == 0 else (self.x_b, self.x_a)
        p = self.params
        if with_state:
            f32 = dict(dtype=torch.float32, device=self.dev)
            x3, v3 = ops.fresh((self.n, 3), **f32), ops.fresh((self.n, 3), **f32)
            # (list carry-over: the reference positions the caller's list will be rebuilt from, should anybody read it)
            r3 = ops.fresh((self.n, 3), **f32) if self._carried is not None else None
            p.final_x3, p.final_v3 = x3.data_ptr(), v3.data_ptr()
            p.final_ref3 = r3.data_ptr() if r3 is not None else None
        try:
            ops.langevin_fused_chunk(p, even, odd, step, 1, 1, 0)
        finally:
            p.final_x3 = p.final_v3 = p.final_ref3 = None
        self._final_state = (x3, v3) if with_state else None
        self._final_ref = r3 if with_state else None
        work, self._host_tail = getattr(self, "_host_tail", None), None
        if work is not None:
            work()  # (see LangevinIntegrator.run: the launches above are still queued)
        if with_state:
            # (taken away before it is called, like _host_tail: the closure holds this object, which must not hold it back
            # -- a reference cycle keeps a run's arrays, the force list among them, alive until the cycle collector comes by)
            pre, self._before_last_wait = self.__dict__.get("_before_last_wait"), None
            if pre is not None:
                pre()
            # A speculative build still unconfirmed (a run shorter than the distance _advance_windowed settles at): its
            # status block is on the host once the SECOND step launch of the generation behind it has begun -- the first may
            # run beside the build's summary launch, which publishes the block (chx_aql_allow_overlap); the second carries
            # the barrier bit and starts behind both.  Confirmed here, with the device still working, not behind the wait;
            # a block that rejects the build -- or may not have arrived: a run of one or two steps -- is left to _settle()
            # behind the wait, which then does what it always did.
            pend = self.__dict__.get("_unsettled")
            if pend is not None and self.mode == "fused" and int(self._gen.host[0]) >= self._gen_step0 + 1:
                owner = pend[4] if len(pend) > 4 else self.nbr_list
                if owner._build_status_good():
                    self._settle()
        self._wait_for_launches()

    def _wait_for_launches(self):
        """Host wait until every launch of the run so far has run.  Through the step queue (the default) the step
        launches are the last thing enqueued and the join is the whole wait -- the stream holds nothing behind the marker
        that let them start; with plain HIP launches (CHX_AQL=0, or no queue) an event on the stream is."""
        if _skip_join_hook():
            return
        ops.steps_join()
        self.integ._graveyard = None  # (everything that read the buried arrays has run)
        if not ops.step_queue_on():
            ev = torch.cuda.Event()
            ev.record()  # (the last thing on the stream: a caller's device-wide synchronize then finds nothing pending)
            ev.synchronize()

    def _advance_windowed(self, done: int, stop: int, final: bool = False) -> int:
        """Steps [done, stop) through ``chx_langevin_fused_run``: the C loop keeps at most ``launch_window``
        launches queued ahead of the device and returns early when it sees the halt word, so a rebuild starts
        behind a handful of no-op launches."""
        chunk = self.integ.chunk_size
        enq = done  # first step not enqueued yet
        self._kick_sent = False
        while True:
            while enq < stop:
                c = min(chunk, stop - enq)
                # The last launches of a segment go out without throttling: the window exists to keep the no-op launches
                # behind a halt few (a list lasts ~200 steps, so one segment tail in seven holds a halt, which then
                # drains up to `chunk` no-ops instead of ~8: +50 us), while a throttled tail makes EVERY segment end with
                # the host asleep in the window and the closing launches -- kick, copy-out -- enqueued late, behind an
                # idle device (measured, round 4: 7-13 us gaps in front of each at the end of a 20-step run()).
                sent = self._enqueue_window(enq, c, unthrottled=(stop - enq <= chunk))
                enq += sent
                work = getattr(self, "_host_tail", None)
                if work is not None and sent > 0:
                    # host work of the run's END that does not depend on the steps (the key and the box the returned
                    # state carries) is done here, with a queue of launches in front of the device
                    self._host_tail = None
                    work()
                if sent < c:
                    break  # the launch loop saw the halt word
                pend = getattr(self, "_unsettled", None)
                if pend is not None and enq - pend[2] >= max(chunk, self.integ.launch_window + 1):
                    # (that many launches after a speculative rebuild its status block has long arrived)
                    if not self._settle():
                        enq = self._unsettled_step
            if enq == stop:
                if final and not self._kick_sent:
                    self._enqueue_kick(stop, with_state=True)  # (waits: everything up to here has run)
                    self._kick_sent = True
                else:
                    self._wait_for_launches()
            if not self._settle():
                # the speculative list these launches ran on was rejected (whatever they halted on or produced is
                # void): the state is back at the rebuild step, on the list the synchronous build made
                enq = self._unsettled_step
                self._kick_sent = False
                self._final_state = None
                continue
            halt = int(self._gen.host[3])
            if halt > stop:
                break
            self._final_state = None
            # the skin check fired while producing x_halt: steps < halt ran, the launches queued behind are no-ops
            # and the rebuild is enqueued behind them
            self._kick_sent = False
            # (diagnostic: launches of steps >= halt that were already enqueued -- what the device drains as no-ops)
            self.integ.noop_launches = getattr(self.integ, "noop_launches", 0) + max(0, enq - halt)
            if halt == 0:
                # (only a carried force list that does not cover x0 halts launch 0: nothing has run, the run's first
                # step still owes its single leading kick)
                self.kicks_next = 1
                self.integ.n_carry_rejected = getattr(self.integ, "n_carry_rejected", 0) + 1
            self.parity = (self._gen_parity0 + halt - self._gen_step0) & 1
            self._close_generation()
            self._halt_rebuild(halt, speculative=True)
            enq = halt
        return stop

    def report_list(self):
        """The list a report's energy is evaluated with (chiron/integrators.py:278): the caller's list -- rebuilt at
        every reference event, as in the reference -- or, on a carried-over force list, that list's owner (same cutoff,
        same skin, valid for the current positions)."""
        return self.nbr_list if self._defer is None else self._owner

    def finish(self):
        """End of ``run()`` (the stream is idle): the reference's rebuild count, and -- carried-over list -- the public
        list, which the reference would have rebuilt at every event.  Here it is rebuilt once, from the positions of
        the LAST event, and only when somebody reads it (NeighborListNsqrd._defer_build)."""
        self.rebuild_log = (list(self._host_events), None)
        d = self._defer
        if d is None:
            return
        gen = getattr(self, "_gen", None)
        n_dev = int(gen.host[8]) if gen is not None else 0
        # list carry-over: the run-start build (chiron/integrators.py:169) was not carried out on the caller's list
        missed = 1 if self._carried is not None else 0
        if n_dev <= 0 and not missed:
            return
        if n_dev > _EV_LOG_CAP:
            log.warning("more reference rebuild events in one run() than the device log holds; rebuild_steps is truncated")
        last_dev = -1
        if n_dev > 0:
            self.n_rebuilds += n_dev
            self.rebuild_log = (list(self._host_events), self.ev_log[:min(n_dev, _EV_LOG_CAP)].clone())
            last_dev = int(gen.host[9])
        last_host = self._host_events[-1] if self._host_events else -1
        if last_dev > last_host or (missed and last_host < 0):
            # (ref4: the positions of the last event -- or x0, the run-start build's, when there was none)
            ref_user, self._final_ref = getattr(self, "_final_ref", None), None
            self.nbr_list._defer_build(ref_user if ref_user is not None else self._unpack(self.ref4), n_dev + missed)
        else:  # the host's own rebuild was the latest: the public list is current
            self.nbr_list.n_builds += n_dev + missed
        self.integ.n_internal_halts = getattr(self.integ, "n_internal_halts", 0) + d.n_internal_halts

    def offer_carry(self, state) -> None:
        """End of ``run()``: leave the force list the run ended on, and the identity of the positions it returned, for
        a next ``run()`` that continues from exactly there (``LangevinIntegrator._take_carry``)."""
        self.integ._carry = None
        if not (self.integ.list_carry and self.mode == "fused" and self.is_verlet and self.fl is not None
                and getattr(self, "_unsettled", None) is None):
            return
        lst, x = self.nbr_list, state.positions
        self.integ._carry = dict(owner=self._owner, public=lst, x=x, x_version=x._version, cs_md=lst._cs_md, skin=lst._skin,
                                 box_token=getattr(lst, "_box_token", None))

    def _halt_rebuild(self, halt: int, speculative: bool):
        """The halt word fired at ``halt``.  Ordinarily that IS the reference's rebuild.  On a carried-over force
        list it means the list has run out of its own guarantee; when its positions are the reference's (the host has
        rebuilt since the run began) the two coincide and the reference's rebuild is carried out here, otherwise only
        the force list is renewed."""
        if self._defer is None or self._snap_is_ref:
            self._rebuild(halt, speculative=speculative)
        else:
            self._rebuild_internal(halt, speculative)

    # -- other paths -------------------------------------------------------
    def _external_force(self):
        """Potentials without a fused kernel: ask the potential (reference integrators.py:171,193)."""
        F = self.potential.compute_force(self.positions_user(), self.nbr_list)
        if isinstance(F, torch.Tensor):
            self.f.copy_(F if self.perm is None else F[self.perm.long()])
        else:
            self.f.fill_(float(F))

    def advance_to(self, done: int, stop: int, final: bool = False) -> int:
        """Run steps [done, stop); returns the new step count (== stop).  ``final``: ``stop`` is the end of the
        run (the fused path then enqueues the trailing kick behind the last chunk, see ``final_kick``)."""
        if self.mode == "fused":
            return self._advance_windowed(done, stop, final)
        if self.mode == "pairs":
            # all-pairs force + BAOAB in one launch per step (k_ap_step); no skin check exists on this
            # path (PairListNsqrd.check only looks at N), so nothing halts and nothing is read back
            chunk = self.integ.chunk_size
            while done < stop:
                c = min(chunk, stop - done)
                even, odd = (self.x_a, self.x_b) if self.parity == 0 else (self.x_b, self.x_a)
                ops.langevin_fused_chunk(self.params, even, odd, done, c, self.kicks_next, 1)
                self.parity ^= c & 1
                self.kicks_next = 2
                done += c
            return done
        # external-force path: one step at a time, like the reference
        while done < stop:
            ops.langevin_chunk(self.params, done, 1, self.kicks_next)
            self.kicks_next = 2
            if self.is_verlet and int(self.ctrl.cpu()[0]) != 0:
                self._rebuild()
            self._external_force()
            done += 1
        return done

    def final_kick(self, n_steps: int):
        """Trailing half-kick of the last step (chiron/integrators.py:195)."""
        if self.mode == "fused":
            if not getattr(self, "_kick_sent", False):
                self._enqueue_kick(n_steps)
            if int(self._gen.host[3]) <= n_steps:
                # the skin check fired in the run's very last step and the kick ran as a no-op: rebuild exactly as
                # _advance_windowed does, then kick
                self._final_state = None
                self.parity = (self._gen_parity0 + n_steps - self._gen_step0) & 1
                self._close_generation()
                self._halt_rebuild(n_steps, speculative=False)
                self._enqueue_kick(n_steps)
            self._kick_sent = False
            self._close_generation()
            self._retire_all()  # (_enqueue_kick waited for the stream)
        elif self.mode == "pairs":
            even, odd = (self.x_a, self.x_b) if self.parity == 0 else (self.x_b, self.x_a)
            ops.langevin_fused_chunk(self.params, even, odd, n_steps, 1, 1, 0)  # force + one kick, positions stay
        else:
            ops.kick(self.v, self.f, self.mass, self.half_dt)
