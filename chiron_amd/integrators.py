"""BAOAB Langevin integrator behind the reference's ``LangevinIntegrator`` API.

Reference: chiron/integrators.py:13-286.  The Python loop of the reference (one
``jax.grad`` + ~12 XLA dispatches + one host sync per step, :174-205) becomes two
HIP launches per step -- a fused B-B-A-O-A-wrap-skin-check kernel and the pair-force
kernel -- enqueued in chunks from C (``chx_langevin_chunk``).  The skin check that the
reference turns into a Python ``if`` every step (:241) sets a device flag instead;
once it fires the rest of the chunk is a sequence of no-op launches, and the host
rebuilds the neighbor list at exactly the step the reference would have and resumes.
"""
from __future__ import annotations

import copy
import ctypes as C
import logging
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops, unit
from ._device import as_tensor
from ._lib import Box, LangevinParams
from .rng import PRNGKey, fold_in
from .states import SamplerState, ThermodynamicState

log = logging.getLogger("chiron_amd")


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
    ``chunk_size`` (steps enqueued per host round-trip) is the only addition.
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
        self.chunk_size = int(chunk_size)
        self.n_rebuilds = 0  # neighbor-list rebuilds triggered inside run()

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
        replaces the device PRNG for the O step (deterministic parity tests)."""
        from .potential import LJPotential
        from .neighbors import NeighborListNsqrd, PairListNsqrd
        from .utils import get_list_of_mass, initialize_velocities

        potential = thermodynamic_state.potential
        mass = get_list_of_mass(potential.topology)
        self.box_vectors = sampler_state.box_vectors
        self.progress_bar = progress_bar
        temperature = thermodynamic_state.temperature
        x0 = sampler_state.positions

        key = sampler_state.new_PRNG_key  # :124

        kbT_unitless = float((self.kB * temperature).value_in_unit_system(unit.md_unit_system))
        mass_unitless = as_tensor(mass.value_in_unit_system(unit.md_unit_system))
        half_dt, a, b = langevin_constants(self.timestep, self.collision_rate)

        # velocity (re)initialisation rules, :140-160
        if (
            self.refresh_velocities
            or sampler_state._velocities is None
            or sampler_state._velocities.shape[0] != sampler_state.positions.shape[0]
        ):
            sampler_state.velocities = initialize_velocities(temperature, potential.topology, key)
        v0 = sampler_state.velocities

        n = x0.shape[0]
        dev = x0.device
        if nbr_list is not None:
            nbr_list.build_from_state(sampler_state)  # :169

        use_lj_kernels = isinstance(potential, LJPotential) and (
            nbr_list is None or isinstance(nbr_list, (NeighborListNsqrd, PairListNsqrd))
        )
        if use_lj_kernels and nbr_list is not None:
            # same validation as compute_energy (chiron/potential.py:263-270)
            if not nbr_list.is_built:
                raise ValueError("Neighborlist must be built before use")
            lc = nbr_list.cutoff
            if lc is not None and lc.value_in_unit_system(unit.md_unit_system) != potential.cutoff:
                raise ValueError(
                    f"Neighborlist cutoff ({nbr_list.cutoff}) must be the same as the potential cutoff ({potential.cutoff})"
                )

        st = _RunState(self, potential, nbr_list, x0, v0, mass_unitless, kbT_unitless, half_dt, a, b,
                       key, noise, use_lj_kernels)
        st.initial_force()

        step = 0
        kicks_first = 1
        steps_iter = None
        if progress_bar:
            try:
                from tqdm import tqdm

                steps_iter = tqdm(total=number_of_steps)
            except Exception:  # pragma: no cover
                steps_iter = None
        wants_reports = (hasattr(self, "reporter") and self.reporter is not None) or self.save_traj_in_memory
        while step < number_of_steps:
            chunk = min(self.chunk_size, number_of_steps - step)
            if not use_lj_kernels:
                chunk = 1
            if wants_reports:
                # end the chunk on the next reporting step so the state can be read out there
                base = self._move_iteration * number_of_steps
                to_report = (-(step + base)) % self.report_interval
                chunk = min(chunk, to_report + 1)
            done = st.advance(step, chunk, kicks_first)
            if steps_iter is not None:
                steps_iter.update(done)
            step += done
            kicks_first = 2
            last = step - 1
            elapsed_step = last + self._move_iteration * number_of_steps
            if elapsed_step % self.report_interval == 0:  # :197-205
                if hasattr(self, "reporter") and self.reporter is not None:
                    self._report(st.positions_user(), potential, st.nbr_list, last, self._move_iteration, elapsed_step)
                if self.save_traj_in_memory:
                    self.traj.append(st.positions_user())
        if steps_iter is not None:
            steps_iter.close()
        if number_of_steps > 0:
            st.final_kick()

        updated_sampler_state = copy.deepcopy(sampler_state)  # :210-218
        updated_sampler_state.positions = st.positions_user()
        updated_sampler_state.velocities = st.velocities_user()
        updated_sampler_state.current_PRNG_key = fold_in(key, number_of_steps)
        self.n_rebuilds = st.n_rebuilds
        return updated_sampler_state, st.nbr_list

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


class _RunState:
    """Device-resident state of one ``run`` call in the kernels' internal layout."""

    def __init__(self, integ, potential, nbr_list, x0, v0, mass, kT, half_dt, a, b, key: PRNGKey,
                 noise, use_lj_kernels: bool):
        from .neighbors import NeighborListNsqrd, PairListNsqrd

        self.integ = integ
        self.potential = potential
        self.nbr_list = nbr_list
        self.kT, self.half_dt, self.a, self.b = kT, half_dt, a, b
        self.key = key
        self.use_lj = use_lj_kernels
        self.n = x0.shape[0]
        self.dev = x0.device
        self.noise = None if noise is None else as_tensor(noise)
        if self.noise is not None and (self.noise.dim() != 3 or self.noise.shape[1:] != (self.n, 3)):
            raise ValueError(f"noise must have shape (steps, {self.n}, 3), got {tuple(self.noise.shape)}")
        self.mass_user = mass
        self.ctrl = torch.zeros((4,), dtype=torch.int32, device=self.dev)
        self.ws = torch.empty((int(ops._lib.load().chx_lj_ws_count(self.n)),), dtype=torch.float64, device=self.dev)
        self.is_verlet = isinstance(nbr_list, NeighborListNsqrd)
        self.is_pairlist = isinstance(nbr_list, PairListNsqrd)
        self.n_rebuilds = 0
        self._load(x0, v0)

    # -- layout ------------------------------------------------------------
    def _load(self, x_user: torch.Tensor, v_user: torch.Tensor):
        """(Re)pack user-order arrays into the internal order of the current force list."""
        lib = ops._lib.load()
        self.fl = self.nbr_list.force_list if (self.is_verlet and self.use_lj) else None
        perm = self.fl.perm if self.fl is not None else None
        self.perm = perm
        n = self.n

        def pack(src, stride):
            out = torch.empty((n, stride), dtype=torch.float32, device=self.dev)
            ops.check(lib.chx_permute_pack(ops._ptr(src), ops._ptr(perm), n, ops._ptr(out), stride, ops._stream()), "chx_permute_pack")
            return out

        self.x4 = pack(ops._f32(x_user), 4)
        self.v = pack(ops._f32(v_user), 3)
        self.mass = self.mass_user if perm is None else self.mass_user[perm.long()].contiguous()
        self.f = torch.zeros((n, 3), dtype=torch.float32, device=self.dev)
        self.ref4 = None
        if self.is_verlet:
            self.ref4 = pack(self.nbr_list.ref_positions, 4)
        box = Box(0, 0.0, 0.0, 0.0)
        wrap = 0
        skin = 0.0
        if self.nbr_list is not None:
            box = self.nbr_list._box_h
            wrap = 1
            if self.is_verlet:
                skin = float(self.nbr_list.skin.value_in_unit_system(unit.md_unit_system))
        p = LangevinParams()
        p.n = n
        p.g = self.fl.g if self.fl is not None else 1
        p.x4 = self.x4.data_ptr()
        p.v = self.v.data_ptr()
        p.f = self.f.data_ptr()
        p.ref4 = self.ref4.data_ptr() if self.ref4 is not None else None
        p.mass = self.mass.data_ptr()
        p.perm = perm.data_ptr() if perm is not None else None
        p.fl = self.fl.fl.data_ptr() if self.fl is not None else None
        p.deg = self.fl.deg.data_ptr() if self.fl is not None else None
        p.box = box
        p.wrap = wrap
        p.skin = skin
        p.half_dt, p.a, p.b, p.kT = self.half_dt, self.a, self.b, self.kT
        if self.use_lj:
            p.sigma, p.epsilon, p.cutoff = self.potential.sigma, self.potential.epsilon, self.potential.cutoff
            if self.is_pairlist and self.nbr_list.cutoff is None:
                p.cutoff = -1.0
        p.k0, p.k1 = self.key[0], self.key[1]
        p.noise = self.noise.data_ptr() if self.noise is not None else None
        p.noise_step0 = 0
        p.ctrl = self.ctrl.data_ptr()
        p.ws = self.ws.data_ptr()
        p.external_force = 0 if self.use_lj else 1
        self.params = p

    def _unpack(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.n, 3), dtype=torch.float32, device=self.dev)
        ops.check(
            ops._lib.load().chx_permute_unpack(ops._ptr(t), t.shape[1], ops._ptr(self.perm), self.n, ops._ptr(out), ops._stream()),
            "chx_permute_unpack",
        )
        return out

    def positions_user(self) -> torch.Tensor:
        return self._unpack(self.x4)

    def velocities_user(self) -> torch.Tensor:
        return self._unpack(self.v)

    # -- dynamics ----------------------------------------------------------
    def _external_force(self):
        """Potentials without a fused kernel: ask the potential (reference integrators.py:171,193)."""
        F = self.potential.compute_force(self.positions_user(), self.nbr_list)
        if isinstance(F, torch.Tensor):
            self.f.copy_(F if self.perm is None else F[self.perm.long()])
        else:
            self.f.fill_(float(F))

    def initial_force(self):
        if self.use_lj:
            ops.langevin_resume(self.params)
        else:
            self._external_force()

    def _rebuild(self):
        """The host half of chiron/integrators.py:241-242 after the device check fired."""
        x_user, v_user = self.positions_user(), self.velocities_user()
        self.nbr_list.build(x_user, self.nbr_list.box_vectors)
        self.ctrl.zero_()
        self._load(x_user, v_user)
        self.n_rebuilds += 1

    def advance(self, step0: int, n_steps: int, kicks_first: int) -> int:
        """Run up to n_steps; returns how many completed (force included)."""
        ops.langevin_chunk(self.params, step0, n_steps, kicks_first)
        if self.use_lj:
            ctrl = self.ctrl.cpu()  # one host sync per chunk
            if int(ctrl[1]) != 0:
                s = int(ctrl[2])
                self._rebuild()
                ops.langevin_resume(self.params)  # force of step s on the fresh list
                return s - step0 + 1
            return n_steps
        # external-force path: one step at a time, like the reference
        if self.is_verlet and int(self.ctrl.cpu()[0]) != 0:
            self._rebuild()
        self._external_force()
        return n_steps

    def final_kick(self):
        ops.kick(self.v, self.f, self.mass, self.half_dt)
