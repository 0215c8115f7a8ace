"""MCMC moves and sampler: same classes, arguments and bookkeeping as chiron/mcmc.py.

``MCMCMove`` (:11-88), ``LangevinDynamicsMove`` (:91-199), ``MCMove`` (:202-548),
``MonteCarloDisplacementMove`` (:551-787), ``MonteCarloBarostatMove`` (:790-1009),
``MoveSchedule`` (:1036-1071), ``MCMCSampler`` (:1074-1155).  Proposals, wrapping,
neighbor-list checks/rebuilds and energies are HIP kernels; the accept/reject
decision and the statistics are host logic exactly as in the reference.
"""
from __future__ import annotations

import copy
import ctypes as C
import math
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops, unit
from .rng import JaxKey, PRNGKey
from .rng import uniform as host_uniform
from .states import SamplerState, ThermodynamicState

# Inner skin (nm) of the inner list the device-resident displacement loop walks between refreshes (include/chx.h:
# chx_mc_t::fli; same energies and decisions with or without it, bit for bit).  CHX_MC_INNER_SKIN=0 turns it off.
_MC_INNER_SKIN = float(os.environ.get("CHX_MC_INNER_SKIN", "0.1"))


class MCMCMove:
    def __init__(self, number_of_moves: int, reporter=None, report_interval: Optional[int] = 100):
        self.number_of_moves = number_of_moves
        self.reporter = reporter
        self.report_interval = report_interval
        self._move_iteration = 0
        self._number_of_attempts_made = 0
        if self.reporter is not None:
            assert self.report_interval is not None

    def update(self, sampler_state, thermodynamic_state, nbr_list=None):
        raise NotImplementedError

    @property
    def number_of_attemps_made(self):
        return self._number_of_attempts_made


class LangevinDynamicsMove(MCMCMove):
    """chiron/mcmc.py:91-199"""

    def __init__(
        self,
        timestep: unit.Quantity = 1.0 * unit.femtoseconds,
        collision_rate: unit.Quantity = 1.0 / unit.picoseconds,
        refresh_velocities: bool = False,
        reporter=None,
        report_interval: int = 100,
        number_of_steps: int = 1_000,
        save_traj_in_memory: bool = False,
    ):
        super().__init__(number_of_moves=number_of_steps, reporter=reporter, report_interval=report_interval)
        self.timestep = timestep
        self.collision_rate = collision_rate
        self.save_traj_in_memory = save_traj_in_memory
        self.traj = []
        from .integrators import LangevinIntegrator

        self.integrator = LangevinIntegrator(
            timestep=self.timestep,
            collision_rate=self.collision_rate,
            refresh_velocities=refresh_velocities,
            report_interval=report_interval,
            reporter=reporter,
            save_traj_in_memory=save_traj_in_memory,
        )

    def update(self, sampler_state, thermodynamic_state, nbr_list=None):
        assert isinstance(sampler_state, SamplerState), f"Sampler state must be SamplerState, not {type(sampler_state)}"
        assert isinstance(
            thermodynamic_state, ThermodynamicState
        ), f"Thermodynamic state must be ThermodynamicState, not {type(thermodynamic_state)}"
        updated_sampler_state, updated_nbr_list = self.integrator.run(
            thermodynamic_state=thermodynamic_state,
            sampler_state=sampler_state,
            number_of_steps=self.number_of_moves,
            nbr_list=nbr_list,
        )
        self._number_of_attempts_made += self.number_of_moves
        if self.save_traj_in_memory:
            self.traj.append(self.integrator.traj)
            self.integrator.traj = []
        self._move_iteration += 1
        return updated_sampler_state, thermodynamic_state, updated_nbr_list


def langevin_moves_update_batch(moves, sampler_states, thermodynamic_states, nbr_lists):
    """``moves[r].update(sampler_states[r], thermodynamic_states[r], nbr_lists[r])`` for every r -- each a
    ``LangevinDynamicsMove`` with the same number of steps -- through ``integrators.run_batch``: one launch per step for all
    replicas (bit-identical states; serial calls when the batch does not apply).  Returns a list of the moves' return
    triples."""
    from .integrators import run_batch

    steps = moves[0].number_of_moves
    out = run_batch([m.integrator for m in moves], sampler_states, thermodynamic_states, steps, nbr_lists)
    result = []
    for m, (state, lst), thermo in zip(moves, out, thermodynamic_states):
        m._number_of_attempts_made += steps
        if m.save_traj_in_memory:
            m.traj.append(m.integrator.traj)
            m.integrator.traj = []
        m._move_iteration += 1
        result.append((state, thermo, lst))
    return result


class MCMove(MCMCMove):
    """chiron/mcmc.py:202-548"""

    def __init__(
        self,
        number_of_moves: int,
        reporter,
        report_interval: int = 1,
        autotune: bool = False,
        autotune_interval: int = 100,
        acceptance_method: str = "Metropolis-Hastings",
    ) -> None:
        super().__init__(number_of_moves=number_of_moves, reporter=reporter, report_interval=report_interval)
        self.acceptance_method = acceptance_method
        self.reset_statistics()
        self.autotune = autotune
        self.autotune_interval = autotune_interval

    def update(self, sampler_state, thermodynamic_state, nbr_list=None):
        """chiron/mcmc.py:243-306"""
        self._current_reduced_potential = None
        for i in range(self.number_of_moves):
            sampler_state, thermodynamic_state, nbr_list = self._step(sampler_state, thermodynamic_state, nbr_list)
            self._number_of_attempts_made += 1
            if hasattr(self, "reporter"):
                if self.reporter is not None:
                    if self._number_of_attempts_made % self.report_interval == 0:
                        self._report(
                            i,
                            self._move_iteration,
                            self._number_of_attempts_made,
                            self.n_accepted / self.n_proposed,
                            sampler_state,
                            thermodynamic_state,
                            nbr_list,
                        )
            if self.autotune:
                if self._number_of_attempts_made % self.autotune_interval == 0 and self._number_of_attempts_made > 0:
                    self._autotune()
        self._move_iteration += 1
        return sampler_state, thermodynamic_state, nbr_list

    def _report(self, step, iteration, number_of_attempts_made, acceptance_probability, sampler_state,
                thermodynamic_state, nbr_list=None):
        raise NotImplementedError

    def _autotune(self):
        raise NotImplementedError

    def _propose(self, current_sampler_state, current_thermodynamic_state, current_reduced_potential,
                 current_nbr_list=None):
        raise NotImplementedError

    def _step(self, current_sampler_state, current_thermodynamic_state, current_nbr_list=None):
        """chiron/mcmc.py:357-463"""
        if self._current_reduced_potential is None:
            current_reduced_potential = current_thermodynamic_state.get_reduced_potential(
                current_sampler_state, current_nbr_list
            )
            self._current_reduced_potential = current_reduced_potential
        else:
            current_reduced_potential = self._current_reduced_potential

        (
            proposed_sampler_state,
            proposed_thermodynamic_state,
            proposed_reduced_potential,
            log_proposal_ratio,
            proposed_nbr_list,
        ) = self._propose(
            current_sampler_state, current_thermodynamic_state, current_reduced_potential, current_nbr_list
        )

        if math.isnan(float(proposed_reduced_potential)):  # :428-429
            decision = False
        else:
            decision = self._accept_or_reject(
                log_proposal_ratio, proposed_sampler_state.new_PRNG_key, acceptance_method=self.acceptance_method
            )
        self._update_statistics(decision)

        if decision:
            self._current_reduced_potential = proposed_reduced_potential
            return proposed_sampler_state, proposed_thermodynamic_state, proposed_nbr_list
        else:
            # the rejected branch still advances the PRNG key (:459-461)
            current_sampler_state._current_PRNG_key = proposed_sampler_state._current_PRNG_key
            return current_sampler_state, current_thermodynamic_state, current_nbr_list

    def _update_statistics(self, decision):
        if decision:
            self.n_accepted += 1
        self.n_proposed += 1

    @property
    def statistics(self):
        return dict(n_accepted=self.n_accepted, n_proposed=self.n_proposed)

    @statistics.setter
    def statistics(self, value):
        self.n_accepted = value["n_accepted"]
        self.n_proposed = value["n_proposed"]

    def reset_statistics(self):
        self.n_accepted = 0
        self.n_proposed = 0

    def _accept_or_reject(self, log_proposal_ratio, key, acceptance_method):
        """chiron/mcmc.py:531-548: the uniform is drawn even when the move is downhill."""
        if acceptance_method == "Metropolis-Hastings":
            compare_to = host_uniform(key)
            log_proposal_ratio = float(log_proposal_ratio)
            if -log_proposal_ratio <= 0.0 or compare_to < math.exp(log_proposal_ratio):
                return True
            else:
                return False


class MonteCarloDisplacementMove(MCMove):
    """All particles (or ``atom_subset``) displaced by N(0, sigma) at once (chiron/mcmc.py:551-787)."""

    def __init__(
        self,
        displacement_sigma=1.0 * unit.nanometer,
        number_of_moves: int = 100,
        atom_subset: Optional[List[int]] = None,
        report_interval: int = 1,
        reporter=None,
        autotune: bool = False,
        autotune_interval: int = 100,
        acceptance_method="Metropolis-Hastings",
    ):
        super().__init__(
            number_of_moves=number_of_moves,
            reporter=reporter,
            report_interval=report_interval,
            autotune=autotune,
            autotune_interval=autotune_interval,
            acceptance_method=acceptance_method,
        )
        self.displacement_sigma = displacement_sigma
        self.atom_subset = atom_subset
        self.atom_subset_mask = None
        self.inner_list_refreshes = 0  # launches of the device loop that renewed their inner list (diagnostics)
        self._dev_ws = None            # the device loop's arrays (``_device_workspace``)

    # trials enqueued per host round trip of the device-resident loop
    _DEVICE_CHUNK = 256

    def update(self, sampler_state, thermodynamic_state, nbr_list=None):
        """chiron/mcmc.py:243-306.  With an LJ potential on a built Verlet list the whole trial
        loop runs on the device (``_update_device``); every other combination takes the
        reference-shaped host loop of ``MCMove.update`` (one kernel sequence per trial)."""
        if self._device_loop_applies(sampler_state, thermodynamic_state, nbr_list):
            return self._update_device(sampler_state, thermodynamic_state, nbr_list)
        return super().update(sampler_state, thermodynamic_state, nbr_list)

    def _device_loop_applies(self, sampler_state, thermodynamic_state, nbr_list) -> bool:
        from .neighbors import NeighborListNsqrd
        from .potential import LJPotential

        if os.environ.get("CHX_MC_HOST_LOOP", "0") == "1":
            return False
        if isinstance(sampler_state._current_PRNG_key, JaxKey):
            return False  # the device-resident chain draws from Philox; the threefry stream takes the host loop
        cls = type(self)
        if (cls._propose is not MonteCarloDisplacementMove._propose or cls._step is not MCMove._step
                or cls._accept_or_reject is not MCMove._accept_or_reject
                or cls._report is not MonteCarloDisplacementMove._report
                or cls._autotune is not MonteCarloDisplacementMove._autotune
                or cls._update_statistics is not MCMove._update_statistics):
            return False  # a subclass changed the trial or its bookkeeping: honour it (host loop calls them all)
        if self.acceptance_method != "Metropolis-Hastings" or self.number_of_moves <= 0:
            return False
        pot = getattr(thermodynamic_state, "potential", None)
        if type(pot) is not LJPotential or not isinstance(nbr_list, NeighborListNsqrd) or not nbr_list.is_built:
            return False
        if not sampler_state.positions.is_cuda or sampler_state.number_of_particles != nbr_list.ref_positions.shape[0]:
            return False
        if nbr_list.cutoff.value_in_unit_system(unit.md_unit_system) != pot.cutoff:
            return False  # the host loop raises the reference's error
        # the proposal wraps in the sampler state's box, the energy uses the list's: they must agree
        if nbr_list.space.periodic and sampler_state.box_vectors is None:
            return False
        b0, b1 = ops.make_box(sampler_state.box_vectors, nbr_list.space.periodic), nbr_list._box_h
        return (b0.periodic, b0.lx, b0.ly, b0.lz) == (b1.periodic, b1.lx, b1.ly, b1.lz)

    def _subset_mask(self, sampler_state):
        if self.atom_subset is not None and self.atom_subset_mask is None:
            # per-particle 0/1 mask (the reference's (N,3)*(N,) product only broadcasts for N in {1,3}; :717-722, :743)
            mask = np.zeros(sampler_state.number_of_particles, dtype=np.float32)
            for atom in self.atom_subset:
                mask[atom] = 1.0
            self.atom_subset_mask = torch.as_tensor(mask, device=sampler_state.positions.device)
        return self.atom_subset_mask if self.atom_subset is not None else None

    def _device_workspace(self, n: int, dev, fl):
        """The device loop's arrays that do not depend on a context's contents: configurations, candidates, energy
        partials, key table with its pinned staging, logs and the inner list's workspace.  Kept on the move, keyed by
        the force list's geometry: a trial that trips the skin check opens a new context (``_device_context``), and with
        a narrow skin that is every few trials -- a pinned host allocation (``hipHostMalloc``) and nine device
        allocations per rebuild otherwise.  The previous context's launches have finished by then (the chunk's state
        is read on the host before anything else happens)."""
        key = (n, str(dev), fl.g, fl.chunks, tuple(fl.fl.shape), self._DEVICE_CHUNK)
        ws = self._dev_ws
        if ws is not None and ws["key"] == key:
            return ws
        rows = self._DEVICE_CHUNK + 1
        ws = {"key": key}
        ws["cur4"] = torch.empty((2, 2, n, 4), dtype=torch.float32, device=dev)
        ws["prop4"] = torch.empty((2, 2, n, 4), dtype=torch.float32, device=dev)
        ws["ws"] = torch.empty((2 * int(_lib.load().chx_lj_ws_count(n)),), dtype=torch.float64, device=dev)
        # the chunk's key chain is worked out on the host by chx_mc_displace_chunk (the chain does not depend on decisions):
        # device table, pinned staging, and the chain key before every trial of the chunk for the host's own use
        ws["keytab"] = torch.empty((rows, 4), dtype=torch.int32, device=dev)
        ws["keytab_h"] = torch.empty((rows, 4), dtype=torch.int32, pin_memory=True)
        ws["chain"] = np.zeros((rows, 2), dtype=np.uint32)
        ws["state"] = torch.zeros((_lib.MC_STATE_WORDS,), dtype=torch.int32, device=dev)
        ws["energy"] = torch.zeros((4,), dtype=torch.float64, device=dev)
        ws["accept_log"] = torch.zeros((self._DEVICE_CHUNK,), dtype=torch.int32, device=dev)
        ws["energy_log"] = torch.zeros((self._DEVICE_CHUNK,), dtype=torch.float32, device=dev)
        # inner list of the trial launches (include/chx.h: chx_mc_t::fli): workspace only, the kernels fill it; the
        # lane lengths start at zero so that no launch can ever walk a length nobody wrote
        if 2 * fl.chunks < 32768:
            ws["fli"] = torch.zeros_like(fl.fl)
            ws["cnti"] = torch.zeros((n, fl.g), dtype=torch.int32, device=dev)
            ws["refi4"] = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self._dev_ws = ws
        return ws

    def __getstate__(self):
        # (copies and pickles of a move do not share or duplicate the device workspace)
        d = dict(self.__dict__)
        d["_dev_ws"] = None
        return d

    def _device_context(self, sampler_state, thermodynamic_state, nbr_list, energy: float, trial: int = 0):
        """Pack the configuration into the force list's internal order and fill ``chx_mc_t``; ``trial`` = index of
        the first trial the device loop will run from this context."""
        pot = thermodynamic_state.potential
        fl = nbr_list.force_list
        dev = sampler_state.positions.device
        n = sampler_state.number_of_particles
        key = sampler_state._current_PRNG_key
        ws = self._device_workspace(n, dev, fl)
        ctx = {"fl": fl, "nbr_list": nbr_list}
        for name in ("cur4", "prop4", "ws", "keytab", "keytab_h", "chain", "state", "energy", "accept_log", "energy_log"):
            ctx[name] = ws[name]
        # configurations and proposal candidates, (parity, copy, n, 4) each (include/chx.h: chx_mc_t): the chain enters at
        # trial `trial`, so the current configuration goes into copy 0 of parity trial & 1
        ctx["cur4"][trial & 1, 0].copy_(fl.pack(sampler_state.positions, 4))
        ctx["ref4"] = fl.pack(nbr_list.ref_positions, 4)
        ctx["key"] = (int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF)
        # two state slots that take turns by trial parity (include/chx.h: chx_mc_displace_chunk): the chain enters the
        # device loop at trial `trial`, so its state goes into slot trial & 1; the halt word is shared (slot 0)
        slot = _lib.MC_SLOT_WORDS * (trial & 1)
        st = np.zeros(_lib.MC_STATE_WORDS, dtype=np.uint32)
        st[_lib.MC_HALT] = 0x7FFFFFFF
        st[slot + _lib.MC_KEY0], st[slot + _lib.MC_KEY1] = key[0], key[1]
        ctx["state"].copy_(torch.from_numpy(st.view(np.int32)))
        ctx["energy"].copy_(torch.from_numpy(np.array([energy, energy, energy, energy], dtype=np.float64)))
        mask = self._subset_mask(sampler_state)
        if thermodynamic_state.beta is None:
            thermodynamic_state.beta = 1.0 / (unit.BOLTZMANN_CONSTANT_kB * (thermodynamic_state.temperature * unit.kelvin))
        beta = float(thermodynamic_state.beta * (unit.Quantity(1.0, unit.kilojoule_per_mole) / unit.AVOGADRO_CONSTANT_NA))
        p = _lib.McParams()
        p.n, p.g, p.fl_chunks = n, fl.g, fl.chunks
        p.cur4, p.prop4, p.ref4 = ctx["cur4"].data_ptr(), ctx["prop4"].data_ptr(), ctx["ref4"].data_ptr()
        p.perm = None if fl.perm is None else fl.perm.data_ptr()
        p.subset_mask = None if mask is None else mask.data_ptr()
        p.fl, p.deg = fl.fl.data_ptr(), fl.deg.data_ptr()
        p.box = nbr_list._box_h
        p.skin = float(nbr_list.skin.value_in_unit_system(unit.md_unit_system))
        p.sigma, p.epsilon, p.cutoff = pot.sigma, pot.epsilon, pot.cutoff
        p.beta = beta
        p.state, p.energy = ctx["state"].data_ptr(), ctx["energy"].data_ptr()
        p.accept_log, p.energy_log = ctx["accept_log"].data_ptr(), ctx["energy_log"].data_ptr()
        p.ws = ctx["ws"].data_ptr()
        p.keytab, p.keytab_h = ctx["keytab"].data_ptr(), ctx["keytab_h"].data_ptr()
        p.chain_out_h = ctx["chain"].ctypes.data
        if _MC_INNER_SKIN > 0.0 and "fli" in ws:
            ctx["fli"], ctx["cnti"], ctx["refi4"] = ws["fli"], ws["cnti"], ws["refi4"]
            p.fli, p.cnti, p.refi4 = ctx["fli"].data_ptr(), ctx["cnti"].data_ptr(), ctx["refi4"].data_ptr()
            p.inner_skin = _MC_INNER_SKIN
        ctx["params"] = p
        return ctx

    def _after_attempt(self, i, energy, sigma_md):
        """Bookkeeping of one finished trial (chiron/mcmc.py:281-303): report, then autotune."""
        self._number_of_attempts_made += 1
        if self.reporter is not None and self._number_of_attempts_made % self.report_interval == 0:
            self.reporter.report(
                {
                    "step": i,
                    "iteration": self._move_iteration,
                    "number_of_attempts_made": self._number_of_attempts_made,
                    "potential_energy": np.float32(energy),
                    "displacement_sigma": sigma_md,
                    "acceptance_probability": self.n_accepted / self.n_proposed,
                }
            )
        if self.autotune and self._number_of_attempts_made % self.autotune_interval == 0 and self._number_of_attempts_made > 0:
            self._autotune()

    def _update_device(self, sampler_state, thermodynamic_state, nbr_list):
        """The loop of ``MCMove.update`` with the trials enqueued in chunks (chx_mc_displace_chunk).

        Same proposal numbers, energies, decisions, statistics, reports and PRNG chain as the
        host loop; a trial whose proposal violates the list's skin runs through ``_step`` (the
        reference's deepcopy-and-rebuild branch) and the device loop resumes after it.
        """
        sampler_state = copy.deepcopy(sampler_state)  # the caller's object keeps its positions
        pot = thermodynamic_state.potential
        energy = float(pot.compute_energy(sampler_state.positions, nbr_list))  # :392-399
        ctx = self._device_context(sampler_state, thermodynamic_state, nbr_list, energy, 0)
        total = self.number_of_moves
        i = 0
        while i < total:
            m = min(total - i, self._DEVICE_CHUNK)
            if self.autotune:  # sigma is a launch constant: a chunk ends where autotune may change it
                m = min(m, self.autotune_interval - self._number_of_attempts_made % self.autotune_interval)
            sigma_md = float(self.displacement_sigma.value_in_unit_system(unit.md_unit_system))
            ctx["params"].sigma_d = sigma_md
            ctx["params"].k0, ctx["params"].k1 = ctx["key"]  # chain key before trial i
            ops.mc_displace_chunk(ctx["params"], i, m)
            st = ctx["state"].cpu().numpy()  # waits for the chunk
            self.inner_list_refreshes += int(st[_lib.MC_REFRESHES]) - ctx.get("refreshes", 0)
            ctx["refreshes"] = int(st[_lib.MC_REFRESHES])
            halt = int(st[_lib.MC_HALT])
            halted = halt < i + m
            done = (halt if halted else i + m) - i
            ctx["key"] = (int(ctx["chain"][done, 0]), int(ctx["chain"][done, 1]))  # chain key before trial i + done
            if done > 0:
                acc = ctx["accept_log"][:done].cpu().numpy()
                elog = ctx["energy_log"][:done].cpu().numpy()
                for k in range(done):
                    self._update_statistics(bool(acc[k]))
                    self._after_attempt(i + k, elog[k], sigma_md)
                energy = float(elog[done - 1])
                i += done
            if halted:
                # trial `halt` trips the skin check: run it on the reference's rebuild branch
                # (i == halt now: the chain state sits in slot i & 1)
                so = _lib.MC_SLOT_WORDS * (i & 1)
                cur = ctx["cur4"][i & 1, int(st[so + _lib.MC_CUR]) & 1]
                sampler_state.positions = ctx["fl"].unpack(cur)
                sampler_state._current_PRNG_key = PRNGKey(*ctx["key"])
                self._current_reduced_potential = thermodynamic_state.reduced_potential_from_energy(energy, sampler_state)
                sampler_state, thermodynamic_state, nbr_list = self._step(sampler_state, thermodynamic_state, nbr_list)
                energy = float(pot.compute_energy(sampler_state.positions, nbr_list))
                self._after_attempt(i, energy, sigma_md)
                i += 1
                if i < total:
                    ctx = self._device_context(sampler_state, thermodynamic_state, nbr_list, energy, i)
                else:
                    ctx = None
        if ctx is not None:
            st = ctx["state"].cpu().numpy()
            so = _lib.MC_SLOT_WORDS * (i & 1)  # i == total trials done: the chain state sits in slot i & 1
            cur = ctx["cur4"][i & 1, int(st[so + _lib.MC_CUR]) & 1]
            sampler_state.positions = ctx["fl"].unpack(cur)
            sampler_state._current_PRNG_key = PRNGKey(*ctx["key"])
        self._current_reduced_potential = thermodynamic_state.reduced_potential_from_energy(energy, sampler_state)
        self._move_iteration += 1
        return sampler_state, thermodynamic_state, nbr_list

    def _report(self, step, iteration, number_of_attempts_made, acceptance_probability, sampler_state,
                thermodynamic_state, nbr_list=None):
        potential = thermodynamic_state.potential.compute_energy(sampler_state.positions, nbr_list)
        self.reporter.report(
            {
                "step": step,
                "iteration": iteration,
                "number_of_attempts_made": number_of_attempts_made,
                "potential_energy": potential,
                "displacement_sigma": self.displacement_sigma.value_in_unit_system(unit.md_unit_system),
                "acceptance_probability": acceptance_probability,
            }
        )

    def _autotune(self):
        acceptance_ratio = self.n_accepted / self.n_proposed
        if acceptance_ratio > 0.6:
            self.displacement_sigma *= 1.1
        elif acceptance_ratio < 0.4:
            self.displacement_sigma /= 1.1

    def _propose(self, current_sampler_state, current_thermodynamic_state, current_reduced_potential,
                 current_nbr_list=None):
        """chiron/mcmc.py:680-787"""
        self._subset_mask(current_sampler_state)
        key = current_sampler_state.new_PRNG_key
        unitless_displacement_sigma = float(self.displacement_sigma.value_in_unit_system(unit.md_unit_system))
        proposed_sampler_state = copy.deepcopy(current_sampler_state)

        # displacement (+ wrap when a pair structure is present, :751-755) in one kernel
        if current_nbr_list is not None:
            box = ops.make_box(proposed_sampler_state.box_vectors, current_nbr_list.space.periodic)
            if current_nbr_list.space.periodic and proposed_sampler_state.box_vectors is None:
                raise ValueError("box_vectors must be provided for a periodic system")
        else:
            box = ops.make_box(None, False)
        normals = None
        if isinstance(key, JaxKey):  # jrandom.normal(key, shape=(nr_of_atoms, 3)), :731
            normals = ops.jax_normal(key, tuple(current_sampler_state.positions.shape), device=current_sampler_state.positions.device)
        proposed_sampler_state.positions = ops.mc_displace(
            current_sampler_state.positions, unitless_displacement_sigma, box, key, 0,
            self.atom_subset_mask if self.atom_subset is not None else None, normals,
        )

        if current_nbr_list is not None:
            if current_nbr_list.check(proposed_sampler_state.positions):
                proposed_nbr_list = copy.deepcopy(current_nbr_list)
                proposed_nbr_list.build(proposed_sampler_state.positions, proposed_sampler_state.box_vectors)
            else:
                proposed_nbr_list = current_nbr_list
        else:
            proposed_nbr_list = None

        proposed_reduced_potential = current_thermodynamic_state.get_reduced_potential(
            proposed_sampler_state, proposed_nbr_list
        )
        log_proposal_ratio = -proposed_reduced_potential + current_reduced_potential
        return (
            proposed_sampler_state,
            current_thermodynamic_state,
            proposed_reduced_potential,
            log_proposal_ratio,
            proposed_nbr_list,
        )


class MonteCarloBarostatMove(MCMove):
    """Isotropic volume move, atomic scaling (chiron/mcmc.py:790-1009)."""

    def __init__(
        self,
        volume_max_scale=0.01,
        number_of_moves: int = 100,
        report_interval: int = 1,
        reporter=None,
        autotune: bool = False,
        autotune_interval: int = 100,
        acceptance_method="Metropolis-Hastings",
    ):
        super().__init__(
            number_of_moves=number_of_moves,
            reporter=reporter,
            report_interval=report_interval,
            autotune=autotune,
            autotune_interval=autotune_interval,
            acceptance_method=acceptance_method,
        )
        self.volume_max_scale = volume_max_scale

    @staticmethod
    def _volume(box) -> float:
        b = box.detach().cpu().numpy() if isinstance(box, torch.Tensor) else np.asarray(box)
        return float(b[0][0]) * float(b[1][1]) * float(b[2][2])

    def _report(self, step, iteration, number_of_attempts_made, acceptance_probability, sampler_state,
                thermodynamic_state, nbr_list=None):
        potential = thermodynamic_state.potential.compute_energy(sampler_state.positions, nbr_list)
        self.reporter.report(
            {
                "step": step,
                "iteration": iteration,
                "number_of_attempts_made": number_of_attempts_made,
                "potential_energy": potential,
                "volume": self._volume(sampler_state.box_vectors),
                "box_vectors": sampler_state.box_vectors,
                "max_volume_scale": self.volume_max_scale,
                "acceptance_probability": acceptance_probability,
            }
        )

    def _autotune(self):
        acceptance_ratio = self.n_accepted / self.n_proposed
        if acceptance_ratio < 0.25:
            self.volume_max_scale /= 1.1
        elif acceptance_ratio > 0.75:
            self.volume_max_scale = min(self.volume_max_scale * 1.1, 0.3)

    def _propose(self, current_sampler_state, current_thermodynamic_state, current_reduced_potential,
                 current_nbr_list=None):
        """chiron/mcmc.py:913-1009"""
        key = current_sampler_state.new_PRNG_key
        nr_of_atoms = current_sampler_state.number_of_particles
        initial_volume = self._volume(current_sampler_state.box_vectors)
        delta_volume_max = self.volume_max_scale * initial_volume
        # uniform(-1, 1) * dV_max  (:967)
        delta_volume = (2.0 * host_uniform(key) - 1.0) * delta_volume_max
        proposed_volume = initial_volume + delta_volume
        length_scaling_factor = (proposed_volume / initial_volume) ** (1.0 / 3.0)

        proposed_sampler_state = copy.deepcopy(current_sampler_state)
        proposed_sampler_state.positions = ops.scale(current_sampler_state.positions, length_scaling_factor)
        proposed_sampler_state.box_vectors = ops.scale(current_sampler_state.box_vectors, length_scaling_factor)

        proposed_nbr_list = None
        if current_nbr_list is not None:
            proposed_nbr_list = copy.deepcopy(current_nbr_list)
            # scaling the box always invalidates the list (:985-990)
            proposed_nbr_list.build(proposed_sampler_state.positions, proposed_sampler_state.box_vectors)

        proposed_reduced_potential = current_thermodynamic_state.get_reduced_potential(
            proposed_sampler_state, proposed_nbr_list
        )
        log_proposal_ratio = -(proposed_reduced_potential - current_reduced_potential) + nr_of_atoms * math.log(
            proposed_volume / initial_volume
        )
        return (
            proposed_sampler_state,
            current_thermodynamic_state,
            proposed_reduced_potential,
            log_proposal_ratio,
            proposed_nbr_list,
        )


class MoveSchedule:
    """chiron/mcmc.py:1036-1071"""

    def __init__(self, move_schedule: List[Tuple[str, MCMCMove]]) -> None:
        self.move_schedule = move_schedule
        self._validate_sequence()

    def _validate_sequence(self):
        for move_name, move_class in self.move_schedule:
            if not isinstance(move_class, MCMCMove):
                raise ValueError(f"Move {move_name} in the sequence is not available.")


class MCMCSampler:
    """chiron/mcmc.py:1074-1155"""

    def __init__(self, move_set: MoveSchedule):
        self.move = move_set

    def run(self, sampler_state, thermodynamic_state, n_iterations: int = 1, nbr_list=None):
        sampler_state = copy.deepcopy(sampler_state)
        thermodynamic_state = copy.deepcopy(thermodynamic_state)
        nbr_list = copy.deepcopy(nbr_list)
        for iteration in range(n_iterations):
            for move_name, move in self.move.move_schedule:
                sampler_state, thermodynamic_state, nbr_list = move.update(
                    sampler_state, thermodynamic_state, nbr_list
                )
        for _, move in self.move.move_schedule:
            if move.reporter is not None:
                move.reporter.flush_buffer()
        return sampler_state, thermodynamic_state, nbr_list

    @staticmethod
    def batchable(samplers) -> bool:
        """Are these samplers -- one per replica -- schedules of ``LangevinDynamicsMove`` only, the same number of moves
        with the same step counts in each, no reporters?  Then ``run_batch`` can advance their replicas side by side."""
        try:
            schedules = [s.move.move_schedule for s in samplers]
        except AttributeError:
            return False
        if len(samplers) < 2 or len({id(s) for s in samplers}) != len(samplers) or not schedules[0]:
            return False
        for sch in schedules:
            if len(sch) != len(schedules[0]):
                return False
            for (_, mv), (_, ref) in zip(sch, schedules[0]):
                if not (type(mv) is LangevinDynamicsMove and mv.reporter is None and not mv.save_traj_in_memory
                        and mv.number_of_moves == ref.number_of_moves):
                    return False
        return True

    @staticmethod
    def run_batch(samplers, sampler_states, thermodynamic_states, n_iterations: int, nbr_lists):
        """``samplers[r].run(sampler_states[r], thermodynamic_states[r], n_iterations, nbr_lists[r])`` for every r with the
        replicas' Langevin moves stepped side by side (chiron/mcmc.py:1074-1155 per replica; same deep copies, same move
        order).  The caller has checked ``batchable(samplers)``."""
        sampler_states = [copy.deepcopy(s) for s in sampler_states]
        thermodynamic_states = [copy.deepcopy(t) for t in thermodynamic_states]
        nbr_lists = [copy.deepcopy(n) for n in nbr_lists]
        for _ in range(n_iterations):
            for k in range(len(samplers[0].move.move_schedule)):
                moves = [s.move.move_schedule[k][1] for s in samplers]
                out = langevin_moves_update_batch(moves, sampler_states, thermodynamic_states, nbr_lists)
                sampler_states = [o[0] for o in out]
                thermodynamic_states = [o[1] for o in out]
                nbr_lists = [o[2] for o in out]
        return list(zip(sampler_states, thermodynamic_states, nbr_lists))
