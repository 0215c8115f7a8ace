"""Multistate sampling: the reference's ``MultiStateSampler`` skeleton and a
one-replica-per-GPU replica-exchange layer.

``MultiStateSampler`` follows chiron/multistate.py:10-742 for what is on the hot
path: ``create`` / ``_allocate_variables`` (:203-309), ``_propagate_replica(s)``
(:414-445, :497-510), ``_compute_energies`` (:512-531), ``_mix_replicas`` /
``_perform_swap_proposals`` (:447-495) and ``run`` (:563-599).  The reference's
swap step is a placeholder that returns the state indices unchanged (:447-460);
that behaviour is the default here (``swap=None``).  MBAR analysis, HDF5/XTC
reporting and minimisation are out of scope.

``ReplicaExchange`` is the multi-GPU path: rank r owns replica r; the
only collective is one all-gather of K potential energies per exchange (RCCL over
xGMI when the backend is ``nccl``), after which every rank takes the same
deterministic neighbour-swap decisions from a shared Philox key and swaps *labels*
(thermodynamic-state indices), rescaling its velocities by sqrt(T_new/T_old).
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import unit
from .rng import PRNGKey, uniform as host_uniform
from .states import SamplerState, ThermodynamicState, calculate_reduced_potential_at_states


# ---------------------------------------------------------------------------
# swap rule shared by the serial and the distributed paths
# ---------------------------------------------------------------------------
def neighbor_swap_decisions(u_kn: np.ndarray, state_of_replica: np.ndarray, iteration: int, key: PRNGKey):
    """Metropolis neighbour swaps on the K x K reduced-potential matrix.

    ``u_kn[r, k]`` = reduced potential of replica r's configuration at state k
    (row r is what chiron/multistate.py:523-531 stores).  State pairs (k, k+1) with
    k = iteration mod 2, +2, ... are attempted; pair (k, k+1) is accepted with
    probability min(1, exp(-(u[a,k+1] + u[b,k] - u[a,k] - u[b,k+1]))) where a, b are
    the replicas currently at k and k+1.  The uniform for pair k at this iteration
    comes from the shared key, so every rank reaches the same decisions.
    Returns (new_state_of_replica, n_proposed[K,K], n_accepted[K,K]).
    """
    K = len(state_of_replica)
    new = np.array(state_of_replica, dtype=np.int64).copy()
    n_prop = np.zeros((K, K), np.int64)
    n_acc = np.zeros((K, K), np.int64)
    replica_at_state = np.empty(K, np.int64)
    replica_at_state[new] = np.arange(K)
    for k in range(iteration % 2, K - 1, 2):
        a, b = int(replica_at_state[k]), int(replica_at_state[k + 1])
        log_p = -(u_kn[a, k + 1] + u_kn[b, k] - u_kn[a, k] - u_kn[b, k + 1])
        n_prop[k, k + 1] += 1
        n_prop[k + 1, k] += 1
        u = host_uniform(key, (iteration << 8) + k)
        if not math.isnan(log_p) and (log_p >= 0.0 or u < math.exp(log_p)):
            new[a], new[b] = k + 1, k
            replica_at_state[k], replica_at_state[k + 1] = b, a
            n_acc[k, k + 1] += 1
            n_acc[k + 1, k] += 1
    return new, n_prop, n_acc


class ReplicaExchange:
    """Replicas sharded over processes/GPUs; temperature (and pressure) ladder exchange.

    K thermodynamic states over W ranks, K a multiple of W: rank r owns the replicas r K/W ... (r + 1) K/W - 1
    (BASELINE's C5 is K = W = 8, one replica per GPU; with fewer cards than states a rank propagates its K/W replicas
    side by side, ``LangevinIntegrator.run_batch``).  Works on any ``torch.distributed`` backend: ``nccl`` (= RCCL) on the
    GPU node, ``gloo`` in the CPU tests.  With ``world_size == 1`` no process group is needed -- but one is used when it
    exists (a one-rank RCCL group runs the very collective the eight-rank job runs).
    """

    def __init__(self, thermodynamic_states: Sequence[ThermodynamicState], key: PRNGKey,
                 rank: Optional[int] = None, world_size: Optional[int] = None, device=None):
        import torch.distributed as dist

        self.states = list(thermodynamic_states)
        self.key = key
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.rank, self.world_size = rank, world_size
        K = len(self.states)
        if K == 0 or K % world_size != 0:
            raise RuntimeError("Number of thermodynamic states must be a (non-zero) multiple of the number of ranks.")
        self.n_states = K
        self.replicas_per_rank = K // world_size
        self.device = device
        self.state_of_replica = np.arange(K, dtype=np.int64)  # multistate.py:277-279
        self.iteration = 0
        self.n_proposed = np.zeros((K, K), np.int64)
        self.n_accepted = np.zeros((K, K), np.int64)
        self.u_kn = np.zeros((K, K))
        self._betas = np.array(
            [float((1.0 / (unit.BOLTZMANN_CONSTANT_kB * s.temperature * unit.AVOGADRO_CONSTANT_NA)).value_in_unit(
                unit.kilojoule_per_mole ** -1)) for s in self.states]
        )
        self._pressures = [s.pressure for s in self.states]

    @property
    def my_replicas(self) -> range:
        """Global indices of the replicas this rank propagates."""
        return range(self.rank * self.replicas_per_rank, (self.rank + 1) * self.replicas_per_rank)

    @property
    def my_state_indices(self) -> List[int]:
        return [int(self.state_of_replica[r]) for r in self.my_replicas]

    @property
    def my_states(self) -> List[ThermodynamicState]:
        return [self.states[k] for k in self.my_state_indices]

    @property
    def my_state_index(self) -> int:
        """(one replica per rank)"""
        return int(self.state_of_replica[self.my_replicas[0]])

    @property
    def my_state(self) -> ThermodynamicState:
        return self.states[self.my_state_index]

    def _reduced_row(self, energy_kj_mol: np.ndarray, volume_nm3: Optional[np.ndarray]) -> np.ndarray:
        """u[r, k] = beta_k (U_r + p_k V_r N_A) -- chiron/states.py:307-325 for T/p ladders."""
        u = energy_kj_mol[:, None] * self._betas[None, :]
        if volume_nm3 is not None:
            for k, p in enumerate(self._pressures):
                if p is not None:
                    pv = float((p * (1.0 * unit.nanometer ** 3) * unit.AVOGADRO_CONSTANT_NA).value_in_unit(unit.kilojoule_per_mole))
                    u[:, k] += self._betas[k] * pv * volume_nm3
        return u

    def gather_energies(self, energy: torch.Tensor) -> np.ndarray:
        """The one collective of the path: all-gather of one fp64 per replica, K/W of them from every rank (the
        reference's energy matrix is fp64, chiron/multistate.py:290-292; the kernels' fp32 energies and host-side fp64
        volumes both travel unrounded).  With RCCL the gather runs on the device tensor in place; ``gloo`` gathers host
        tensors (CPU tests, single-GPU rehearsals)."""
        import torch.distributed as dist

        e = energy.detach().reshape(-1).to(torch.float64)
        if e.numel() != self.replicas_per_rank:
            raise ValueError(f"expected {self.replicas_per_rank} energies (one per replica of this rank), got {e.numel()}")
        if not (dist.is_available() and dist.is_initialized()):
            if self.world_size != 1:
                raise RuntimeError("ReplicaExchange over several ranks needs an initialised process group")
            return e.cpu().numpy()
        if dist.get_backend() == "gloo":
            e = e.cpu()
        out = torch.empty(self.n_states, dtype=torch.float64, device=e.device)
        dist.all_gather_into_tensor(out, e.contiguous())
        return out.cpu().numpy()

    def exchange_many(self, energies: torch.Tensor, volumes_nm3=None):
        """Gather, decide, relabel.  ``energies``: the potential energies of this rank's replicas, in ``my_replicas``
        order.  Returns one (new_state_index, velocity_scale) per replica of this rank."""
        gathered = self.gather_energies(energies)
        vols = None
        if volumes_nm3 is not None:
            vols = self.gather_energies(torch.as_tensor(volumes_nm3, dtype=torch.float64, device=energies.device).reshape(-1))
        self.u_kn = self._reduced_row(gathered, vols)
        old = self.my_state_indices
        new, n_prop, n_acc = neighbor_swap_decisions(self.u_kn, self.state_of_replica, self.iteration, self.key)
        self.state_of_replica = new
        self.n_proposed += n_prop
        self.n_accepted += n_acc
        self.iteration += 1
        out = []
        for o, n_ in zip(old, self.my_state_indices):
            scale = 1.0
            if n_ != o:
                t_old = self.states[o].temperature.value_in_unit(unit.kelvin)
                t_new = self.states[n_].temperature.value_in_unit(unit.kelvin)
                scale = math.sqrt(t_new / t_old)
            out.append((n_, scale))
        return out

    def exchange(self, energy: torch.Tensor, volume_nm3: Optional[float] = None):
        """One replica per rank: returns (new_state_index, velocity_scale) for this rank."""
        if self.replicas_per_rank != 1:
            raise RuntimeError("this rank owns several replicas: use exchange_many")
        return self.exchange_many(energy, None if volume_nm3 is None else [volume_nm3])[0]


# ---------------------------------------------------------------------------
# the reference's serial sampler
# ---------------------------------------------------------------------------
class MultiStateSampler:
    """chiron/multistate.py:10-742 (propagate / energy-matrix / mix skeleton)."""

    def __init__(self, mcmc_sampler, reporter=None, swap: Optional[str] = None, key: Optional[PRNGKey] = None,
                 concurrent_replicas: int = 1, batch_replicas: bool = True):
        # concurrent_replicas > 1 (not in the reference, whose loop is serial): that many replicas are propagated side
        # by side, each by a thread of its own on a HIP stream of its own.  A step of a medium-size system is a
        # latency chain that leaves the card mostly idle (two N = 32768 replicas on one MI355X run at 1.7x the
        # aggregate rate of one), and replicas are independent between mixing steps, so results do not change.
        # batch_replicas (the default; not in the reference either): replicas whose samplers are Langevin moves on LJ Verlet
        # lists are propagated SIDE BY SIDE, one launch per step for all of them (integrators.run_batch: the kernels take a
        # replica index, every replica keeps its own box, temperature, key, lists and rebuild schedule).  Eight N = 32768
        # replicas on one MI355X then run at well over twice the aggregate rate of the serial loop, with bit-identical
        # states.  Where it does not apply (other moves, reporters, CPU tensors) the loop below runs as before.
        self._batch = bool(batch_replicas)
        self._concurrent = max(1, int(concurrent_replicas))
        self._thermodynamic_states = None
        self._sampler_states = None
        self._replica_thermodynamic_states = None
        self._iteration = None
        self._energy_thermodynamic_states = None
        self._n_accepted_matrix = None
        self._n_proposed_matrix = None
        self._nbr_lists = None
        self._reporter = reporter
        self._mcmc_sampler = copy.deepcopy(mcmc_sampler)
        self._swap = swap  # None reproduces the reference (no swaps); "neighbors" enables the Metropolis rule
        self._key = key or PRNGKey(0x5EED, 0)

    @property
    def number_of_thermodynamic_states(self) -> int:
        return 0 if self._thermodynamic_states is None else len(self._thermodynamic_states)

    @property
    def number_of_replicas(self) -> int:
        return 0 if self._sampler_states is None else len(self._sampler_states)

    @property
    def iteration(self):
        return self._iteration

    @property
    def mcmc_sampler(self):
        return copy.deepcopy(self._mcmc_sampler)

    @property
    def sampler_states(self) -> Optional[List[SamplerState]]:
        return None if self._sampler_states is None else copy.deepcopy(self._sampler_states)

    @property
    def is_periodic(self):
        if self._sampler_states is None:
            return None
        return self._sampler_states[0].box_vectors is not None

    def create(self, thermodynamic_states, sampler_states, nbr_lists):
        """chiron/multistate.py:203-238"""
        if len(thermodynamic_states) != len(sampler_states):
            raise RuntimeError("Number of thermodynamic states and sampler states must be equal.")
        self._allocate_variables(thermodynamic_states, sampler_states, nbr_lists)
        if self._reporter is None:
            from .reporters import MultistateReporter

            self._reporter = MultistateReporter()

    def _allocate_variables(self, thermodynamic_states, sampler_states, nbr_lists) -> None:
        """chiron/multistate.py:240-309"""
        from .mcmc import MCMCSampler

        self._thermodynamic_states = copy.deepcopy(thermodynamic_states)
        self._sampler_states = copy.deepcopy(sampler_states)
        self._nbr_lists = copy.deepcopy(nbr_lists)
        assert len(self._thermodynamic_states) == len(self._sampler_states)
        assert len(self._thermodynamic_states) == len(self._nbr_lists)
        for nbr_list, state in zip(self._nbr_lists, self._sampler_states):
            if nbr_list is not None:
                # (a speed hint: these replicas share one card -- the same in the serial and the batched loop, so that the
                # two stay bit-identical)
                if hasattr(nbr_list, "set_replicas_sharing_card") and getattr(state.positions, "is_cuda", False):
                    nbr_list.set_replicas_sharing_card(len(self._nbr_lists))
                nbr_list.build(state.positions, state.box_vectors)
        K = len(thermodynamic_states)
        self._replica_thermodynamic_states = np.arange(K, dtype=int)
        self._n_accepted_matrix = np.zeros([K, K], np.int64)
        self._n_proposed_matrix = np.zeros([K, K], np.int64)
        self._energy_thermodynamic_states = np.zeros([self.number_of_replicas, K], np.float64)
        self._traj = [[] for _ in range(self.number_of_replicas)]
        if isinstance(self._mcmc_sampler, MCMCSampler):
            self._mcmc_sampler = [copy.deepcopy(self._mcmc_sampler) for _ in range(K)]
        elif len(self._mcmc_sampler) != K:
            raise RuntimeError(
                f"The number of MCMCMoves ({len(self._mcmc_sampler)}) and ThermodynamicStates ({K}) must be the same."
            )
        self._iteration = 0

    def _compute_replica_reduced_potential(self, replica_id: int) -> np.ndarray:
        """chiron/multistate.py:178-201 (with the replica's neighbor list, cf. the bug noted in SURVEY 3.3)."""
        return calculate_reduced_potential_at_states(
            self._sampler_states[replica_id], self._thermodynamic_states, self._nbr_lists[replica_id]
        )

    def _propagate_replica(self, replica_id: int):
        """chiron/multistate.py:414-445"""
        state_id = self._replica_thermodynamic_states[replica_id]
        (
            self._sampler_states[replica_id],
            self._thermodynamic_states[state_id],
            self._nbr_lists[replica_id],
        ) = self._mcmc_sampler[state_id].run(
            self._sampler_states[replica_id], self._thermodynamic_states[state_id], self.n_mcmc_iterations,
            self._nbr_lists[replica_id],
        )
        self._traj[replica_id].append(self._sampler_states[replica_id].positions)

    def _propagate_replicas_batched(self) -> bool:
        """All replicas through ``MCMCSampler.run_batch``; False when that does not apply."""
        from .integrators import batch_applies
        from .mcmc import MCMCSampler

        K = self.number_of_replicas
        if not self._batch or K < 2:
            return False
        state_ids = [int(self._replica_thermodynamic_states[r]) for r in range(K)]
        samplers = [self._mcmc_sampler[s] for s in state_ids]
        if not MCMCSampler.batchable(samplers):
            return False
        thermo = [self._thermodynamic_states[s] for s in state_ids]
        integrators = [s.move.move_schedule[0][1].integrator for s in samplers]
        steps = samplers[0].move.move_schedule[0][1].number_of_moves
        if not batch_applies(integrators, self._sampler_states, thermo, self._nbr_lists, steps):
            return False
        out = MCMCSampler.run_batch(samplers, self._sampler_states, thermo, self.n_mcmc_iterations, self._nbr_lists)
        for r, (ss, ts, nl) in enumerate(out):  # chiron/multistate.py:414-445
            self._sampler_states[r], self._thermodynamic_states[state_ids[r]], self._nbr_lists[r] = ss, ts, nl
            self._traj[r].append(ss.positions)
        return True

    def _propagate_replicas(self) -> None:
        if self._propagate_replicas_batched():
            return
        k = min(self._concurrent, self.number_of_replicas)
        on_device = k > 1 and self._sampler_states[0].positions.is_cuda
        if not on_device:
            for replica_id in range(self.number_of_replicas):  # chiron/multistate.py:497-510
                self._propagate_replica(replica_id)
            return
        import threading

        streams = getattr(self, "_streams", None)
        if streams is None or len(streams) != k:
            streams = self._streams = [torch.cuda.Stream() for _ in range(k)]
        here = torch.cuda.current_stream()
        errors = []

        def work(lane: int):
            # thread `lane` takes replicas lane, lane + k, ... in turn, all on its own stream (torch's current stream
            # is per thread; the launch loop in C and every kernel launch release the GIL)
            try:
                with torch.cuda.stream(streams[lane]):
                    streams[lane].wait_stream(here)
                    for replica_id in range(lane, self.number_of_replicas, k):
                        self._propagate_replica(replica_id)
            except BaseException as exc:  # surfaced below, in the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(lane,)) for lane in range(k)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in streams:
            here.wait_stream(s)  # what follows on the caller's stream sees every replica's results
        if errors:
            raise errors[0]

    def _perform_swap_proposals(self):
        """Reference: placeholder returning the indices unchanged (chiron/multistate.py:447-460)."""
        if self._swap is None:
            return self._replica_thermodynamic_states
        new, n_prop, n_acc = neighbor_swap_decisions(
            self._energy_thermodynamic_states, self._replica_thermodynamic_states, self._iteration, self._key
        )
        self._n_proposed_matrix += n_prop
        self._n_accepted_matrix += n_acc
        return new

    def _mix_replicas(self) -> np.ndarray:
        """chiron/multistate.py:462-495"""
        self._n_accepted_matrix[:, :] = 0
        self._n_proposed_matrix[:, :] = 0
        old = np.array(self._replica_thermodynamic_states)
        new = np.array(self._perform_swap_proposals())
        if self._swap is not None:
            for r in np.nonzero(new != old)[0]:
                t_old = self._thermodynamic_states[old[r]].temperature.value_in_unit(unit.kelvin)
                t_new = self._thermodynamic_states[new[r]].temperature.value_in_unit(unit.kelvin)
                st = self._sampler_states[r]
                if st.velocities is not None and t_old != t_new:
                    from . import ops

                    st.velocities = ops.scale(st.velocities, math.sqrt(t_new / t_old))
            self._replica_thermodynamic_states = new
        return new

    def _compute_energies(self) -> None:
        """chiron/multistate.py:512-531"""
        self._energy_thermodynamic_states = np.zeros((self.number_of_replicas, self.number_of_thermodynamic_states))
        for replica_id in range(self.number_of_replicas):
            self._energy_thermodynamic_states[replica_id, :] = self._compute_replica_reduced_potential(replica_id)

    def _is_completed(self, iteration_limit: Optional[int] = None) -> bool:
        return iteration_limit is not None and self._iteration >= iteration_limit

    def _report_iteration(self):
        if self._reporter is not None:
            self._reporter.report({"u_kn": self._energy_thermodynamic_states.T.copy(),
                                   "state_index": np.array(self._replica_thermodynamic_states)})

    def run(self, n_iterations: int = 10, n_mcmc_iterations: Optional[int] = None) -> None:
        """chiron/multistate.py:563-599.  The reference passes the *total* n_iterations as the
        per-call MCMC iteration count (:441-443); ``n_mcmc_iterations`` overrides that."""
        self.number_of_iterations = n_iterations
        self.n_mcmc_iterations = n_iterations if n_mcmc_iterations is None else n_mcmc_iterations
        if self._iteration == 0:
            self._compute_energies()
            self._report_iteration()
        while not self._is_completed(n_iterations):
            self._iteration += 1
            self._mix_replicas()
            self._propagate_replicas()
            self._compute_energies()
            self._report_iteration()
        if self._reporter is not None:
            self._reporter.flush_buffer()
