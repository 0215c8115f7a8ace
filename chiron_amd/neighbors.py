"""Simulation spaces and pair structures behind the reference's operator API.

Mirrors chiron/neighbors.py: ``Space`` / ``OrthogonalPeriodicSpace`` /
``OrthogonalNonPeriodicSpace`` (:15-175), ``PairsBase`` (:178-375),
``NeighborListNsqrd`` (:378-907) and ``PairListNsqrd`` (:910-1289) -- same
constructor arguments, attributes, return tuples and exceptions -- plus
``NeighborListCell``, an O(N) cell-list build that the reference lacks (its
comments at :529-530 anticipate one) and that returns the *bit-identical* list.
All arithmetic runs in the HIP kernels of ``csrc/chx_neighbors.hip``.
"""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import ops, unit
from ._device import as_tensor, is_array
from .states import SamplerState

log = logging.getLogger("chiron_amd")
_DEBUG_BUILD = os.environ.get("CHX_DEBUG_BUILD") == "1"


# ---------------------------------------------------------------------------
# spaces
# ---------------------------------------------------------------------------
class Space(ABC):
    """chiron/neighbors.py:15-36"""

    periodic = False

    @abstractmethod
    def displacement(self, xyz_1, xyz_2, box_vectors):
        pass

    @abstractmethod
    def wrap(self, xyz, box_vectors):
        pass


class OrthogonalPeriodicSpace(Space):
    """chiron/neighbors.py:39-112"""

    periodic = True

    def displacement(self, xyz_1, xyz_2, box_vectors) -> Tuple[torch.Tensor, torch.Tensor]:
        if box_vectors is None:
            raise ValueError("box_vectors must be provided for a periodic system")
        return ops.displacement(as_tensor(xyz_1), as_tensor(xyz_2), ops.make_box(box_vectors, True))

    def wrap(self, xyz, box_vectors) -> torch.Tensor:
        if box_vectors is None:
            raise ValueError("box_vectors must be provided for a periodic system")
        return ops.wrap(as_tensor(xyz), ops.make_box(box_vectors, True))


class OrthogonalNonPeriodicSpace(Space):
    """chiron/neighbors.py:115-175"""

    periodic = False

    def displacement(self, xyz_1, xyz_2, box_vectors=None) -> Tuple[torch.Tensor, torch.Tensor]:
        return ops.displacement(as_tensor(xyz_1), as_tensor(xyz_2), ops.make_box(None, False))

    def wrap(self, xyz, box_vectors=None) -> torch.Tensor:
        return as_tensor(xyz)


# ---------------------------------------------------------------------------
# pairs base
# ---------------------------------------------------------------------------
def _strip_positions(positions):
    if isinstance(positions, unit.Quantity):
        if not positions.unit.is_compatible(unit.nanometer):
            raise ValueError(f"Positions require distance units, not {positions.unit}")
        positions = positions.value_in_unit_system(unit.md_unit_system)
    return positions


def _strip_box(box_vectors):
    if isinstance(box_vectors, unit.Quantity):
        if not box_vectors.unit.is_compatible(unit.nanometer):
            raise ValueError(f"Box vectors require distance unit, not {box_vectors.unit}")
        box_vectors = box_vectors.value_in_unit_system(unit.md_unit_system)
    if is_array(box_vectors):
        if tuple(box_vectors.shape) != (3, 3):
            raise ValueError(f"box_vectors should be a 3x3 array, shape provided: {tuple(box_vectors.shape)}")
    return box_vectors


class PairsBase(ABC):
    """chiron/neighbors.py:178-375"""

    def __init__(self, space: Space, cutoff: Optional[unit.Quantity] = unit.Quantity(1.2, unit.nanometer)):
        if not isinstance(space, Space):
            raise TypeError(f"space must be of type Space, found {type(space)}")
        if not cutoff.unit.is_compatible(unit.angstrom):
            raise ValueError(
                f"cutoff must be a unit.Quantity with units of distance, cutoff.unit = {cutoff.unit}"
            )
        self.cutoff = cutoff
        self.space = space

    @abstractmethod
    def build(self, positions, box_vectors):
        pass

    def _validate_build_inputs(self, positions, box_vectors):
        # chiron/neighbors.py:262-308
        positions = _strip_positions(positions)
        if is_array(positions):
            if positions.shape[1] != 3:
                raise ValueError(f"positions should be a Nx3 array, shape provided: {tuple(positions.shape)}")
        self.ref_positions = as_tensor(positions)
        box_vectors = _strip_box(box_vectors)
        self.box_vectors = None if box_vectors is None else as_tensor(box_vectors)

    def build_from_state(self, sampler_state: SamplerState):
        # chiron/neighbors.py:310-331
        if not isinstance(sampler_state, SamplerState):
            raise TypeError(f"Expected SamplerState, got {type(sampler_state)} instead")
        self._box_token = None
        self.build(sampler_state.positions, sampler_state.box_vectors)
        # lets the integrator recognise "same box as at the last build" without reading the device
        self._box_token = getattr(sampler_state, "_box_token", None)

    @abstractmethod
    def calculate(self, positions):
        pass

    @abstractmethod
    def check(self, positions) -> bool:
        pass

    def _box(self) -> ops.Box:
        return self._box_h


# ---------------------------------------------------------------------------
# Verlet neighbor list
# ---------------------------------------------------------------------------
class NeighborListNsqrd(PairsBase):
    """Half (unique-pair) Verlet list with skin, padded to ``n_max_neighbors``.

    Reference: chiron/neighbors.py:378-907.  The outputs (``neighbor_list`` (N,M),
    ``neighbor_mask`` (N,M) int32, ``n_neighbors`` (N,) int32) are bit-identical to the
    reference's ``vmap(_build_neighborlist)`` including padding and the growth loop.
    Neighbor ids are stored in an int32 tensor holding the uint32 payload.

    ``algorithm`` (not in the reference) picks the build kernel: ``"nsq"`` is the O(N^2)
    wave-per-row kernel that follows the reference's own loop structure, ``"cell"`` the O(N)
    cell-list kernel; both produce the same bits, so the default ``"auto"`` takes the cell
    kernel whenever the periodic box holds at least 3 cells per axis and the O(N^2) kernel
    otherwise (non-periodic spaces, small boxes).
    """

    _algorithm = "auto"

    def __init__(
        self,
        space: Space,
        cutoff: unit.Quantity = unit.Quantity(1.2, unit.nanometer),
        skin: unit.Quantity = unit.Quantity(0.4, unit.nanometer),
        n_max_neighbors: int = 200,
        *,
        algorithm: Optional[str] = None,
    ):
        algorithm = self._algorithm if algorithm is None else algorithm
        if algorithm not in ("auto", "nsq", "cell"):
            raise ValueError(f"algorithm must be 'auto', 'nsq' or 'cell', got {algorithm!r}")
        self._use_cell_list = algorithm != "nsq"
        if not isinstance(space, Space):
            raise TypeError(f"space must be of type Space, found {type(space)}")
        if not skin.unit.is_compatible(unit.angstrom):
            raise ValueError(
                f"cutoff must be a unit.Quantity with units of distance, skin.unit = {skin.unit}"
            )
        self.cutoff = cutoff
        self.skin = skin
        self.n_max_neighbors = n_max_neighbors
        self.space = space
        self.is_built = False
        self._force_list = None
        self._nl = self._nl_mask = self._first = None
        self.n_builds = 0

    @property
    def cutoff(self) -> unit.Quantity:
        return self._cutoff

    @cutoff.setter
    def cutoff(self, cutoff: unit.Quantity) -> None:
        if not cutoff.unit.is_compatible(unit.nanometer):
            raise ValueError(
                f"cutoff must be a unit.Quantity with units of distance, cutoff.unit = {cutoff.unit}"
            )
        self._cutoff = cutoff
        self._cs_md = None
        self.is_built = False  # chiron/neighbors.py:485

    @property
    def skin(self) -> unit.Quantity:
        return self._skin

    @skin.setter
    def skin(self, skin: unit.Quantity) -> None:
        if not skin.unit.is_compatible(unit.nanometer):
            raise ValueError(f"skin must be a unit.Quantity with units of distance, skin.unit = {skin.unit}")
        self._skin = skin
        self._cs_md = None
        self.is_built = False  # chiron/neighbors.py:502

    # -- build -------------------------------------------------------------
    def set_replicas_sharing_card(self, k: int) -> None:
        """Tell the list that its system is one of ``k`` replicas propagated side by side on one card
        (``MultiStateSampler``: integrators.run_batch steps them in one launch per MD step).  A speed hint only: the force
        list's lanes per particle are chosen for k x N particles in flight instead of N (eight N = 32 768 replicas in one
        grid are a throughput problem like N = 262 144: two lanes per particle instead of four, +3 % aggregate rate).
        The pairs, the half rows and every result stay what they are; forces change in summation order only.  Takes
        effect with the next build."""
        k = max(1, int(k))
        if k != getattr(self, "_replicas_sharing_card", 1):
            self._replicas_sharing_card = k
            self._fl_geo = None  # (the sticky geometry was chosen for another lane count)

    def _fl_geometry(self, n: int):
        """(lanes per particle, chunks) of the force list the one-pass cell build writes into."""
        geo = getattr(self, "_fl_geo", None)
        if geo is not None and geo[0] == n:
            # sticky: identical tensor sizes from rebuild to rebuild let the caching allocator hand the
            # same blocks back (a changed size costs a hipMalloc, i.e. milliseconds of idle GPU)
            return geo[1], geo[2]
        # first build: expected number of neighbors inside cutoff+skin; capacity with head-room
        b = self._box_h
        rho = n / (b.lx * b.ly * b.lz)
        expected = rho * 4.18879 * self._cutoff_and_skin ** 3
        g = ops._choose_g(int(1.2 * expected) + 4, n * getattr(self, "_replicas_sharing_card", 1))
        chunks = ops.fl_chunks_for(int(1.6 * expected) + 24, g)
        self._fl_geo = (n, g, chunks)
        return g, chunks

    def _build_pass(self, m: int) -> dict:
        """One fixed-width pass.  Returns the arrays plus ``max_n`` (largest row count) and ``n_eq``
        (rows holding exactly m entries, the reference's growth trigger) from ONE host read-back."""
        cs = self._cutoff_and_skin
        n = self.ref_positions.shape[0]
        if self._use_cell_list and self._box_h.periodic:
            ncell, nc = ops.cell_grid(self._box_h, cs, n)
            if ncell > 0:
                g, chunks = self._fl_geometry(n)
                for _ in range(5):
                    if getattr(self, "_scratch", None) is None:
                        self._scratch = {}
                    nl, first, n_nbr, sorted_ids, fl, deg, status = ops.nl_build_cell_fl(
                        self.ref_positions, self._box_h, cs, m, ncell, nc, g, chunks, self._scratch)
                    ops.steps_join()  # (inside a run() the build's launches ride on the step queue: chx_steps_join)
                    st = status.cpu().tolist()  # the build's only host sync (the reference syncs at :709)
                    if _DEBUG_BUILD:
                        print(f"[chx build] n={n} m={m} g={g} chunks={chunks} status={st}", flush=True)
                    if st[5]:  # a cell held more particles than its id slots: the same build without the slots
                        self._scratch["no_slots"] = True
                        continue
                    if st[0]:
                        log.debug("cell build overflowed its on-chip candidate buffer; using the O(N^2) kernel")
                        break
                    if st[4] > 0:  # some force-list row did not fit: grow the (sticky) capacity
                        g = ops._choose_g(st[3], n * getattr(self, "_replicas_sharing_card", 1))
                        chunks = max(ops.fl_chunks_for(st[3] + 8, g), -(-3 * chunks // 2))
                        self._fl_geo = (n, g, chunks)
                        continue
                    self._last_maxdeg = st[3]
                    self._last_max_n = st[1]
                    force_list = ops.ForceList.prebuilt(fl, deg, sorted_ids, g, chunks, st[3])
                    return dict(nl=nl, first=first, mask=None, n_nbr=n_nbr, perm=sorted_ids, fl=force_list,
                                max_n=st[1], n_eq=st[2])
        nl, mask, n_nbr = ops.nl_build_nsq(self.ref_positions, self._box_h, cs, m)
        summary = torch.stack([n_nbr.max(), (n_nbr == m).sum().to(n_nbr.dtype)]).cpu().tolist() if n > 0 else [0, 0]
        return dict(nl=nl, first=None, mask=mask, n_nbr=n_nbr, perm=None, fl=None, max_n=summary[0], n_eq=summary[1])

    def build(self, positions, box_vectors):
        """chiron/neighbors.py:628-729"""
        positions = _strip_positions(positions)
        self.__dict__.pop("_deferred", None)  # (a rebuild still pending from an earlier run() is superseded)
        self.__dict__.pop("_half_rows_pending", None)  # (so are half rows an integrator rebuild left for later)
        private = getattr(self, "_adopt_positions", False)  # the integrator's own rebuild inside run()
        if (private and box_vectors is not None and box_vectors is getattr(self, "box_vectors", None)
                and getattr(self, "_box_h", None) is not None):
            pass  # the very tensor this list already holds, which nobody else can reach: host copy is current
        else:
            # a build requested from outside says nothing about the state it will be used with: forget the
            # "same box as the sampler state" token (build_from_state sets it again after the build), and keep a
            # private copy of the box -- unlike a JAX array a torch tensor can be mutated in place by its owner
            self._box_token = None
            box_vectors = _strip_box(box_vectors)
            if box_vectors is None:
                self.box_vectors = None
            else:
                bt = as_tensor(box_vectors)
                self.box_vectors = bt.clone() if bt is box_vectors else bt
            if self.space.periodic and self.box_vectors is None:
                # the reference raises from Space.displacement (chiron/neighbors.py:71-72)
                raise ValueError("box_vectors must be provided for a periodic system")
            self._box_h = ops.make_box(self.box_vectors, self.space.periodic)  # reads the box on the host
        pos = as_tensor(positions)
        # ``_adopt_positions``: the integrator hands over a tensor nobody else holds (no copy needed)
        self.ref_positions = pos if getattr(self, "_adopt_positions", False) and pos is positions else pos.clone()
        if self._cs_md is None:  # unit arithmetic is slow Python: once per (cutoff, skin)
            self._cs_md = float((self.cutoff + self.skin).value_in_unit_system(unit.md_unit_system))
        self._cutoff_and_skin = self._cs_md
        n = self.ref_positions.shape[0]
        ids = getattr(self, "particle_ids", None)
        if ids is None or ids.shape[0] != n or ids.device != self.ref_positions.device:
            self.particle_ids = torch.arange(n, dtype=torch.int32, device=self.ref_positions.device)

        res = self._build_pass(int(self.n_max_neighbors))
        # growth loop, chiron/neighbors.py:709-727 (equality test, as in the reference)
        while res["n_eq"] > 0:
            new_max = int(res["max_n"]) + 10
            log.debug(f"Increasing n_max_neighbors from {self.n_max_neighbors} to at {new_max}")
            self.n_max_neighbors = new_max
            res = self._build_pass(int(self.n_max_neighbors))
        truncated = res["max_n"] > self.n_max_neighbors
        if truncated:
            log.warning(
                "NeighborListNsqrd: some rows hold more than n_max_neighbors entries and none holds exactly "
                "that many; like the reference (chiron/neighbors.py:709) the list is truncated silently."
            )
        self._nl, self._nl_mask, self._first = res["nl"], res["mask"], res["first"]
        self.n_neighbors = res["n_nbr"]
        self._perm = res["perm"]
        # a truncated public list interacts through its listed pairs only: derive the table from it
        self._force_list = None if truncated else res["fl"]
        self.is_built = True
        self.n_builds += 1

    # -- unsynchronised rebuild (integrator only) ---------------------------------------------
    def _prepare_next_build(self, like: torch.Tensor) -> None:
        """The arrays a next speculative build allocates BEFORE its first launch (the copy of the positions, the internal
        order), made now -- by an integrator whose launches are still running, with nothing else to do -- so that a short
        run() that follows gets its first launch out 2.5 us earlier.  Plain empties; unused, they are simply dropped."""
        if self.__dict__.get("_next_copy") is None:
            self._next_copy = torch.empty_like(like)
        sc = self.__dict__.get("_scratch")
        if sc is not None and sc.get("next_ids") is None:
            sc["next_ids"] = torch.empty((like.shape[0],), dtype=torch.int32, device=like.device)

    def _build_speculative(self, positions: torch.Tensor, publish: bool = True, pack=None, ctrl=None, ctrl_keep: int = 0,
                           copy_positions: bool = False, half_rows: bool = True) -> bool:
        """Rebuild from ``positions`` WITHOUT waiting for the build's status block.

        The reference synchronises inside every build (chiron/neighbors.py:709) to decide whether
        ``n_max_neighbors`` has to grow.  Between two rebuilds of one trajectory it almost never does,
        so the integrator enqueues the build -- whose last launch stores the status block in pinned host memory --
        and carries on with the arrays it got; ``_confirm_build`` reads the block later (by then it has
        long arrived) and, if a capacity did overflow, redoes the build through ``build()`` -- the
        caller then discards whatever it ran on the speculative list.  The overflowing kernel bounds
        its writes, so a speculative list is always safe to walk.  Returns False (nothing enqueued)
        when this fast path does not apply.  ``publish=True`` records an event behind the build that
        ``_confirm_build`` waits for; with ``publish=False`` the caller has its own way of knowing that the build has run
        (the integrator: a later launch has begun) and confirms with ``_confirm_build(wait=False)``.
        Riders of the build's launches (include/chx.h: chx_build_riders_t): ``pack`` = (v3, mass, ref3 or None, x4_out,
        v_out, mass_out, ref4_out) -- the Langevin state packed into the new internal order (the caller vouches for them:
        contiguous fp32 device tensors of this list's particle count -- they are not checked again on the way down); ``ctrl`` / ``ctrl_keep`` --
        the integrator's control block, reset; ``copy_positions`` -- ``positions`` is somebody else's tensor: the list
        keeps a copy made by the binning launch instead of the tensor itself.  ``half_rows=False``: the reference-format
        half rows (``neighbor_list``) are counted, not written -- the integrator's step kernels read the force list only;
        the rows are built from the same positions the first time somebody reads them (``__getattr__``).
        """
        if not (self.is_built and self._use_cell_list and getattr(self, "_box_h", None) is not None
                and self._box_h.periodic and getattr(self, "_fl_geo", None) is not None and self._cs_md is not None):
            return False
        n = positions.shape[0]
        if self._fl_geo[0] != n or positions.dtype != torch.float32 or not positions.is_cuda or not positions.is_contiguous():
            return False
        cs = self._cs_md
        gc = self.__dict__.get("_grid_cache")  # (the grid is a function of the box object, cutoff + skin and n)
        if gc is None or gc[0] is not self._box_h or gc[1] != cs or gc[2] != n:
            gc = self._grid_cache = (self._box_h, cs, n) + tuple(ops.cell_grid(self._box_h, cs, n))
        ncell, nc = gc[3], gc[4]
        if ncell <= 0:
            return False
        m = int(self.n_max_neighbors)
        g, chunks = self._fl_geometry(n)
        d = self.__dict__
        if d.get("_scratch") is None:
            self._scratch = {}
        d.pop("_deferred", None)  # (a rebuild still pending from an earlier run() is superseded)
        host = d.get("_status_host")
        if host is None:
            host = self._status_host = torch.empty((8,), dtype=torch.int32, pin_memory=True)
        if d.get("_status_np") is None:
            self._status_np = host.numpy()  # (a view: reading a word costs a tenth of a tensor op)
        # (word 7 of the block the build's last launch publishes is always 0: until then it says "not yet" --
        # _wait_for_status; the previous speculative build has been confirmed by now, nothing else writes there)
        self._status_np[7] = 0x7FFFFFFF
        riders = {"publish": host}
        if copy_positions:
            ready = self.__dict__.pop("_next_copy", None)  # (_prepare_next_build)
            riders["copy"] = ready if ready is not None and ready.shape == positions.shape and ready.device == positions.device \
                else torch.empty_like(positions)
        if pack is not None:
            riders["pack"] = pack
        if ctrl is not None:
            riders["ctrl"], riders["ctrl_keep"] = ctrl, ctrl_keep
        self.ref_positions = riders["copy"] if copy_positions else positions  # (else: a fresh tensor nobody else holds)
        self._cutoff_and_skin = cs
        nl, first, n_nbr, sorted_ids, fl, deg, status = ops.nl_build_cell_fl(
            positions, self._box_h, cs, m, ncell, nc, g, chunks, self._scratch, riders=riders, half_rows=half_rows,
            checked=True)  # (positions: above; the copy: made here; the pack rider's arrays: the integrator's, see `pack`)
        ev = getattr(self, "_spec_event", None)
        if ev is None:
            ev = self._spec_event = torch.cuda.Event()  # (one event, recorded again by every speculative build)
        if publish:
            ev.record()
        self._spec_recorded = bool(publish)
        self._spec = (ev, m, g, chunks)
        if half_rows:
            self._nl, self._nl_mask, self._first = nl, None, first
        else:  # (read through __getattr__: built on first use)
            for name in self._HALF_ROW_ATTRS:
                self.__dict__.pop(name, None)
            self.__dict__["_half_rows_pending"] = True
        self.n_neighbors = n_nbr
        self._perm = sorted_ids
        self._force_list = ops.ForceList.prebuilt(fl, deg, sorted_ids, g, chunks, getattr(self, "_last_maxdeg", 0))
        self.n_builds += 1
        return True

    def _wait_for_status(self, timeout_s: float = 0.25) -> bool:
        """Host wait until the last ``_build_speculative``'s status block has ARRIVED (its word 7 turns 0), at most
        ``timeout_s``: a wait for the build alone, not for everything queued behind it -- what a caller with launches of
        its own still in flight wants before ``_confirm_build(wait=False)``.  False: not seen in time (wait for the
        queue instead)."""
        import time

        st = self.__dict__.get("_status_np")
        if st is None:
            return False
        t_end = time.perf_counter() + timeout_s
        while int(st[7]) != 0:
            if time.perf_counter() > t_end:
                return False
        return True

    def _build_status_good(self) -> bool:
        """Does the status block of the last ``_build_speculative`` -- which the caller knows to have arrived -- accept the
        build?  Looks only: ``_confirm_build`` is what acts on it."""
        spec = getattr(self, "_spec", None)
        if spec is None:
            return True
        st = self._status_host.tolist()
        return st[0] == 0 and st[2] == 0 and st[4] == 0 and st[5] == 0 and st[1] <= spec[1]

    def _confirm_build(self, wait: bool = True) -> bool:
        """True if the last ``_build_speculative`` produced the list ``build()`` would have; otherwise
        the list has been rebuilt synchronously from the same positions and False is returned."""
        spec = getattr(self, "_spec", None)
        if spec is None:
            return True
        self._spec = None
        ev, m, g, chunks = spec
        if wait:
            ops.steps_join()  # (a build enqueued inside a run() rides on the step queue, which the stream does not see)
            if getattr(self, "_spec_recorded", True):
                ev.synchronize()
            else:
                torch.cuda.current_stream().synchronize()
        st = self._status_host.tolist()
        # [0] on-chip candidate overflow, [1] max n_neighbors, [2] rows with exactly m entries (the
        # reference's growth trigger), [3] max force-list degree, [4] force-list rows that did not fit
        if st[5] and self._scratch is not None:  # (a cell overflowed its id slots: the build below goes without them)
            self._scratch["no_slots"] = True
        if st[0] == 0 and st[2] == 0 and st[4] == 0 and st[5] == 0 and st[1] <= m:
            self._last_maxdeg = st[3]
            self._last_max_n = st[1]
            self._force_list.maxdeg = st[3]
            return True
        if _DEBUG_BUILD:
            print(f"[chx build] speculative rebuild rejected, status={st}", flush=True)
        self.n_builds -= 1
        self._adopt_positions = True
        try:
            self.build(self.ref_positions, self.box_vectors)
        finally:
            self._adopt_positions = False
        return False

    # -- list carry-over (integrator only) ------------------------------------------------------
    def _clone_builder(self):
        """A private list builder with this list's geometry and capacities and buffers of its own: what the integrator
        renews a carried-over force list with (``LangevinIntegrator``: list carry-over) without touching the caller's
        list.  None when the one-pass cell build does not apply."""
        if not (self.is_built and self._use_cell_list and getattr(self, "_box_h", None) is not None and self._box_h.periodic
                and getattr(self, "_fl_geo", None) is not None and self._cs_md is not None
                and (self.__dict__.get("_perm") is not None or "_deferred" in self.__dict__)):
            return None
        b = self.__class__.__new__(self.__class__)
        b._use_cell_list = True
        b._cutoff, b._skin, b._cs_md = self._cutoff, self._skin, self._cs_md
        b.n_max_neighbors = self.n_max_neighbors
        b.space = self.space
        b.box_vectors, b._box_h, b._box_token = self.box_vectors, self._box_h, None
        b._fl_geo = self._fl_geo
        b._last_maxdeg = getattr(self, "_last_maxdeg", 0)
        b._force_list = None
        b._nl = b._nl_mask = b._first = None
        b._scratch = None
        b.n_builds = 0
        b.is_built = True
        return b

    def _detach_for_carry(self):
        """The list as it stands -- rows, force list, the cell-ordered positions it was built from -- handed over to a
        private list object the integrator keeps (``LangevinIntegrator``: list carry-over between ``run()`` calls).
        The arrays of a build are never written again, so both objects may go on reading them; the work arrays go
        with the private object (the next build of this list allocates its own: it must not overwrite the positions
        the carried force list is validated against).  None when the last build was not a confirmed one-pass cell
        build."""
        sc = self.__dict__.get("_scratch")
        fl = self.__dict__.get("_force_list")
        if (sc is None or sc.get("arrays") is None or fl is None or self.__dict__.get("_perm") is None
                or getattr(self, "_spec", None) is not None or fl.perm is not self._perm):
            return None
        b = self._clone_builder()
        if b is None or sc["arrays"][4].shape[0] != self._perm.shape[0]:
            return None
        for name in self._DEFERRED_ATTRS:
            if name in self.__dict__:
                b.__dict__[name] = self.__dict__[name]
        if self.__dict__.get("_half_rows_pending"):
            b.__dict__["_half_rows_pending"] = True
        b._scratch, self._scratch = sc, None
        b._last_max_n = getattr(self, "_last_max_n", None)
        return b

    _DEFERRED_ATTRS = ("ref_positions", "n_neighbors", "_nl", "_nl_mask", "_first", "_perm", "_force_list")
    _HALF_ROW_ATTRS = ("_nl", "_nl_mask", "_first")  # what a build with half_rows=False leaves for the first reader

    def _defer_build(self, ref_positions: torch.Tensor, n_events: int) -> None:
        """The reference rebuilt this list ``n_events`` times inside ``run()`` and the last rebuild was from
        ``ref_positions`` (chiron/integrators.py:241-242) -- but nobody has looked at it since: the arrays are built
        from those positions the first time one of them is read (``__getattr__``).  ``is_built`` stays True, as in the
        reference; ``n_builds`` counts the reference's rebuilds."""
        for name in self._DEFERRED_ATTRS:
            self.__dict__.pop(name, None)
        self.__dict__.pop("_half_rows_pending", None)
        self.__dict__["_deferred"] = ref_positions
        self.n_builds += int(n_events)

    def __getattr__(self, name):
        # (reached only when normal lookup fails: costs nothing on the hot path)
        if name in NeighborListNsqrd._HALF_ROW_ATTRS and self.__dict__.get("_half_rows_pending"):
            # The integrator's last rebuild wrote the force list and the row counts only.  The half rows are a pure
            # function of (reference positions, box, cutoff + skin, n_max_neighbors): the synchronous build from the same
            # positions writes what that rebuild would have (and redoes the force list, bit for bit the same).
            self._confirm_build()
            if self.__dict__.get("_half_rows_pending"):  # (a rejected speculative build has already been redone in full)
                builds = self.n_builds
                self._adopt_positions = True  # the list's own tensor
                try:
                    self.build(self.ref_positions, self.box_vectors)
                finally:
                    self._adopt_positions = False
                self.n_builds = builds
            return self.__dict__[name]
        if name in NeighborListNsqrd._DEFERRED_ATTRS and "_deferred" in self.__dict__:
            pos = self.__dict__["_deferred"]
            builds = self.n_builds
            self._adopt_positions = True  # a tensor nobody else holds
            try:
                # (build() drops the `_deferred` entry itself; if it raises -- out of memory, a box that no longer
                # fits -- the pending positions are put back, so a later read tries again instead of finding a list
                # that claims to be built and holds no arrays)
                self.build(pos, self.box_vectors)
            except BaseException:
                self.__dict__["_deferred"] = pos
                self.n_builds = builds
                raise
            finally:
                self._adopt_positions = False
            self.n_builds = builds  # already counted by _defer_build
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def _finish_padding(self):
        """Padding (:603-620) and mask (:623) are produced on first use: the MD loop never reads them."""
        if self._nl_mask is None:
            self._nl_mask = ops.nl_finish_padding(self._nl, self._first, self.n_neighbors)

    @property
    def neighbor_list(self) -> torch.Tensor:
        """(N, n_max_neighbors) neighbor ids (uint32 payload in an int32 tensor), padded as the reference pads."""
        self._finish_padding()
        return self._nl

    @neighbor_list.setter
    def neighbor_list(self, value):
        self._nl = value

    @property
    def neighbor_mask(self) -> torch.Tensor:
        self._finish_padding()
        return self._nl_mask

    @neighbor_mask.setter
    def neighbor_mask(self, value):
        self._nl_mask = value

    @property
    def force_list(self) -> ops.ForceList:
        """Internal full table for the force kernel (from the one-pass build, else derived from the half list)."""
        if self._force_list is None:
            self._force_list = ops.ForceList(self._nl, self.n_neighbors, self._perm)
        return self._force_list

    # -- calculate / check -------------------------------------------------
    def calculate(self, positions):
        """chiron/neighbors.py:789-826 -> (n_neighbors, neighbor_list, padding_mask, dist, r_ij)"""
        positions = as_tensor(positions)
        cutoff = float(self.cutoff.value_in_unit_system(unit.md_unit_system))
        n_pairs, mask, dist, rij = ops.nl_calculate(positions, self.neighbor_list, self.neighbor_mask, self._box_h, cutoff)
        return n_pairs, self.neighbor_list, mask, dist, rij

    def check_device(self, positions) -> Optional[torch.Tensor]:
        """Device flag (int32[1]) of the displacement test, or None when N changed."""
        positions = as_tensor(positions)
        if self.ref_positions.shape[0] != positions.shape[0]:
            return None
        skin = float(self.skin.value_in_unit_system(unit.md_unit_system))
        return ops.nl_check(positions, self.ref_positions, self._box_h, skin)

    def check(self, positions) -> bool:
        """chiron/neighbors.py:872-907"""
        flag = self.check_device(positions)
        if flag is None:
            return True
        return bool(flag.item())

    def __deepcopy__(self, memo):
        import copy

        self._confirm_build()
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_force_list":
                new._force_list = None  # rebuilt lazily; it is a pure function of the list
            elif k == "_scratch":
                new._scratch = None  # kernel work arrays are not part of the list
            elif k in ("_spec", "_status_host", "_status_np", "_grid_cache", "_next_copy"):
                setattr(new, k, None)
            elif k in ("_spec_event",):
                continue  # plumbing of the speculative rebuild: the copy starts without any
            elif isinstance(v, torch.Tensor):
                setattr(new, k, v.clone())
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new


class NeighborListCell(NeighborListNsqrd):
    """Same contract as ``NeighborListNsqrd`` with an O(N) cell-list build.

    The reference has no cell list; this one bins the reference positions into
    cells of edge >= cutoff+skin, scans the 27 surrounding cells with the same
    fp32 predicate, and emits each row in ascending-id order with the same
    padding, so ``neighbor_list`` / ``neighbor_mask`` / ``n_neighbors`` are
    bit-identical to ``NeighborListNsqrd`` on the same input.  Falls back to the
    O(N^2) kernel when the box holds fewer than 3 cells per axis.  (Kept as an explicit name;
    ``NeighborListNsqrd`` with its default ``algorithm="auto"`` takes the same kernels.)
    """

    _algorithm = "cell"


# ---------------------------------------------------------------------------
# all-pairs list
# ---------------------------------------------------------------------------
class PairListNsqrd(PairsBase):
    """chiron/neighbors.py:910-1289"""

    def __init__(self, space: Space, cutoff: Optional[unit.Quantity] = None):
        if not isinstance(space, Space):
            raise TypeError(f"space must be of type Space, found {type(space)}")
        self.cutoff = cutoff
        self.space = space
        self.is_built = False

    @property
    def cutoff(self):
        return self._cutoff

    @cutoff.setter
    def cutoff(self, cutoff):
        if cutoff is not None:
            if not cutoff.unit.is_compatible(unit.angstrom):
                raise ValueError(
                    f"cutoff must be a unit.Quantity with units of distance, cutoff.unit = {cutoff.unit}"
                )
        self._cutoff = cutoff  # no rebuild needed (chiron/neighbors.py:1006-1007)

    def build(self, positions, box_vectors):
        """chiron/neighbors.py:1069-1104"""
        self._validate_build_inputs(positions, box_vectors)
        if self.space.periodic and self.box_vectors is None:
            raise ValueError("box_vectors must be provided for a periodic system")
        self._box_h = ops.make_box(self.box_vectors, self.space.periodic)
        self.n_particles = self.ref_positions.shape[0]
        dev = self.ref_positions.device
        self.particle_ids = torch.arange(self.n_particles, dtype=torch.int32, device=dev)
        self.all_pairs, self.reduction_mask = ops.pl_build(self.n_particles, dev)
        self.is_built = True

    def _cutoff_md(self) -> Optional[float]:
        if self.cutoff is None:
            return None
        return float(self.cutoff.value_in_unit_system(unit.md_unit_system))

    def calculate(self, positions):
        """chiron/neighbors.py:1218-1270 -> (n_neighbors, all_pairs, padding_mask, dist, r_ij)"""
        positions = as_tensor(positions)
        if positions.shape[0] != self.n_particles:
            raise ValueError(
                f"Number of particles cannot changes without rebuilding. "
                f"Positions must have shape ({self.n_particles}, 3), found {tuple(positions.shape)}"
            )
        n_pairs, mask, dist, rij = ops.pl_calculate(positions, self._box_h, self._cutoff_md())
        return n_pairs, self.all_pairs, mask, dist, rij

    def check(self, positions) -> bool:
        """chiron/neighbors.py:1272-1289"""
        return positions.shape[0] != self.n_particles

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                setattr(new, k, v.clone())
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new
