"""Helper of test_gpu_parity.py::test_both_cell_build_kernels_write_the_same_bits: builds a few neighbor lists with
whatever kernel CHX_BUILD_KERNEL selects and dumps every array the build produces.  usage: <out.npz>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chiron_amd import neighbors, unit as u
from oracle import chiron_oracle as O

out = {}
# (particles, reduced density, skin, initial n_max_neighbors): dilute and dense cells, a tiny initial capacity
# (rows longer than n_max_neighbors: the growth loop and the cell-wise kernel's long-row path), a box of 3 cells per axis
for tag, (n, rho, skin, m0) in {"dilute": (4000, 0.1, 0.5, 180), "dense": (6000, 0.8, 0.3, 180), "grow": (3000, 0.6, 0.4, 12),
                                "three_cells": (700, 0.5, 0.1, 60),
                                # ~15 particles per cell: passes on either side of the cell-wise kernel's on-chip-row limit,
                                # and half rows of the lowest ids longer than an on-chip row
                                "mixed": (5000, 0.15, 0.5, 180),
                                # ~215 neighbours per particle with candidate lists in place: more than the filtered row kernel
                                # (k_nl_build_sub_fl) keeps per row on chip, every row goes back to the general routine
                                "crowded": (3000, 0.7, 0.4, 260),
                                # coordinates up to three box lengths outside the box: the filter of the filtered row kernel
                                # passes such pairs untested, and the predicate takes its general minimum-image path
                                "unwrapped": (3000, 0.3, 0.3, 180)}.items():
    x, box = O.lj_lattice(n, rho, 0.34, seed=17)
    x = (x + np.random.default_rng(3).normal(scale=0.03, size=x.shape)).astype(np.float32)
    if tag == "unwrapped":
        x = (x + np.random.default_rng(5).integers(-3, 4, size=x.shape) * np.diag(box)).astype(np.float32)
    if tag == "mixed":  # uniformly random instead of a jittered lattice: cell occupancies scatter like a Poisson variable
        x = (np.random.default_rng(4).uniform(size=x.shape) * np.diag(box)).astype(np.float32)
    nbr = neighbors.NeighborListCell(neighbors.OrthogonalPeriodicSpace(), cutoff=1.02 * u.nanometer, skin=skin * u.nanometer,
                                        n_max_neighbors=m0)
    xt = torch.tensor(x, device="cuda")
    nbr.build(xt, torch.tensor(box, device="cuda"))
    fl = nbr.force_list
    out[tag + "_nl"] = nbr.neighbor_list.cpu().numpy()
    out[tag + "_mask"] = nbr.neighbor_mask.cpu().numpy()
    out[tag + "_n"] = nbr.n_neighbors.cpu().numpy()
    out[tag + "_m"] = np.array([nbr.n_max_neighbors])
    if fl is not None:
        deg = fl.deg.cpu().numpy()
        table = fl.fl.cpu().numpy()
        out[tag + "_deg"] = deg
        out[tag + "_perm"] = fl.perm.cpu().numpy()
        out[tag + "_geo"] = np.array([fl.g, fl.chunks])
        # only the entries the kernels may read are defined: whole chunks up to the row's end (at least chunk 0)
        w = 4 * fl.g
        used = np.maximum((deg + w - 1) // w, 1)
        keep = np.arange(table.shape[0])[:, None] < used[None, :]
        out[tag + "_fl"] = np.where(keep[:, :, None], table, 0)
    # The integrator's rebuilds (half rows counted, not written: ops.nl_build_cell_fl(half_rows=False)), which below the
    # cell-wise kernel's sizes go through the sixteen-lanes-per-row kernel (k_nl_build_sub_fl, CHX_BUILD_SUB=0: the
    # per-particle one) -- with the force-list capacity the build settled on and with a single chunk (rows overflow: the
    # status words that report it)
    if fl is not None:
        from chiron_amd import ops
        cs = nbr._cs_md
        ncell, nc = ops.cell_grid(nbr._box_h, cs, n)
        for sfx, chunks in (("", fl.chunks), ("_tight", 1)):
            _, _, n_nbr, ids, table, deg, status = ops.nl_build_cell_fl(xt, nbr._box_h, cs, int(nbr.n_max_neighbors), ncell, nc,
                                                                      fl.g, chunks, {}, half_rows=False)
            torch.cuda.synchronize()
            deg, table = deg.cpu().numpy(), table.cpu().numpy()
            w = 4 * fl.g
            keep = np.arange(table.shape[0])[:, None] < np.maximum((deg + w - 1) // w, 1)[None, :]
            out[tag + "_run_fl" + sfx] = np.where(keep[:, :, None], table, 0)
            out[tag + "_run_deg" + sfx] = deg
            out[tag + "_run_n" + sfx] = n_nbr.cpu().numpy()
            out[tag + "_run_ids" + sfx] = ids.cpu().numpy()
            out[tag + "_run_status" + sfx] = status.cpu().numpy()[:7]
np.savez(sys.argv[1], **out)
