"""Child process of tests/test_gpu_rccl.py: a ONE-RANK RCCL group on the one MI355X of the box, carrying the very
collective the eight-rank job runs (chiron_amd.multistate.ReplicaExchange.gather_energies: all_gather_into_tensor of
fp64 device tensors) with energies that come from the HIP kernels.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    k_states = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[3] if len(sys.argv) > 3 else "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from chiron_amd import _lib, unit
    from chiron_amd.multistate import ReplicaExchange
    from chiron_amd.neighbors import NeighborListCell, OrthogonalPeriodicSpace
    from chiron_amd.potential import LJPotential
    from chiron_amd.rng import PRNGKey
    from chiron_amd.states import SamplerState, ThermodynamicState, calculate_reduced_potential_at_states
    from chiron_amd.testsystems import LennardJonesFluid

    _lib.load()
    sigma, eps = 0.34 * unit.nanometer, 0.238 * unit.kilocalories_per_mole
    fluid = LennardJonesFluid(n, reduced_density=0.1, sigma=sigma, epsilon=eps, seed=1234)
    pot = LJPotential(fluid.topology, sigma=sigma, epsilon=eps, cutoff=3.0 * sigma)
    ladder = [ThermodynamicState(pot, temperature=300.0 * (1.05 ** k) * unit.kelvin) for k in range(k_states)]
    rex = ReplicaExchange(ladder, PRNGKey(2024, 7))  # rank / world size from the process group
    assert rex.world_size == 1 and rex.replicas_per_rank == k_states
    # K replicas: the same fluid displaced by a replica-specific jitter, each with its own list
    rng = np.random.default_rng(5)
    x0 = fluid.positions.value_in_unit(unit.nanometer)
    states, lists, energies = [], [], []
    for r in range(k_states):
        x = (x0 + rng.normal(scale=0.01 * (1 + r), size=x0.shape)).astype(np.float32)
        st = SamplerState(unit.Quantity(x, unit.nanometer), PRNGKey(9, r), box_vectors=fluid.box_vectors)
        nb = NeighborListCell(OrthogonalPeriodicSpace(), cutoff=3.0 * sigma, skin=0.5 * unit.nanometer, n_max_neighbors=180)
        nb.build_from_state(st)
        states.append(st), lists.append(nb)
        energies.append(pot.compute_energy(st.positions, nb).reshape(1))
    e_dev = torch.cat(energies).to(torch.float64)
    assert e_dev.is_cuda
    out = []
    for it in range(4):
        moves = rex.exchange_many(e_dev)
        out.append([int(v) for v in rex.state_of_replica])
        assert len(moves) == k_states
    e_host = e_dev.cpu().numpy()
    ukn_ok = bool(np.array_equal(rex.u_kn, rex._reduced_row(e_host, None)))
    # row r of the matrix against the reference's own route (chiron/states.py:335-366) on replica r's configuration
    row_err = 0.0
    for r in (0, k_states - 1):
        ref_row = np.asarray(calculate_reduced_potential_at_states(states[r], ladder, lists[r]), dtype=np.float64)
        row_err = max(row_err, float(np.abs(ref_row - rex.u_kn[r]).max() / max(1.0, np.abs(ref_row).max())))
    # the other collectives bench.py's multi-rank path uses
    t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    with open("/proc/self/maps") as fh:
        maps = fh.read()
    print(json.dumps({
        "backend": dist.get_backend(), "world": dist.get_world_size(), "k_states": k_states, "u_kn_exact": ukn_ok,
        "row_rel_err": row_err, "all_reduce": float(t.item()), "state_maps": out,
        "rccl_mapped": ("librccl" in maps), "libchx_mapped": ("libchx.so" in maps),
        "swap_acceptance": float(rex.n_accepted.sum() / max(1, rex.n_proposed.sum())),
    }), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
