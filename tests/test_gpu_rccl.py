"""The RCCL path on the one GPU a box has (chiron/multistate.py:497-531): a one-rank ``nccl`` process
group executes ``ReplicaExchange``'s device-tensor all-gather, the barrier and the all-reduce of bench.py's multi-rank
path -- the same calls the eight-rank job makes over xGMI, on the same library (librccl), with energies computed by the
HIP kernels.  Runs in a fresh child process (a process group is a per-process affair, and the launcher environment must
be set before anything touches the GPU there)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_one_rank_rccl_group_drives_the_exchange_of_eight_states():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank_child.py"), "8", "4096", str(_free_port())],
                         env=_child_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1 and d["k_states"] == 8
    assert d["rccl_mapped"] and d["libchx_mapped"]
    assert d["u_kn_exact"], "the gathered matrix is not beta_k * U_r of the energies that went in"
    assert d["row_rel_err"] < 1e-6, d["row_rel_err"]  # (fp32 energy through two routes: the same kernel, the same bits)
    assert d["all_reduce"] == 3.5
    for m in d["state_maps"]:
        assert sorted(m) == list(range(8))
