"""Register / LDS / scratch use of the library's kernels, straight from the compiler.

    python scripts/kernel_resources.py [pattern] [file.hip ...]

Compiles the named translation units (default: chx_lj.hip) for the device only with the product flags plus
``-Rpass-analysis=kernel-resource-usage`` and prints one line per kernel whose demangled name contains ``pattern``
(default: k_lj_step).  Nothing is written into the tree.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chiron_amd import build  # noqa: E402


def main():
    pattern = sys.argv[1] if len(sys.argv) > 1 else "k_lj_step"
    files = sys.argv[2:] or ["chx_lj.hip"]
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    for name in files:
        src = name if os.path.exists(name) else os.path.join(build.CSRC, name)
        with tempfile.TemporaryDirectory() as tmp:
            cmd = ["hipcc"] + flags + ["--cuda-device-only", "-c", src, "-o", os.path.join(tmp, "x.co"),
                                       "-Rpass-analysis=kernel-resource-usage"]
            err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur, rows = None, {}
        for line in err.splitlines():
            m = re.search(r"remark: (.*)", line)
            if not m:
                continue
            t = m.group(1)
            f = re.match(r"Function Name: (\S+)", t)
            if f:
                cur = f.group(1)
                rows[cur] = {}
                continue
            kv = re.match(r"\s*([A-Za-z ]+?)\s*(?:\[.*\])?:\s*(\S+)", t)
            if kv and cur:
                rows[cur][kv.group(1).strip()] = kv.group(2)
        demangle = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.split("\n")
        for mangled, nice in zip(rows, demangle):
            if pattern in nice:
                r = rows[mangled]
                short = re.sub(r"\(.*", "", nice)
                print(f"{short:60s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} SGPR {r.get('SGPRs', '?'):>4} "
                      f"scratch {r.get('ScratchSize', '?'):>5} LDS {r.get('LDS Size', '?'):>6} occupancy {r.get('Occupancy', '?')}")


if __name__ == "__main__":
    main()
