"""Compile the C restatement of the reference algorithm (test infrastructure / cpu_baseline).

Output: oracle/_build/liboracle.so (git-ignored via *.so; travels to the GPU box with the snapshot).  The binary
carries the sha256 of oracle.c and the compiler flags; ``build()`` recompiles exactly when that differs from the
text beside it (file times are not consulted: a snapshot copy or a checkout may reset them), so a stale checker
cannot pass for a fresh one.
"""
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")
FLAGS = ["-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared"]
_MARK = b"ORC_SOURCE_HASH="


def source_hash() -> str:
    h = hashlib.sha256()
    with open(SRC, "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def binary_hash(path: str = None):
    try:
        with open(path or OUT, "rb") as fh:
            m = re.search(re.escape(_MARK) + rb"([0-9a-f]{64})", fh.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    want = source_hash()
    if force or binary_hash() != want:
        tmp = OUT + ".tmp"
        subprocess.run(["gcc"] + FLAGS + [f'-DORC_SOURCE_HASH="{want}"', SRC, "-o", tmp, "-lm"], check=True)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
