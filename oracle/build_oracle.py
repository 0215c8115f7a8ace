"""Compile the C restatement of the reference algorithm (test infrastructure / cpu_baseline).

Output: oracle/_build/liboracle.so (git-ignored via *.so; travels to the GPU box with the snapshot).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(SRC) > os.path.getmtime(OUT):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", SRC, "-o", OUT, "-lm"],
                       check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
