"""Build libchx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries a sha256 over its sources and flags (``chx_source_hash()``): ``build()`` recompiles
exactly when that hash differs from the sources on disk -- file times are not consulted, so a checkout
that resets mtimes or a stale ``.so`` shipped beside edited sources cannot pass for a fresh build.
"""
from __future__ import annotations

import glob
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libchx.so")
# the step kernels once more as a bare code object: what the step queue (csrc/chx_aql.hip) loads through the HSA runtime
# and dispatches with its own AQL packets -- the device half of STEP_UNIT compiled by itself, same flags, same ISA
STEP_UNIT = "chx_lj.hip"
STEP_OUT = os.path.join(HERE, "libchx_step.hsaco")
# ... and the device halves of the units whose kernels ride on the queue while a run() is in progress (unpack, the list
# build chain, pack: csrc/chx_launch.h): libchx_k_<unit>.hsaco, loaded when present
QUEUE_UNITS = {"chx_core.hip": "core", "chx_neighbors.hip": "neighbors", "chx_cell_fl.hip": "cell_fl", "chx_mc.hip": "mc"}


def queue_object(unit: str) -> str:
    return os.path.join(HERE, f"libchx_k_{QUEUE_UNITS[unit]}.hsaco")

# -ffp-contract=off: the neighbor predicates and the BAOAB update must round once
# per operation, like the reference's un-fused fp32 ops.
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the wave instead
# of through a first scalar load (gfx940+); the step kernel leads with its control block and step index,
# which removes one dependent round trip from every launch (5.5 -> 5.25 us per step at N = 32768).
# -fno-slp-vectorize: the packed fp32 math of the list kernels is written out explicitly (f2 / v_pk_*); the SLP
# vectoriser adds nothing to it, but it does pair up scalar reads of the step kernel's by-value argument block
# (box half-lengths) into overlapping vector loads that SROA cannot split -- the leftover 12-byte array then lands in
# LDS, indexed by the flat work-item id, which makes every wavefront fetch the workgroup size from the AQL dispatch
# packet in HOST memory: 24-30 us per launch instead of 5 (measured in round 2).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-fPIC", "-shared", "-std=c++17"]
_MARK = b"CHX_SOURCE_HASH="


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _hashed_files():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def source_hash(flags=None) -> str:
    """sha256 over file names + contents of every source/header and over the compiler flags."""
    h = hashlib.sha256()
    for path in _hashed_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS if flags is None else flags).encode())
    return h.hexdigest()


def binary_hash(path: str = OUT):
    """The hash baked into a built library (read from the file, nothing is loaded); None if absent."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    m = re.search(re.escape(_MARK) + rb"([0-9a-f]{64})", blob)
    return m.group(1).decode() if m else None


def step_object_current() -> bool:
    """libchx_step.hsaco carries no hash of its own: a stamp file beside it names the tree it was made from."""
    try:
        with open(STEP_OUT + ".hash") as fh:
            return (fh.read().strip() == source_hash() and os.path.exists(STEP_OUT)
                    and all(os.path.exists(queue_object(u)) for u in QUEUE_UNITS))
    except OSError:
        return False


def needs_build() -> bool:
    return binary_hash() != source_hash() or not step_object_current()


def _object_key(src: str, flags) -> str:
    """Cache key of one translation unit: its own text, every header (any of them may be included) and the flags."""
    h = hashlib.sha256()
    for path in [src] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:24]


def build(force: bool = False, verbose: bool = False) -> str:
    """Translation units are compiled in parallel into ``_obj/`` (kept out of git and off the GPU box), each keyed by
    the hash of its text + headers + flags, so an edit recompiles one file; the link step bakes the tree's hash in
    through a one-line stamp unit."""
    if not force and not needs_build():
        if verbose:
            print(f"libchx.so is current (source hash {source_hash()[:16]})")
        return OUT
    import fcntl
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    # One builder at a time (several ranks may call build() at once): the others wait here and then find the library
    # current.  Temporary files carry the pid besides, so that even builders that cannot share the lock (another
    # filesystem view) never write through each other's half-made files.
    lock = open(os.path.join(objdir, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not needs_build():
            return OUT
        return _build_locked(hipcc, objdir, force, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc: str, objdir: str, force: bool, verbose: bool) -> str:
    from concurrent.futures import ThreadPoolExecutor

    cflags = [f for f in FLAGS if f != "-shared"]
    tree = source_hash()
    tag = f".{os.getpid()}.tmp"

    def compile_one(src):
        # (the hash string is compiled into chx_core.hip only: every other unit is independent of the tree's hash)
        extra = [f'-DCHX_SOURCE_HASH="{tree}"'] if os.path.basename(src) == "chx_core.hip" else []
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + "." + _object_key(src, cflags + extra) + ".o")
        if force or not os.path.exists(obj):
            cmd = [hipcc] + cflags + extra + ["-c", src, "-o", obj + tag]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(obj + tag, obj)
        return obj

    def compile_step_object(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + "." + _object_key(src, cflags) + ".hsaco")
        if force or not os.path.exists(obj):
            cmd = [hipcc] + [f for f in cflags if f != "-fPIC"] + ["--cuda-device-only", "--no-gpu-bundle-output", src, "-o", obj + tag]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(obj + tag, obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        step_job = pool.submit(compile_step_object, os.path.join(CSRC, STEP_UNIT))
        queue_jobs = {u: pool.submit(compile_step_object, os.path.join(CSRC, u)) for u in QUEUE_UNITS}
        objs = list(pool.map(compile_one, sources()))
        step_obj = step_job.result()
        queue_objs = {u: j.result() for u, j in queue_jobs.items()}
    for stale in set(glob.glob(os.path.join(objdir, "*.hsaco"))) - {step_obj} - set(queue_objs.values()):
        os.remove(stale)
    for src_obj, dst in [(step_obj, STEP_OUT)] + [(o, queue_object(u)) for u, o in queue_objs.items()]:
        shutil.copyfile(src_obj, dst + tag)
        os.replace(dst + tag, dst)
    for stale in set(glob.glob(os.path.join(objdir, "*.o"))) - set(objs):  # (under the lock: nobody is about to link them)
        os.remove(stale)
    tmp = OUT + tag
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)  # never truncate a library some process may have mapped
    if binary_hash(OUT) != tree:
        raise RuntimeError("libchx.so does not carry the hash of the tree it was built from")
    with open(STEP_OUT + ".hash" + tag, "w") as fh:
        fh.write(tree + "\n")
    os.replace(STEP_OUT + ".hash" + tag, STEP_OUT + ".hash")
    return OUT
