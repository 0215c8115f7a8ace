"""A/B builds of libchx for kernel experiments:  python scripts/build_variant.py NAME [-DMACRO=VALUE ...]
Compiles chx_lj.hip with the extra flags (the other translation units are taken from the current in-tree build) into
scripts/_variants/NAME/{libchx.so, libchx_step.hsaco, libchx_k_*.hsaco}; run with CHX_LIB=scripts/_variants/NAME/libchx.so
(the library finds its code objects beside itself; the loader does not compare hashes for a CHX_LIB library)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chiron_amd import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
unit = os.environ.get("VARIANT_UNIT", "chx_lj.hip")
b.build()
out = os.path.join(ROOT, "scripts", "_variants", name)
os.makedirs(out, exist_ok=True)
cflags = [f for f in b.FLAGS if f != "-shared"]
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
objdir = os.path.join(b.HERE, "_obj")
src = os.path.join(b.CSRC, unit)
obj = os.path.join(out, unit[:-4] + ".o")
hs = os.path.join(out, unit[:-4] + ".hsaco")
p1 = subprocess.Popen([hipcc] + cflags + extra + ["-c", src, "-o", obj])
p2 = subprocess.Popen([hipcc] + [f for f in cflags if f != "-fPIC"] + extra + ["--cuda-device-only", "--no-gpu-bundle-output", src, "-o", hs])
assert p1.wait() == 0 and p2.wait() == 0
others = [o for o in glob.glob(os.path.join(objdir, "*.o")) if not os.path.basename(o).startswith(unit[:-4] + ".")]
subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + others + [obj, "-o", os.path.join(out, "libchx.so")], check=True)
for u in b.QUEUE_UNITS:
    shutil.copyfile(b.queue_object(u), os.path.join(out, os.path.basename(b.queue_object(u))))
shutil.copyfile(b.STEP_OUT, os.path.join(out, "libchx_step.hsaco"))
if unit == b.STEP_UNIT:
    shutil.copyfile(hs, os.path.join(out, "libchx_step.hsaco"))
elif unit in b.QUEUE_UNITS:
    shutil.copyfile(hs, os.path.join(out, os.path.basename(b.queue_object(unit))))
print("built", out)
