#!/bin/bash
# AddressSanitizer + UBSan build of libchx's HOST side, plus the driver that exercises the host half of the C ABI
# (tests/host_asan/driver.cpp).  The sanitizers apply to the host code of every .hip file (entry points, argument
# validation, launch plumbing, host helpers); the gfx950 code objects are compiled as usual (device ASan needs
# xnack+, which this pool does not offer -- clang says so and ignores the option for the offload arch).  No GPU is
# needed to build or to run the driver: it never reaches a kernel launch.
#   scripts/build_host_asan.sh [outdir]    -> outdir/libchx_host_asan.so, outdir/host_asan_driver
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/tests/host_asan/_build}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -Wno-option-ignored"
"$HIPCC" --offload-arch=gfx950 $SAN -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -std=c++17 \
  "$ROOT"/chiron_amd/csrc/*.hip -o "$OUT/libchx_host_asan.so"
"$HIPCC" --offload-arch=gfx950 $SAN -std=c++17 -x hip "$ROOT/tests/host_asan/driver.cpp" -L"$OUT" -lchx_host_asan \
  -Wl,-rpath,"$OUT" -o "$OUT/host_asan_driver"
echo "$OUT/host_asan_driver"
