#!/bin/bash
# Build libchx.so with extra -D flags (experiments): scripts/build_variant.sh -DCHX_FL_UNROLL=8
# The result carries no source hash, so the loader refuses it unless CHX_ALLOW_STALE_LIB=1 is set
# (experiments set it; the product build -- chiron_amd/build.py -- always replaces such a binary).
cd "$(dirname "$0")/../chiron_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -fPIC -shared -std=c++17 "$@" *.hip -o ../libchx.so
