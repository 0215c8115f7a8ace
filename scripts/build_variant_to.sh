#!/bin/bash
# Build a variant of libchx.so next to the product one: scripts/build_variant_to.sh <name> [-D flags...]
# -> chiron_amd/variants/libchx_<name>.so (git-ignored, travels with gpurun); select it with CHX_LIB=<path>.
name=$1; shift
cd "$(dirname "$0")/../chiron_amd/csrc"
mkdir -p ../variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -fPIC -shared -std=c++17 "$@" *.hip -o ../variants/libchx_$name.so
