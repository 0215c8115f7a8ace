#!/bin/bash
# Timing decomposition of the one-pass cell build (GPU box only): what does k_nl_build_cell_fl spend its time on?
# Every variant is a throw-away build of libchx.so; the product build is restored at the end.
set -e
cd "$(dirname "$0")/.."
export CHX_ALLOW_STALE_LIB=1
export CHX_BUILD_KERNEL=particle  # the switches live in the per-particle routine
for v in "" "-DCHX_EXP_BUILD_NOFL" "-DCHX_EXP_BUILD_NONL" "-DCHX_EXP_BUILD_NOSORT" "-DCHX_EXP_BUILD_NOFL -DCHX_EXP_BUILD_NONL -DCHX_EXP_BUILD_NOSORT" "-DCHX_EXP_BUILD_NOTEST"; do
  bash scripts/build_variant.sh $v
  echo "== variant: ${v:-product}"
  python scripts/exp_build.py c3 c4 2>&1 | grep chx_nl_build
done
python -c "from chiron_amd import build; build.build(force=True)"
