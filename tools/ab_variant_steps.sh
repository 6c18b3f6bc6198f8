#!/bin/bash
# Whole steps (tools/even_vs_guided.py --default-only: no events, distinct and equal masses) with the shipped library and with a
# `make variant` build next to it, same box.   bash tools/ab_variant_steps.sh VARIANT_NAME N [N ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
V="$1"; shift
for lib in shipped "$V"; do
  if [ $lib = shipped ]; then unset NBODY_AMD_LIB; else export NBODY_AMD_LIB="$ROOT/parallelnbody_amd/libnbody_amd.$V.so"; fi
  echo "## library: $lib"
  python3 "$ROOT/tools/even_vs_guided.py" --default-only "$@"
done
