#!/bin/bash
# theta = 1 frames (tools/bh_ticks.py) with a `make variant` build and with the shipped library, same box, alternating.
#   bash tools/ab_bh_variant.sh VARIANT "N K [scene]" ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
V="$1"; shift
for spec in "$@"; do
  set -- $spec
  for rep in 1 2; do
    for lib in "$V" shipped; do
      if [ $lib = shipped ]; then unset NBODY_AMD_LIB; else export NBODY_AMD_LIB="$ROOT/parallelnbody_amd/libnbody_amd.$V.so"; fi
      echo "[$lib] $(python3 "$ROOT/tools/bh_ticks.py" $1 $2 step 1.0 ${3:-})"
    done
  done
done
