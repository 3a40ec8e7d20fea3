#!/bin/bash
# Build a variant of the library with extra flags for the sweep translation units:
#   scripts/dev/build_variant.sh NAME "-DPGP_ORDER=1 ..." [pair]   -> scripts/dev/ab/NAME.so
# (the other objects come from the in-tree build; run python -m safeopt_amd.build first;
# "pair": only sweep_pair.hip is rebuilt)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2; ONLY=$3
C=safeopt_amd/csrc; O=/tmp/variant_$NAME; mkdir -p $O scripts/dev/ab
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -Wno-inline-asm"
/opt/rocm/bin/hipcc $BASE $FLAGS -c $C/sweep_pair.hip -o $O/sweep_pair.o &
if [ "$ONLY" = pair ]; then cp $C/sweep.o $O/sweep.o; else
/opt/rocm/bin/hipcc $BASE $FLAGS -mllvm -amdgpu-spill-vgpr-to-agpr=0 -c $C/sweep.hip -o $O/sweep.o &
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/ab/$NAME.so \
  $C/api.o $O/sweep.o $O/sweep_pair.o $C/sweep_mid.o $C/sweep_tiny.o $C/step_small.o $C/factor.o $C/sets.o $C/swarm.o -ldl
echo built scripts/dev/ab/$NAME.so
