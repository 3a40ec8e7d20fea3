#!/bin/bash
# "What does the sweep cost without X": rebuilds the library with
# -DSGP_INSTRUMENT, times the sweep under every mask, restores the normal build.
#   bash scripts/ablate.sh [configs...]      (default 2 3 4; MASKS="0 1 2 ..." overrides)
cd "$(dirname "$0")/.."
SGP_HIPCC_FLAGS=-DSGP_INSTRUMENT python -m safeopt_amd.build --force > /dev/null || exit 1
for c in ${@:-2 3 4}; do for a in ${MASKS:-0 1 2 4 8 7 10 12 15}; do
  SGP_ABLATE=$a timeout 120 python scripts/ablate.py $c 2>&1 | tail -1
done; done
python -m safeopt_amd.build --force > /dev/null
