#!/bin/bash
# same-box A/B of library variants x schedules, interleaved repetitions:
#   scripts/dev/ab_libs.sh "cur r05 v1" "pair pair-unmerged" "3 4 5" [reps]
cd "$(dirname "$0")/../.."
VARS=$1; MODES=${2:-pair}; CFGS=${3:-"3"}; REPS=${4:-2}
for rep in $(seq $REPS); do
for v in $VARS; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  for m in $MODES; do
    SAFEOPT_HIP_LIB=$lib AB_ONLY=$m AB_TAG="$v/$m" timeout 300 python scripts/dev/ab_sweep.py $CFGS 2>&1 | grep "^cfg"
  done
done; done | awk '{k=$2" "$NF; s[k]+=$4; n[k]++; if(!(k in mn)||$4<mn[k])mn[k]=$4} END{for(k in s) printf "cfg %s  mean %.3f ms  min %.3f ms  (n=%d)\n", k, s[k]/n[k], mn[k], n[k]}' | sort
