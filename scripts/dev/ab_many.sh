#!/bin/bash
# same-box A/B of library variants, interleaved repetitions:
#   scripts/dev/ab_many.sh "cur v1 v2" "3 4 5" [reps]
cd "$(dirname "$0")/../.."
export AB_ONLY=${AB_ONLY:-pair}
VARS=$1; CFGS=${2:-"3 4 5"}; REPS=${3:-3}
for rep in $(seq $REPS); do
for v in $VARS; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 300 python scripts/dev/ab_sweep.py $CFGS 2>&1 | grep "^cfg"
done; done | sort -k2,2n -k8,8 -s | awk '{k=$2" "$NF; s[k]+=$4; n[k]++; if(!(k in mn)||$4<mn[k])mn[k]=$4} END{for(k in s) printf "cfg %s  mean %.3f ms  min %.3f ms  (n=%d)\n", k, s[k]/n[k], mn[k], n[k]}' | sort
