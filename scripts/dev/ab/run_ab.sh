#!/bin/bash
# same-box A/B of library builds: scripts/dev/ab/*.so against the in-tree one
cd "$(dirname "$0")/../../.."
CFG=${1:-3}
cp safeopt_amd/libsafeopt_hip.so /tmp/cur.so
one() { python bench.py --config $1 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg', $1, j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'])"; }
for rep in 1 2; do
  for so in scripts/dev/ab/*.so /tmp/cur.so; do
    cp $so safeopt_amd/libsafeopt_hip.so
    echo "== $so"
    one $CFG
  done
done
cp /tmp/cur.so safeopt_amd/libsafeopt_hip.so
