#!/bin/bash
# round 5, block h: the one-launch step with 1024 threads (128 registers, small spills) for n <= 24
cd "$(dirname "$0")/../../.."
for v in stepstamps step1024; do echo "== $v"
SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/$v.so SGP_STEP_STAMPS=1 python scripts/dev/small_step_time.py 2>&1 | grep -E "stamps|grid" | awk '/stamps/{c++; if (c%4==0) print; next} {print}' | head -12
done
