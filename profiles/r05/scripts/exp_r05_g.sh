#!/bin/bash
# round 5, block g: where the one-launch step of a small grid spends its time (-DSTEP_STAMPS)
cd "$(dirname "$0")/../../.."
SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/stepstamps.so SGP_STEP_STAMPS=1 python scripts/dev/small_step_time.py 2>&1 | grep -E "stamps|grid" | awk '/stamps/{c++; if (c%4==0) print; next} {print}' | head -40
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "long_axis or prefix_of_the_points" 2>&1 | tail -4
