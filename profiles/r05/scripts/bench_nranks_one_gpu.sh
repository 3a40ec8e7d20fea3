#!/bin/bash
# 2 and 8 ranks on the one GPU (SAFEOPT_HIP_DEVICE=0 for every rank): RCCL cannot come up with
# two ranks on one device -> the chain must fall through to the TCP transport and still print a line
cd "$(dirname "$0")/../../.."
export SAFEOPT_HIP_DEVICE=0
for n in 2 8; do
  timeout 800 python bench.py --gpus $n --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_${n}ranks_one_gpu.json 2> gpurun_out/bench_${n}ranks_one_gpu.err
  echo "n=$n rc=$?"; python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/bench_${n}ranks_one_gpu.json").read().strip().splitlines()[-1])
    print({k:r.get(k) for k in ("n_gpus","ms_per_step","value","transport","nrank_selfcheck","nrank_step","chosen_index")})
    print("config4_strong", r.get("config4_strong"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/bench_${n}ranks_one_gpu.err").read()[-2000:])
PY
done
