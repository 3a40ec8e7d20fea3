#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04n; mkdir -p $OUT; cd $R
SAFEOPT_HIP_LIB=scripts/dev/ab/stamps.so timeout 300 python scripts/dev/small_n.py 8 20 64 200 2>&1 | awk '/^stamps/{last=$0} !/^stamps/{if(last!="")print last; last=""; print}' | tee $OUT/stamps_small_n.txt
