#!/bin/bash
# usage (on the GPU box, through gpurun): tools/gpu_round.sh <tag> [tests|notests]
# -> gpurun_out/<tag>_gpu_tests.log, then the profile set of tools/profile_round.sh
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest $R/tests -m gpu -x -q > $O/${tag}_gpu_tests.log 2>&1 || { tail -30 $O/${tag}_gpu_tests.log; exit 1; }
  tail -3 $O/${tag}_gpu_tests.log
fi
timeout -k 10 600 bash $R/tools/profile_round.sh $tag
