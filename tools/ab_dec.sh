#!/bin/bash
# usage: tools/ab_dec.sh lib1.so lib2.so ...   (libs in scratch/abl): decode kernel ms on cfg2, same box
for l in "$@"; do echo "== $l"; CBC_GPU_LIB=$GRAFT_REPO_ROOT/scratch/abl/$l python $GRAFT_REPO_ROOT/tools/dec_bench.py 2>&1 | tail -1; done
