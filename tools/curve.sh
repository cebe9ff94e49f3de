#!/bin/bash
# usage: tools/curve.sh lib.so   -- decode kernel time of the FIRST k x 256 blocks of cfg2 (same data, k blocks per CU): what
# sharing a CU costs one block's serial chain.  Run on the GPU box.
for k in 1 2 3 4 6 8; do
  echo "== $1: $k block(s) per CU"
  DEC_NBLK=$((256 * k)) CBC_GPU_LIB=$GRAFT_REPO_ROOT/scratch/abl/$1 python $GRAFT_REPO_ROOT/tools/dec_bench.py 2>&1 | grep -v amdgpu.ids | grep "decode kernel ms"
done
