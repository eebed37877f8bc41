#!/bin/bash
# Developer aid: best register-staged vs best LDS-DMA variant per representative layer shape.
cd ${GRAFT_REPO_ROOT:-.}
for shape in "40 64 64 16 3 1" "40 128 128 8 3 1" "40 128 128 4 3 1" "40 128 128 2 3 1" "40 64 192 16 1 1" "40 128 128 16 3 1" "40 256 256 8 3 1" "20 128 128 128 3 1" "20 256 256 32 3 1" "20 384 384 16 3 1"; do
  echo "== $shape"
  python tools/conv_codes_check.py $shape 1 2>&1 | grep "^code" | sort -k8 -n | awk '{g=($6=="gl=0")?"reg":"dma"; if(!(g in seen)){seen[g]=1; print "   best " g ": " $0}}'
done
