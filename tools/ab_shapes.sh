#!/bin/bash
# Developer aid: best LDS-DMA variant per large layer shape for two builds of the library.
cd ${GRAFT_REPO_ROOT:-.}
for lib in "$1" "$2"; do
  export LFVDM_LIB_PATH=$PWD/$lib
  for shape in "40 128 128 16 3 1" "40 256 256 8 3 1" "20 128 128 128 3 1" "20 256 256 32 3 1" "20 384 384 16 3 1"; do
    python tools/conv_codes_check.py $shape 1 2>&1 | grep "^code" | grep -v "gl=0" | sort -k8 -n | head -1 | sed "s|^|$(basename $lib) [$shape] |"
  done
done
