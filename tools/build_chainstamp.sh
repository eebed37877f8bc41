#!/bin/bash
# Diagnostic build for tools/chain_stamps.py: the product objects with level_chain.hip recompiled with -DLFVDM_CHAIN_STAMP
# (per-stage, per-workgroup phase stamps of the persistent level chain) -> devlib/liblfvdm_chainstamp.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/latent-flexible-video-diffusion-modeling_amd
python3 $PKG/build.py > /dev/null
mkdir -p $ROOT/devlib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DLFVDM_CHAIN_STAMP -I $ROOT/include -I $PKG/csrc \
    -c $PKG/csrc/level_chain.hip -o $ROOT/devlib/level_chain_stamp.o
OBJS=$(ls $PKG/lib/*.o | grep -v level_chain.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/devlib/liblfvdm_chainstamp.so $OBJS $ROOT/devlib/level_chain_stamp.o
echo $ROOT/devlib/liblfvdm_chainstamp.so
