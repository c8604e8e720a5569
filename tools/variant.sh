#!/bin/bash
# Builds a variant of libfx.so with extra -D flags for one kernel source (default fx_pca; SRC=fx_sil ... picks another):
#   tools/variant.sh <name> <flags...>  ->  build_dbg/libfx_<name>.so   (run with FX_LIB=build_dbg/libfx_<name>.so)
set -e
name=$1; shift
src=${SRC:-fx_pca}
cd "$(dirname "$0")/../image-processing_amd/csrc"
mkdir -p ../../build_dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
hipcc $F "$@" -c $src.hip -o ../../build_dbg/${src}_$name.o
others=""
for o in fx_api fx_bin fx_fps fx_hu fx_pca fx_png fx_render fx_sil fx_synth fx_io; do
    [ $o = $src ] || others="$others $o.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dbg/libfx_$name.so $others ../../build_dbg/${src}_$name.o -lpthread -lz -ldl
