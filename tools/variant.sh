#!/bin/bash
# Builds a variant of libfx.so with extra -D flags for fx_pca.hip: tools/variant.sh <name> <flags...> -> build_dbg/libfx_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../image-processing_amd/csrc"
mkdir -p ../../build_dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
hipcc $F "$@" -c fx_pca.hip -o ../../build_dbg/fx_pca_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dbg/libfx_$name.so fx_api.o fx_bin.o fx_fps.o fx_hu.o fx_render.o fx_sil.o fx_synth.o fx_io.o ../../build_dbg/fx_pca_$name.o -lpthread -lz -ldl
