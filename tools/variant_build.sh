#!/bin/bash
# Cross-compiles diagnostic variants of the library HERE (no GPU needed): one source file rebuilt with extra flags, linked with
# the current objects of every other file.  Outputs build_variants/libcpg_<tag>.so (git-ignored; travels with gpurun).
#   bash tools/variant_build.sh gemm.hip noload "-DCPG_ABLATE=1"
# then on the GPU box:  CPG_LIB_PATH=build_variants/libcpg_noload.so python tools/kb.py
set -e
cd "$(dirname "$0")/.."
SRC=controlled-peptide-generation_amd/csrc
OBJ=controlled-peptide-generation_amd/_build
FILE=$1; TAG=$2; shift 2
mkdir -p build_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DCPG_DIAG "$@" -I $SRC -c $SRC/$FILE -o build_variants/$TAG.$FILE.o
OTHERS=$(ls $OBJ/*.hip.o | grep -v "/$FILE.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS build_variants/$TAG.$FILE.o -o build_variants/libcpg_$TAG.so
rm -f build_variants/$TAG.$FILE.o
echo built build_variants/libcpg_$TAG.so
