#!/bin/bash
# tools/ab_build.sh <tag> <file.hip> [extra hipcc flags...] -- A/B build: libacdsp_<tag>.so with ONE source recompiled
# with extra flags (ablation macros); select it with ACDSP_LIB=ac_dsp_amd/lib/libacdsp_<tag>.so
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
make -s -j8 >/dev/null
O=/tmp/ab_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c ac_dsp_amd/csrc/$SRC -o $O 2>/dev/null
OBJS=$(ls ac_dsp_amd/csrc/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o ac_dsp_amd/lib/libacdsp_$TAG.so
echo built ac_dsp_amd/lib/libacdsp_$TAG.so
