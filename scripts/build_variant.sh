#!/bin/bash
# Build an experiment variant of the library next to the product one: scripts/variants/libmrla_hip_<name>.so, with extra
# compiler flags for ONE source file (the others are taken from the product build).  The micro-benchmarks load it with
# KBENCH_LIB=<path>; the product never does.
# Usage: bash scripts/build_variant.sh <name> <source.hip> "<extra flags>"     e.g.  fused4w light_nhwc_wide.hip "-DMRLA_FUSED_MAXWAVES=4"
set -eu
NAME=$1; SRC=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mrla_amd/csrc"
make -j4 > /dev/null
mkdir -p build/variants "$ROOT/scripts/variants"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wall -Wno-unused-function"
case $SRC in light_nhwc_bwd.hip|light_nhwc_wide.hip|light_nhwc_lean.hip|tokens_nhwc.hip) FLAGS="$FLAGS -fno-slp-vectorize";; esac
OBJ=build/variants/${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o $OBJ
OTHERS=$(ls build/*.o | grep -v "build/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJ -o "$ROOT/scripts/variants/libmrla_hip_${NAME}.so"
ls -la "$ROOT/scripts/variants/libmrla_hip_${NAME}.so"
