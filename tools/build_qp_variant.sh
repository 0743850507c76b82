#!/bin/bash
# Builds robobee3d_amd/variants/libumpc_<name>.so from a scratch COPY of the tree under generator switches taken from the
# environment (UMPC_QP_*, UMPC_ASM*_*), for A/B timing inside ONE gpurun call (select with UMPC_LIB=robobee3d_amd/variants/...).
# The tracked headers and the shipped library are never touched (_lib.build() refuses switches outside UMPC_VARIANT_ROOT).
# usage: UMPC_QP_TIMING=1 tools/build_qp_variant.sh timing
set -e
NAME=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EXP=/tmp/umpc_variant_$NAME
rm -rf "$EXP"; mkdir -p "$EXP"
cp -r "$ROOT/robobee3d_amd" "$ROOT/include" "$EXP/"
rm -rf "$EXP/robobee3d_amd/variants"
(cd "$EXP" && UMPC_VARIANT_ROOT="$EXP" python3 -c "import sys; sys.path.insert(0, '.'); from robobee3d_amd import _lib; print(_lib.build())")
mkdir -p "$ROOT/robobee3d_amd/variants"
cp "$EXP/robobee3d_amd/libumpc_mi355x.so" "$ROOT/robobee3d_amd/variants/libumpc_$NAME.so"
echo "$ROOT/robobee3d_amd/variants/libumpc_$NAME.so"
