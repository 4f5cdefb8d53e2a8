#!/bin/bash
# Build the working tree's kernel sources with extra compiler flags into values_amd/libvalues_amd_<name>.so (objects in
# /tmp, the product objects are not touched) -- for same-process A/B with tools/ab_layers.py --base / --new:
#   tools/build_variant.sh pin1 -DVX_S16_PINMODE=1
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OBJ=/tmp/vx_variant_$NAME
rm -rf $OBJ; mkdir -p $OBJ/values_amd/csrc $OBJ/include
cp $ROOT/values_amd/csrc/*.hip $ROOT/values_amd/csrc/*.h $ROOT/values_amd/csrc/*.cpp $ROOT/values_amd/csrc/Makefile $OBJ/values_amd/csrc/
cp $ROOT/include/values_amd.h $OBJ/include/
mkdir -p $OBJ/tools; cp $ROOT/tools/rsrc_table.py $OBJ/tools/      # (the Makefile prints each object's diagnostics through it)
make -C $OBJ/values_amd/csrc -j8 EXTRA="$*" OUT=$ROOT/values_amd/libvalues_amd_$NAME.so > /dev/null
ls -la $ROOT/values_amd/libvalues_amd_$NAME.so
