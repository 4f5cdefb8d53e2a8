#!/bin/bash
# Diagnostic build of the library with in-kernel phase stamps (-DVX_CONV_STAMPS) next to the product build:
# values_amd/libvalues_amd_stamps.so, objects in /tmp/vx_stamps_obj (the product objects are not touched).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OBJ=/tmp/vx_stamps_obj
mkdir -p $OBJ/values_amd/csrc $OBJ/include
cp $ROOT/values_amd/csrc/*.hip $ROOT/values_amd/csrc/*.h $ROOT/values_amd/csrc/*.cpp $ROOT/values_amd/csrc/Makefile $OBJ/values_amd/csrc/
cp $ROOT/include/values_amd.h $OBJ/include/
mkdir -p $OBJ/tools; cp $ROOT/tools/rsrc_table.py $OBJ/tools/
make -C $OBJ/values_amd/csrc -j8 EXTRA=-DVX_CONV_STAMPS OUT=$ROOT/values_amd/libvalues_amd_stamps.so
