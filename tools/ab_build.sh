#!/bin/bash
# Same-box A/B of two builds of the library: builds the kernel sources of a git revision (default HEAD) into
# values_amd/libvalues_amd_base.so (objects under /tmp; the working tree's own build is not touched).  The GPU box then
# runs `VX_LIB_PATH=values_amd/libvalues_amd_base.so python bench.py ...` next to the plain command.
set -e
REV="${1:-HEAD}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OBJ=/tmp/vx_ab_obj
rm -rf $OBJ && mkdir -p $OBJ
git -C "$ROOT" archive "$REV" values_amd/csrc include tools/rsrc_table.py | tar -x -C $OBJ
make -C $OBJ/values_amd/csrc -j8 OUT=$ROOT/values_amd/libvalues_amd_base.so > /dev/null
ls -la $ROOT/values_amd/libvalues_amd_base.so
