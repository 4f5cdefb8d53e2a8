#!/bin/bash
# Fails when any kernel instance of libvalues_amd.so carries scratch (register spills into memory), from the per-object
# resource reports the Makefile leaves next to the objects (hipcc -Rpass-analysis=kernel-resource-usage).
# Run by __graft_entry__.build() after make.  `tools/check_spills.sh -v` prints the whole table.
set -e
cd "$(dirname "$0")/../values_amd/csrc"
# read-only: an object without its report (objects copied from elsewhere, reports cleaned) fails the check -- run make
for src in *.hip; do
  base="${src%.hip}"
  if [ -f "$base.o" ] && [ ! -f "$base.rsrc" ]; then echo "check_spills: $base.o has no resource report ($base.rsrc): make -B $base.o"; exit 1; fi
done
ls *.rsrc >/dev/null 2>&1 || { echo "check_spills: no .rsrc reports (run make first)"; exit 1; }
if [ "$1" = "-v" ]; then python3 ../../tools/rsrc_table.py *.rsrc; fi
python3 ../../tools/rsrc_table.py --fail *.rsrc | grep "SCRATCH" || true
python3 ../../tools/rsrc_table.py --fail *.rsrc > /dev/null
echo "check_spills: no kernel instance with scratch ($(cat *.rsrc | grep -c 'Function Name') kernels)"
