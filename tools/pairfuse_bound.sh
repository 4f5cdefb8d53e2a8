#!/bin/bash
# Evidence for DESIGN.md section 5d (the decoder conv-pair fusion that round 4 did NOT build): phase ablations of the three
# full-resolution z-column launches in the diagnostic build (tools/build_stamps.sh; VX_XP_ABL switches phases off -- wrong
# results by design, diagnostic library only), 320 samples at 64^3 as the bench runs them.
#   VX_XP_ABL bits: 1 no multiply, 2 no epilogue, 4 no commit (staging -> LDS), 8 no loads
cd "$(dirname "$0")/.."
export STAMP_N=${STAMP_N:-320} STAMP_REPS=${STAMP_REPS:-60}
E11=16:8:64:1:1:0:1:1     # upscale2 + expand_1_1: two chunks, fused up-convolution, normalise-on-load skip half
E12=8:8:64:1:1:1          # expand_1_2 + final: LeakyReLU + hash dropout + fused head
C12=8:8:64:0:0:0:0:2:1    # contr_1_2: pre-split input, statistics + pooled output
for spec in $E11 $E12 $C12; do
  echo "=== $spec"
  for abl in 0 14 12 3 2 8; do
    case $abl in 0) what="everything";; 14) what="multiply only";; 12) what="consumers only (multiply + epilogue)";;
      3) what="producers only (loads + commit)";; 2) what="no epilogue";; 8) what="no loads";; esac
    echo -n "abl=$abl ($what): "
    VX_XP_ABL=$abl STAMP_TERSE=1 python3 tools/stamp_s16.py $spec | sed 's/.*: \([0-9.]* ms\/launch\).*/\1/'
  done
  echo "--- per-role stamps, everything on"
  VX_XP_ABL=0 python3 tools/stamp_s16.py $spec | tail -n +2
done
