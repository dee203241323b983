#!/bin/bash
# SFX_FUSE_TICK modes of ONE library on the BASELINE configs without joints (one box, one call):
#   scripts/ab_tick_modes.sh outdir "modes" configs...
# 0: k_integrate + a launch per stage; 3 (default): integration on load; 1: all substeps in one launch; 2: per substep a
# narrowphase launch + one launch of both sweeps.  Same bits in all four (tests/test_gpu_fused_tick.py).
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=$1; MODES=$2; shift 2
mkdir -p $OUT
for c in "$@"; do
  for m in $MODES; do
    echo "== $c SFX_FUSE_TICK=$m"
    SFX_FUSE_TICK=$m timeout -k 10 240 python scripts/configs.py $c 200 400 > $OUT/${c}_mode$m.log 2>&1 || { tail $OUT/${c}_mode$m.log; exit 1; }
    grep -E "^device|^pairs" $OUT/${c}_mode$m.log
  done
done
