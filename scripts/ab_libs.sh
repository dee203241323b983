#!/bin/bash
# two libraries on the BASELINE configs, alternating, one box: scripts/ab_libs.sh libA.so libB.so outdir "ticks settle" configs...
set -o pipefail
cd $GRAFT_REPO_ROOT
A=$1; B=$2; OUT=$3; TS=$4; shift 4
mkdir -p $OUT
for c in "$@"; do
  for rep in 1 2; do
    for l in $A $B; do
      echo "== $c $l"
      SFX_LIBRARY=$GRAFT_REPO_ROOT/$l timeout -k 10 240 python scripts/configs.py $c $TS > $OUT/${c}_$(basename $(dirname $l))_$rep.log 2>&1 || { tail $OUT/${c}_$(basename $(dirname $l))_$rep.log; exit 1; }
      grep -E "^device" $OUT/${c}_$(basename $(dirname $l))_$rep.log
    done
  done
done
