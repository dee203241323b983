#!/bin/bash
# the BASELINE configs under two settings of one environment variable: scripts/ab_configs.sh lib.so outdir VAR a b [configs...]
set -o pipefail
cd $GRAFT_REPO_ROOT
LIB=$1; OUT=$2; VAR=$3; A=$4; B=$5; shift 5
mkdir -p $OUT
for c in "$@"; do
  for v in $A $B; do
    echo "== $c $VAR=$v"
    env SFX_LIBRARY=$GRAFT_REPO_ROOT/$LIB $VAR=$v timeout -k 10 240 python scripts/configs.py $c 200 400 > $OUT/${c}_$v.log 2>&1 || { tail $OUT/${c}_$v.log; exit 1; }
    grep -E "^device|^pairs" $OUT/${c}_$v.log
  done
done
