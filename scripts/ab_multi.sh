#!/bin/bash
# A/B of library variants, each under its own environment, on the steady-state pile (rocprofv3 kernel trace + wall):
#   scripts/ab_multi.sh outdir "devlib/a" "devlib/b VAR=1" ...
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $OUT
timeout -k 10 200 python scripts/dump_scene.py pile 100000 /tmp/pile_steady.bin 600 300 > $OUT/dump.log 2>&1 || { tail $OUT/dump.log; exit 1; }
cp /tmp/pile_steady.bin $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
for spec in "$@"; do
  set -- $spec
  lib=$1; shift
  echo "#### $lib $*"
  env "$@" bash scripts/probe_trace.sh "$lib" $OUT/pile_steady.bin 200 0 $OUT || exit 1
done
rm -f $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
find $GRAFT_REPO_ROOT/$OUT -name "*kernel_trace.csv" -delete
