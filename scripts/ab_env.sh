#!/bin/bash
# A/B of ONE library under several environment settings on the steady-state pile (rocprofv3 kernel trace + wall):
#   scripts/ab_env.sh devlib/x outdir "VAR=a" "VAR=b" ...
set -o pipefail
cd $GRAFT_REPO_ROOT
LIB=$1; OUT=$2; shift 2
mkdir -p $OUT
timeout -k 10 200 python scripts/dump_scene.py pile 100000 /tmp/pile_steady.bin 600 300 > $OUT/dump.log 2>&1 || { tail $OUT/dump.log; exit 1; }
cp /tmp/pile_steady.bin $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
for kv in "$@"; do
  tag=$(echo "$kv" | tr '= ' '__')
  mkdir -p devlib/$tag && cp $LIB/libsfx.so devlib/$tag/libsfx.so
  echo "#### $kv"
  env $kv bash scripts/probe_trace.sh "devlib/$tag" $OUT/pile_steady.bin 200 0 $OUT || exit 1
  rm -rf devlib/$tag
done
rm -f $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
find $GRAFT_REPO_ROOT/$OUT -name "*kernel_trace.csv" -delete
