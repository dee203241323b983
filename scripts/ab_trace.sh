#!/bin/bash
# A/B of library variants on the steady-state pile (per-kernel rocprofv3 trace + wall):
#   scripts/ab_trace.sh "devlib/base devlib/new" outdir [width]
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=${2:-gpurun_out/ab}
mkdir -p $OUT
timeout -k 10 200 python scripts/dump_scene.py pile 100000 /tmp/pile_steady.bin 600 300 > $OUT/dump.log 2>&1 || { tail $OUT/dump.log; exit 1; }
cp /tmp/pile_steady.bin $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
bash scripts/probe_trace.sh "$1" $OUT/pile_steady.bin 200 0 $OUT
rm -f $GRAFT_REPO_ROOT/$OUT/pile_steady.bin
find $GRAFT_REPO_ROOT/$OUT -name "*kernel_trace.csv" -delete
