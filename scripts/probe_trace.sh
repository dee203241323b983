#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of library variants on one scene file:
#   scripts/probe_trace.sh "lib dirs" scene.bin ticks settle outdir
set -o pipefail
OUT=$GRAFT_REPO_ROOT/${5:-gpurun_out/probe}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $1; do
  tag=$(basename $lib)
  export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/$lib
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- $GRAFT_REPO_ROOT/tools/sfx_run $GRAFT_REPO_ROOT/$2 ${3:-40} ${4:-0} > $OUT/$tag.log 2>&1 || { tail $OUT/$tag.log; exit 1; }
  echo "== $tag"; tail -2 $OUT/$tag.log | head -1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void sfx::", "")[:40]
    if "k_" in n:
        print(f"   {n:40s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
