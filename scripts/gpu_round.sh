#!/bin/bash
# One GPU-box pass: GPU tests, bench, rocprofv3 kernel stats, then PMC probes.  Every GPU step has its
# own time limit; the script stops at the first failure (no retries).  Output under gpurun_out/$1.
set -o pipefail
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
step() { echo "== $*" | tee -a $OUT/steps.log; }
step tests
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/gputests.log 2>&1 || { tail -30 $OUT/gputests.log; exit 1; }
tail -3 $OUT/gputests.log
step bench
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
step rocprof-stats
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 20 --no-cpu --settle-frames 400 > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err || { tail -20 $OUT/bench_rocprof.err; exit 1; }
cat $OUT/bench_rocprof.json
step rocprof-same
# the default bench command itself under the kernel trace (CPU leg off): scripts/trace_window.py reads its settled ticks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_same -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-survey-scene > $OUT/bench_rocprof_same.json 2> $OUT/bench_rocprof_same.err || { tail -20 $OUT/bench_rocprof_same.err; exit 1; }
step native
cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/sfx_run data/pile100k.bin 200 400 > $OUT/sfx_run.log 2>&1 || { tail $OUT/sfx_run.log; exit 1; }
cat $OUT/sfx_run.log
step pmc
timeout -k 10 200 python scripts/dump_scene.py pile 100000 /tmp/pile_settled.bin 600 300 > $OUT/dump.log 2>&1 || { tail $OUT/dump.log; exit 1; }
PMC_LIMIT=150 bash scripts/gpu_pmc.sh $TAG/pmc $GRAFT_REPO_ROOT/tools/sfx_run /tmp/pile_settled.bin 4 0 > $OUT/pmc.log 2>&1 || { tail $OUT/pmc.log; exit 1; }
tail -2 $OUT/pmc.log
echo all-done
