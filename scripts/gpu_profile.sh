#!/bin/bash
# One GPU-box pass of the measurements the bench line cites: bench.py, the same command under the rocprofv3 kernel
# trace, the native driver, then the PMC passes.  Every GPU step has its own time limit; stops at the first failure.
# Output under gpurun_out/$1; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
step() { echo "== $*" | tee -a $OUT/steps.log; }
step bench
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
step rocprof-same
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_same -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-survey-scene > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { tail -20 $OUT/bench_under_rocprof.err; exit 1; }
cd $GRAFT_REPO_ROOT
python scripts/trace_window.py $(find $OUT/prof_same -name 'bench_kernel_trace.csv' | head -1) 200 70 > $OUT/bench_kernel_trace_settled.txt 2>&1 || true
cp $(find $OUT/prof_same -name 'bench_kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null || true
head -20 $OUT/bench_kernel_trace_settled.txt
rm -rf $OUT/prof_same   # the raw trace is tens of MB
step pmc
timeout -k 10 200 python scripts/dump_scene.py pile 100000 /tmp/pile_settled.bin 600 300 > $OUT/dump.log 2>&1 || { tail $OUT/dump.log; exit 1; }
timeout -k 10 120 tools/sfx_run /tmp/pile_settled.bin 240 0 > $OUT/sfx_run.log 2>&1 || { tail $OUT/sfx_run.log; exit 1; }
cat $OUT/sfx_run.log
PMC_LIMIT=150 bash scripts/gpu_pmc.sh $TAG/pmc $GRAFT_REPO_ROOT/tools/sfx_run /tmp/pile_settled.bin 4 0 > $OUT/pmc.log 2>&1 || { tail $OUT/pmc.log; exit 1; }
python scripts/pmc_report.py $OUT/pmc $OUT/pmc_pile100k.json $OUT/pmc_solve.json > $OUT/pmc_report.log 2>&1 || true
find $OUT/pmc -name '*.csv' -size +2M -delete
echo all-done
