#!/bin/bash
# PMC passes, one run per counter group (rocprofv3 does not split counters over passes); each under a hard
# time limit; the script stops at the first failure.  $1 = tag, $2.. = program and arguments to profile.
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift; ctrs=$1; shift
  echo "== pmc pass $name: $ctrs" | tee -a $OUT/steps.log
  timeout -s KILL ${PMC_LIMIT:-90} rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$name -o pmc -- "$@" > $OUT/$name.log 2>&1
  rc=$?
  echo "rc=$rc" | tee -a $OUT/steps.log
  tail -3 $OUT/$name.log
  return $rc
}
pass waves "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "$@" || exit 1
pass fetch "FETCH_SIZE" "$@" || exit 1
pass write "WRITE_SIZE" "$@" || exit 1
pass lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "$@" || exit 1
pass tcc "TCC_HIT_sum TCC_MISS_sum" "$@" || exit 1
echo pmc-done
