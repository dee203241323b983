#!/bin/bash
# times library variants (devlib/*/libsfx.so) on scene files: scripts/variants.sh "lib dirs" "scene files" [ticks] [settle]
set -o pipefail
for scene in $2; do
  for lib in $1; do
    echo "== $lib $scene"
    LD_LIBRARY_PATH=$lib timeout -k 10 150 tools/sfx_run $scene ${3:-100} ${4:-400} | tail -3 || exit 1
  done
done
