#!/bin/bash
# tiles per batch (SFX_GROUP) x tile-solve workgroup size (SFX_TILE_NT) on one scene file
set -o pipefail
for nt in 512 256; do
  for grp in 1 2 3 4 6; do
    echo "== NT $nt GROUP $grp"
    SFX_GROUP=$grp SFX_TILE_NT=$nt timeout -k 10 150 tools/sfx_run $1 ${2:-100} ${3:-400} | tail -3 || exit 1
  done
done
