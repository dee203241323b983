#!/bin/bash
# workgroup size of the position / velocity tile solves separately, x tiles per batch, on one scene file
set -o pipefail
for grp in ${4:-0}; do
for p in 512 256; do for v in 512 256; do
  echo "== POS $p VEL $v GROUP $grp"
  if [ $grp = 0 ]; then SFX_TILE_NT_POS=$p SFX_TILE_NT_VEL=$v timeout -k 10 150 tools/sfx_run $1 ${2:-100} ${3:-400} | tail -2 || exit 1
  else SFX_GROUP=$grp SFX_TILE_NT_POS=$p SFX_TILE_NT_VEL=$v timeout -k 10 150 tools/sfx_run $1 ${2:-100} ${3:-400} | tail -2 || exit 1; fi
done; done; done
