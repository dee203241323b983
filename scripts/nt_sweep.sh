#!/bin/bash
# times the tile-solve workgroup size variants (SFX_TILE_NT) on scene files: scripts/nt_sweep.sh "scene files" [ticks] [settle]
set -o pipefail
for scene in $1; do
  for nt in auto 512 256 128; do
    echo "== NT $nt $scene"
    if [ $nt = auto ]; then timeout -k 10 150 tools/sfx_run $scene ${2:-100} ${3:-400} | tail -3 || exit 1
    else SFX_TILE_NT=$nt timeout -k 10 150 tools/sfx_run $scene ${2:-100} ${3:-400} | tail -3 || exit 1; fi
  done
done
