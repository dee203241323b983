#!/bin/bash
# timing matrix over solve workgroup size (library variant), tile width and batch group
set -e
for lib in starframe_amd/csrc devlib/nt256; do
  for tc in 0 1; do
    for grp in 0 1 2 4; do
      echo "== lib $lib tile_cells $tc group $grp"
      if [ $grp = 0 ]; then
        LD_LIBRARY_PATH=$lib timeout -k 10 120 tools/sfx_run data/pile100k.bin 50 400 $tc 0 | tail -3
      else
        SFX_GROUP=$grp LD_LIBRARY_PATH=$lib timeout -k 10 120 tools/sfx_run data/pile100k.bin 50 400 $tc 0 | tail -3
      fi
    done
  done
done
