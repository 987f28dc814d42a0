#!/bin/bash
for v in "" "-DGSR_DBG_NOCOOP" "-DGSR_DBG_NOOWN"; do
  GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  timeout 120 python tools/dbg/train_kernels.py 2>&1 | grep -v amdgpu.ids | grep -o "^[0-9]* \|'tile_count[^}]*'tile_emit': [0-9.]*" | paste - -
done
