#!/bin/bash
# culling property test: product build vs a build without any culling must give bit-identical images (tools/cull_check.py)
python gs_localization_amd/build.py > /dev/null 2>&1
CASES=${CASES:-200} timeout 600 python tools/cull_check.py dump gpurun_out/cull 2>&1 | grep -v amdgpu.ids | tail -2
GSR_DEFS="-DGSR_NO_CULL" python gs_localization_amd/build.py > /dev/null 2>&1
CASES=${CASES:-200} timeout 600 python tools/cull_check.py compare gpurun_out/cull 2>&1 | grep -v amdgpu.ids | tail -3
python gs_localization_amd/build.py > /dev/null 2>&1
