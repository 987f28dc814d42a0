#!/bin/bash
# usage: tools/dbg/run_sweeps.sh <tag>   (GPU box) -- scene sweep + train step bench + short bench
tag=$1
mkdir -p gpurun_out/r02
python tools/scene_sweep.py > gpurun_out/r02/sweep_$tag.log 2>&1; grep -v amdgpu.ids gpurun_out/r02/sweep_$tag.log
python tools/train_step_bench.py > gpurun_out/r02/train_$tag.log 2>&1; grep -v amdgpu.ids gpurun_out/r02/train_$tag.log
