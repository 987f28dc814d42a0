#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box, via gpurun): tools/dbg/gpu_session.sh <step> [...]  -- the round's GPU sessions, one named step per call;
# everything lands in gpurun_out/r04/<step>*.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
step=$1; shift
case $step in
  newtests)
    python -m pytest tests/test_gpu_refine.py::test_host_redo_of_a_warm_started_call_survives_an_overflowing_complete_list_bin \
      tests/test_gpu_parity.py::test_precomputed_inputs_mode tests/test_gpu_parity.py::test_isotropic_scaling_through_the_pose_package \
      tests/test_gpu_fuzz.py tests/test_gpu_multirank.py::test_rccl_path_executes_with_one_rank -q -x 2>&1 | tail -25 > $o/newtests.log ;;
  suite)
    python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $o/suite$1.log ;;
  profiles)      # evidence for BASELINE configs 3 and 4
    tag=${1:-r04}
    tools/profile_r04.sh $tag cam_plain s_3m_cam plain
    tools/profile_r04.sh $tag cam_spec s_3m_cam
    tools/profile_r04.sh $tag cam1024_plain s_3m_cam_1024 plain
    tools/profile_r04.sh $tag cam1024_spec s_3m_cam_1024
    tools/profile_r04.sh $tag train train
    tools/profile_r04.sh $tag s1m_plain s_1m_640 plain ;;
  bench)
    python bench.py "$@" 2>$o/bench$BTAG.err | tail -1 > $o/bench$BTAG.json ;;
  plain)         # quick A/B numbers of the complete-list path: it/s + per-kernel HIP-event times
    for sc in s_1m_640 s_3m_cam s_3m_cam_1024; do SCENE=$sc LOOP_PLAIN=1 python tools/loop_only.py 100 2>/dev/null | tail -1; done > $o/plain$1.log
    python tools/dbg/train_kernels.py 2>/dev/null | tail -2 >> $o/plain$1.log ;;
  timing)        # phase clocks of the complete-list path (diagnostic build; the box is thrown away afterwards)
    export GSR_LIB_PATH=/tmp/gsr_timing.so; GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
    for sc in ${@:-s_1m_640 s_3m_cam}; do echo "== $sc plain"; SCENE=$sc LOOP_PLAIN=1 python tools/phase_timing.py 2>/dev/null | grep -v amdgpu; done > $o/timing_plain.log
    unset GSR_LIB_PATH ;;
  tail)          # the slowest waves of k_render_fwd on complete lists, by phase (diagnostic build)
    export GSR_LIB_PATH=/tmp/gsr_timing.so; GSR_TIMING=1 GSR_DEFS="-DGSR_TIMING_ORDER $TAILDEFS" python gs_localization_amd/build.py > $o/tail_build.log 2>&1
    for sc in ${@:-s_1m_640}; do echo "== $sc"; SCENE=$sc LOOP_PLAIN=1 python tools/dbg/tail_rows.py; done > $o/tail.log 2>&1 ;;
  timing_spec)   # phase clocks of the speculative loop (diagnostic build)
    export GSR_LIB_PATH=/tmp/gsr_timing.so; GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
    for sc in ${@:-s_1m_640}; do echo "== $sc speculative"; SCENE=$sc python tools/phase_timing.py 2>/dev/null | grep -v amdgpu; done > $o/timing_spec.log ;;
  fuzz)          # randomised campaigns: per-pixel parity against the oracle, speculation bit for bit under the deterministic option
    CASES=${1:-300} SEED=${2:-4001} timeout 2400 python tools/fuzz_parity.py 2>&1 | grep -v amdgpu | tail -6 > $o/fuzz_parity.log
    CASES=${1:-300} SEED=${2:-4002} timeout 2400 python tools/fuzz_speculation.py 2>&1 | grep -v amdgpu | tail -4 > $o/fuzz_spec.log
    BIG=1 CASES=60 SEED=${2:-4003} timeout 2400 python tools/fuzz_speculation.py 2>&1 | grep -v amdgpu | tail -4 > $o/fuzz_spec_big.log ;;
  *) echo "unknown step $step" ;;
esac
