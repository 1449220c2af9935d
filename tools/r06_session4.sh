#!/bin/bash
# round 6, GPU session 4g: se50 failure -- the lane kernel held to 2 wavefronts per SIMD (202 VGPRs, no vector-register spills)
cd ${GRAFT_REPO_ROOT:-.}
V=$PWD/star_amd/lib/variants
mkdir -p gpurun_out/r06s4
for spec in "lanew2 X=1" "lanew2 STARAMD_LANE_CLASS=31" "lanew2 X=1" "lanew2 STARAMD_LANE_CLASS=31" "lanew2 X=1" "lanew2 STARAMD_LANE_CLASS=31"; do
  v=${spec%% *}; envs=${spec#"$v"}
  env $envs STARAMD_ENGINE_LIB=$V/libstaramd_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference_outputs" > gpurun_out/r06s4/se50_$v.log 2>&1; echo "variant $v [$envs]: rc $? $(tail -1 gpurun_out/r06s4/se50_$v.log)"
done
