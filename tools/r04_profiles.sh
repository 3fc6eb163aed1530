#!/bin/bash
# round 4: the committed profiles — kernel stats + PMC of the bench command (one IR at a time; pairs forced), path-kernel PMC on the C4 scene
cd "${GRAFT_REPO_ROOT:-$PWD}"
bash tools/profile.sh r04 > gpurun_out/r04_profile.log 2>&1
echo "profile r04 rc $?"; tail -5 gpurun_out/r04_profile.log
bash tools/profile.sh r04pairs RVB_PATH_LANES=2 > gpurun_out/r04pairs_profile.log 2>&1
echo "profile r04pairs rc $?"; tail -3 gpurun_out/r04pairs_profile.log
SWEEP_SCENE=atrium SWEEP_TRIANGLES=262000 SWEEP_REFLECTIONS=256 bash tools/path_pmc.sh r04c4 100000 2 4 > gpurun_out/r04c4_pmc.log 2>&1
echo "c4 pmc rc $?"; head -5 gpurun_out/r04c4_path_pmc.txt
for L in 1 2 4; do SWEEP_SCENE=atrium SWEEP_TRIANGLES=262000 SWEEP_REFLECTIONS=256 RVB_PATH_LANES=$L timeout -k 10 300 python tools/rays_sweep.py 100000 200000 400000 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r04_rays_sweep_c4_n1.txt
rm -rf gpurun_out/r04_stats gpurun_out/r04_stats_default gpurun_out/r04_pmc_* gpurun_out/r04pairs_stats gpurun_out/r04pairs_stats_default gpurun_out/r04pairs_pmc_* gpurun_out/r04c4_pp_*
ls gpurun_out | head -50
