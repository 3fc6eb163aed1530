#!/bin/bash
# round 4, after the pair path kernel's rewrite: kernel stats + PMC of the bench command again (one IR at a time; the default lane choice, then pairs forced)
cd "${GRAFT_REPO_ROOT:-$PWD}"
bash tools/profile.sh r04c > gpurun_out/r04c_profile.log 2>&1
echo "profile r04c rc $?"; tail -5 gpurun_out/r04c_profile.log
bash tools/profile.sh r04cpairs RVB_PATH_LANES=2 > gpurun_out/r04cpairs_profile.log 2>&1
echo "profile r04cpairs rc $?"; tail -3 gpurun_out/r04cpairs_profile.log
rm -rf gpurun_out/r04c_stats gpurun_out/r04c_stats_default gpurun_out/r04c_pmc_* gpurun_out/r04cpairs_stats gpurun_out/r04cpairs_stats_default gpurun_out/r04cpairs_pmc_*
ls gpurun_out | grep r04c | head -50
