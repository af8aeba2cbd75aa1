#!/bin/bash
# Round-end evidence run on one MI355X: every stage of tools/gpu_stage.sh in one call; tools/collect_profiles.py turns
# gpurun_out/<stage>/ into profiles/<tag>_*.   gpurun --timeout 1800 -- 'bash tools/gpu_final.sh'
# (round 5: two calls -- STAGES="tests bench prof hetero warm5" and STAGES="sweep cfgtraffic fuzz" -- keep each under its timeout)
bash "$(dirname "$0")/gpu_stage.sh" ${STAGES:-tests bench prof hetero warm5}
