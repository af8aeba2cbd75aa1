#!/bin/bash
# Round-end evidence run on one MI355X: every stage of tools/gpu_stage.sh in one call; tools/collect_profiles.py turns
# gpurun_out/<stage>/ into profiles/<tag>_*.   gpurun --timeout 1800 -- 'bash tools/gpu_final.sh'
# (round 4, second evidence run: the stages whose subjects changed since the first one -- 8bf4b0b --, most important first)
CFG_ENTRIES="${CFG_ENTRIES:-config4 config4_state_cone config4_both_cones}" bash "$(dirname "$0")/gpu_stage.sh" ${STAGES:-tests bench prof cfgtraffic probes4 sweep}
