#!/bin/bash
# Round-end evidence run on one MI355X: every stage of tools/gpu_stage.sh in one call; tools/collect_profiles.py turns
# gpurun_out/<stage>/ into profiles/<tag>_*.   gpurun --timeout 1800 -- 'bash tools/gpu_final.sh'
bash "$(dirname "$0")/gpu_stage.sh" tests bench prof sweep adaptive counters cfgtraffic probes
