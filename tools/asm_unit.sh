#!/bin/bash
# tools/asm_unit.sh <nx> <nu> <N>: gfx950 assembly + register report of one kernel unit (tinympc_amd/csrc/_asm/), then the one-row
# instantiations' VGPR / scratch / LDS figures
set -e
cd "$(dirname "$0")/../tinympc_amd/csrc"
make -s _gen/units.mk
mkdir -p _asm && cd _asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -Wno-unused-but-set-variable -save-temps \
    -Rpass-analysis=kernel-resource-usage -c ../_gen/u_$1_$2_$3.hip -o u_$1_$2_$3.o 2> resource_usage_$1_$2_$3.txt || { grep -B2 -A8 "error" resource_usage_$1_$2_$3.txt | head -60; exit 1; }
python3 - "$1" "$2" "$3" <<'PY'
import re, sys
nx, nu, n = sys.argv[1:4]
txt = open("resource_usage_%s_%s_%s.txt" % (nx, nu, n)).read()
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split()[0]
    if "admm_solve_kernel" not in name:
        continue
    g = lambda k: re.search(k + r"[^:]*: (\d+)", blk).group(1)
    targs = re.search(r"admm_solve_kernelI(.*?)EEv", name).group(1)
    print("%-70s VGPR %s AGPR %s scratch %s occ %s LDS %s" % (targs, g("VGPRs"), g("AGPRs"), g("ScratchSize"), g("Occupancy"), g("LDS Size")))
PY
