# which box is this?  partition modes, clocks, serial (the warm-regime numbers of the PREFETCH form differ from box to box)
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -8
rocm-smi --showserial --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -24
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Uuid" | head -12
