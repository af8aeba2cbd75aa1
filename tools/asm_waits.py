#!/usr/bin/env python3
"""tools/asm_waits.py <file.s> <substring of the mangled kernel name>: the VMEM side of a kernel in program order -- every vmcnt wait,
register-returning load, LDS-DMA piece, store, atomic and scratch access with its line number (loop headers marked).  What the
PREFETCH form must NOT have: a vmcnt wait the compiler put between the pieces of the next tile and this tile's iterations."""
import re
import sys
txt = open(sys.argv[1]).read()
for name in [m.group(1) for m in re.finditer(r"^(_Z\S*?):", txt, re.M) if sys.argv[2] in m.group(1)]:
    i = txt.index(name + ":")
    body = txt[i:txt.index(".Lfunc_end", i)].split("\n")
    print(name)
    last = None
    run = 0
    for k, l in enumerate(body):
        t = l.strip()
        kind = None
        if re.match(r"s_waitcnt.*vmcnt", t): kind = "WAIT  " + t
        elif t.startswith("global_load_lds"): kind = "dma"
        elif t.startswith("global_load") or t.startswith("flat_load"): kind = "load"
        elif t.startswith("global_store"): kind = "store"
        elif "_atomic" in t: kind = "ATOMIC " + t[:60]
        elif t.startswith("scratch_"): kind = "SCRATCH " + t[:50]
        elif "Loop Header" in t: kind = "---- " + t[:60]
        elif t.startswith("v_fmac_f64_dpp"): kind = "dppfma"
        if kind is None:
            continue
        if kind == last and kind in ("dma", "load", "store", "dppfma"):
            run += 1
            continue
        if last in ("dma", "load", "store", "dppfma") and run:
            print("        ... x%d" % (run + 1))
        run = 0
        print("%5d %s" % (k, kind))
        last = kind
