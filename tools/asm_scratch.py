#!/usr/bin/env python3
"""tools/asm_scratch.py <file.s> <substring of the mangled kernel name>: where a kernel's scratch (spill) operations sit --
line numbers inside the function, which loop they belong to (by the innermost back-edge that encloses them)."""
import re
import sys
txt = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\S*?):", txt, re.M) if sys.argv[2] in m.group(1)]
for name in names:
    i = txt.index(name + ":")
    j = txt.index(".Lfunc_end", i)
    body = txt[i:j].split("\n")
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k))
    print(name, len(body), "lines")
    big = sorted(set(loops), key=lambda t: t[1] - t[0])
    for k, l in enumerate(body):
        if "scratch_" in l or "global_load_lds" in l:
            enc = [t for t in big if t[0] <= k <= t[1]]
            print("%5d %-72s loop %s" % (k, l.strip()[:72], enc[0] if enc else None))
    dpp = [k for k, l in enumerate(body) if "v_fmac_f64_dpp" in l]
    inner = [t for t in big if sum(1 for d in dpp if t[0] <= d <= t[1]) > 100]
    print("loops holding > 100 DPP FMAs:", inner[:4])
