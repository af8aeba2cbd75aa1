#!/usr/bin/env python3
"""Experiment builds: copies of the product library in which ONE kernel translation unit is rebuilt with extra -D flags and / or a
patched COPY of admm_kernel.hip.h (nothing in csrc/ is modified); run a tool against one with TINYMPC_AMD_LIB=<path>.
    python tools/build_variants.py [tag ...]        -> tinympc_amd/libtinympc_amd_<tag>.so
  prim0..prim4  (12,4,10): the x|u store plain / nontemporal (the default) / sc1 / sc0 sc1 / sc0 sc1 nt (TINYMPC_PRIM_STORE)
  ref0          (12,4,10): per-instance Xref|Uref records read with plain loads (TINYMPC_REF_LOAD=0; the default is nontemporal)
  socclk        (6,3,10): s_memtime phase clocks of the cone kernel's iteration (backward, forward, cone step, tail) in the four
                residual outputs (shader cycles summed over the iterations of a solve; s_memtime answers on the counter LDS reads
                use, so every clock read also waits for the prefetches in flight: phases come out longer than they run)
  idm           (6,3,10): the identity-mode stretch experiment, see profiles/r04_negative_results.md"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tinympc_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-but-set-variable -Wno-unused-variable".split()

SOCCLK = [
    ("                for (int it = iter0; it < P.max_iter; ++it) {\n",
     "                long long clkB = 0, clkF = 0, clkC = 0, clkT = 0;\n                for (int it = iter0; it < P.max_iter; ++it) {\n                    const long long t0 = clock64();\n"),
    ("                    double pmax = 0.0, dmax = 0.0;\n",
     "                    const long long t1 = clock64();\n                    double pmax = 0.0, dmax = 0.0;\n"),
    ("                    // ---- termination_condition, admm.cpp:310-328 (the box residuals: the cone slacks do not enter them)\n",
     "                    const long long t2 = clock64();\n"),
    ("                    iter += 1;                                                      // :394\n",
     "                    const long long t3 = clock64();\n                    iter += 1;\n"),
    ("                    if (conv) { solved = 1; break; }                                // :431-441 (returns before v = vnew)\n#pragma unroll\n                    for (int s = 0; s < N; ++s) VP[s] = VN[s];                      // :445-446\n                    vp_touched = true;\n",
     "                    if (conv) { solved = 1; break; }\n#pragma unroll\n                    for (int s = 0; s < N; ++s) VP[s] = VN[s];\n                    vp_touched = true;\n"
     "                    const long long t4 = clock64();\n                    clkB += t1 - t0; clkF += t2 - t1; clkC += t3 - t2; clkT += t4 - t3;\n"),
    ("                acc_iter += (unsigned)(iter - iter0);\n",
     "                acc_iter += (unsigned)(iter - iter0);\n                rp = (double)clkB; rd = (double)clkF; rc_ = (double)clkC; rt_ = (double)clkT;\n"),
    ("            double rp = 0.0, rd = 0.0;\n", "            double rp = 0.0, rd = 0.0, rc_ = 0.0, rt_ = 0.0;\n"),
    ("            const double ps = grp_maxw<RL>(is_state ? rp : 0.0), pi = grp_maxw<RL>(is_input ? rp : 0.0);\n            const double ds = grp_maxw<RL>(is_state ? rd : 0.0), di = grp_maxw<RL>(is_input ? rd : 0.0);\n",
     "            const double ps = rp, pi = rd, ds = rc_, di = rt_;\n"),
]
# socclk + a count of the passes that took their exact path, added to the "tail" clock in units of 1e6
SOCCLK_CNT = SOCCLK + [("                                gcz[p] = false;\n                                any_exact = true;\n",
                        "                                gcz[p] = false;\n                                any_exact = true;\n                                clkT += 1000000;\n")]
# which lanes fail the all-inside test (the ballot of the last pass, in the "tail" slot; the "cone" slot: u0 and q of lane 0)
SOCCLK_MASK = SOCCLK + [("                            if (soc_all_inside(s0, s1, s2, item_mu[p])) {\n",
                         "                            { const double u0_ = s2 * (double)item_mu[p]; const double q_ = s0 * s0 + s1 * s1;\n"
                         "                              const bool si_ = (u0_ > 1e-30) & (u0_ < 1e300) & (q_ <= (u0_ * u0_) * (1.0 - 0x1p-20)) & (q_ < 1e70);\n"
                         "                              dbgmask = (long long)__builtin_amdgcn_ballot_w64(!si_); dbgu0 = u0_; dbgq = q_; }\n"
                         "                            if (soc_all_inside(s0, s1, s2, item_mu[p])) {\n"),
                        ("                long long clkB = 0, clkF = 0, clkC = 0, clkT = 0;\n", "                long long clkB = 0, clkF = 0, clkC = 0, clkT = 0, dbgmask = 0; double dbgu0 = 0, dbgq = 0;\n"),
                        ("rp = (double)clkB; rd = (double)clkF; rc_ = (double)clkC; rt_ = (double)clkT;", "rp = dbgu0 * (double)(iter - iter0); rd = dbgq * (double)(iter - iter0); rc_ = (double)clkC; rt_ = (double)dbgmask * (double)(iter - iter0);")]

# the cone kernel's two iteration modes: shader cycles and iteration counts of the general / the identity-mode iterations of a solve
# (in the four residual outputs: cycles general, cycles identity, iterations general, iterations identity)
SOCMODE = [
    ("                    if (!(SOC && redo)) {\n",
     "                    const long long tg0 = clock64();\n                    if (!(SOC && redo)) {\n"),
    ("                    vp_touched = true;\n                    ++it;\n",
     "                    vp_touched = true;\n                    ++it;\n                    clkG += clock64() - tg0; nG += 1;\n"),
    ("                            while (it < P.max_iter) {\n                                backward_sweep(bool_c<true>{});\n",
     "                            while (it < P.max_iter) {\n                                const long long ti0 = clock64();\n                                backward_sweep(bool_c<true>{});\n"),
    ("                                for (int s = 0; s < N; ++s) VP[s] = VN[s];\n                                ++it;\n",
     "                                for (int s = 0; s < N; ++s) VP[s] = VN[s];\n                                ++it;\n                                clkI += clock64() - ti0; nI += 1;\n"),
    ("                int it = iter0;\n", "                int it = iter0;\n                long long clkG = 0, clkI = 0, nG = 0, nI = 0;\n"),
    ("                                    redo = true;\n", "                                    redo = true; nG += 1000;\n"),
    ("                acc_iter += (unsigned)(iter - iter0);\n",
     "                acc_iter += (unsigned)(iter - iter0);\n                rp = (double)clkG; rd = (double)clkI; rc_ = (double)nG; rt_ = (double)nI;\n"),
    ("            double rp = 0.0, rd = 0.0;\n", "            double rp = 0.0, rd = 0.0, rc_ = 0.0, rt_ = 0.0;\n"),
    ("            const double ps = grp_maxw<RL>(is_state ? rp : 0.0), pi = grp_maxw<RL>(is_input ? rp : 0.0);\n            const double ds = grp_maxw<RL>(is_state ? rd : 0.0), di = grp_maxw<RL>(is_input ? rd : 0.0);\n",
     "            const double ps = rp, pi = rd, ds = rc_, di = rt_;\n"),
]
VARIANTS = {
    "prim0": ("u_12_4_10", ["-DTINYMPC_PRIM_STORE=0"], []),
    "ref0": ("u_12_4_10", ["-DTINYMPC_REF_LOAD=0"], []),
    "prim1": ("u_12_4_10", ["-DTINYMPC_PRIM_STORE=1"], []),
    "prim2": ("u_12_4_10", ["-DTINYMPC_PRIM_STORE=2"], []),
    "prim3": ("u_12_4_10", ["-DTINYMPC_PRIM_STORE=3"], []),
    "prim4": ("u_12_4_10", ["-DTINYMPC_PRIM_STORE=4"], []),
    "socclk": ("u_6_3_10", [], SOCCLK),
    "socclkcnt": ("u_6_3_10", [], SOCCLK_CNT),
    "socmask": ("u_6_3_10", [], SOCCLK_MASK),
    # the abandoned identity-mode stretch of the cone kernel (tools/experiments/idm_mode.diff, profiles/r04_negative_results.md):
    # on / compiled out / with its mode counters
    "idm": ("u_6_3_10", [], [], "idm_mode.diff"),
    "idm0": ("u_6_3_10", ["-DTINYMPC_SOC_IDM=0"], [], "idm_mode.diff"),
    "idm_socmode": ("u_6_3_10", [], SOCMODE, "idm_mode.diff"),
    "socpd3": ("u_6_3_10", ["-DTINYMPC_SOC_PD=3"], []),     # LDS prefetch distance of the cone slack cells (default 2): measured equal
    # (20,8,50) tile forms that stream v|z to its record (LM bit 4): the per-iteration store removed (timing only, WRONG results) /
    # issued nontemporal (results unchanged)
    "vpg_nostore": ("u_20_8_50", [], [("(l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];\n                        } else dmax", "} else dmax"),
                                      ("(l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];\n                            } else dmax", "} else dmax")]),
    "vpg_nt": ("u_20_8_50", [], [("(l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];\n                        } else dmax", "__builtin_nontemporal_store(VN[l], (l == 0 ? vpp0 : vpp) + l * NZ);\n                        } else dmax"),
                                 ("(l == 0 ? vpp0 : vpp)[l * NZ] = VN[l];\n                            } else dmax", "__builtin_nontemporal_store(VN[l], (l == 0 ? vpp0 : vpp) + l * NZ);\n                            } else dmax")]),
    # round 5, the warm regime beyond the Infinity Cache (batch 262 144): the warm-start record stores / loads issued nontemporal too
    # (results unchanged) -- beyond L3 nothing a launch stores is still cached when the next launch wants it
    "ntst": ("u_12_4_10", [], [("                    if (P.store_mask & 2) P.slack[off] = VN[s];\n                    if (P.store_mask & 4) P.dual[off] = G[s];\n                    if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = VP[s];",
                                "                    if (P.store_mask & 2) __builtin_nontemporal_store(VN[s], P.slack + off);\n                    if (P.store_mask & 4) __builtin_nontemporal_store(G[s], P.dual + off);\n                    if ((P.store_mask & 8) && vp_touched) __builtin_nontemporal_store(VP[s], P.slack_prev + off);")]),
    "ntld": ("u_12_4_10", [], [("                VN[s] = warm ? P.slack[off] : 0.0;\n                G[s] = warm ? P.dual[off] : 0.0;\n                VP[s] = warm ? P.slack_prev[off] : 0.0;",
                                "                VN[s] = warm ? __builtin_nontemporal_load(P.slack + off) : 0.0;\n                G[s] = warm ? __builtin_nontemporal_load(P.dual + off) : 0.0;\n                VP[s] = warm ? __builtin_nontemporal_load(P.slack_prev + off) : 0.0;")]),
    "ntboth": ("u_12_4_10", [], [("                    if (P.store_mask & 2) P.slack[off] = VN[s];\n                    if (P.store_mask & 4) P.dual[off] = G[s];\n                    if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = VP[s];",
                                  "                    if (P.store_mask & 2) __builtin_nontemporal_store(VN[s], P.slack + off);\n                    if (P.store_mask & 4) __builtin_nontemporal_store(G[s], P.dual + off);\n                    if ((P.store_mask & 8) && vp_touched) __builtin_nontemporal_store(VP[s], P.slack_prev + off);"),
                                 ("                VN[s] = warm ? P.slack[off] : 0.0;\n                G[s] = warm ? P.dual[off] : 0.0;\n                VP[s] = warm ? P.slack_prev[off] : 0.0;",
                                  "                VN[s] = warm ? __builtin_nontemporal_load(P.slack + off) : 0.0;\n                G[s] = warm ? __builtin_nontemporal_load(P.dual + off) : 0.0;\n                VP[s] = warm ? __builtin_nontemporal_load(P.slack_prev + off) : 0.0;")]),
    # round 6, the PREFETCH form's own costs (timing only: the accumulated iteration counters come out wrong / unchanged results)
    "pf_noaccum": ("u_12_4_10", [], [("                    if constexpr (PF) {                      // (no value comes back: a persistent wave must not wait for one here)\n                        __hip_atomic_fetch_add(&P.accum[b].x, acc_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n                        __hip_atomic_fetch_add(&P.accum[b].y, acc_solved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n                    } else {",
                                      "                    if constexpr (PF) {\n                    } else {")]),
    "pf_keep0": ("u_12_4_10", [], [("            else if ((P.store_mask & 6) == 6) asm volatile(\"s_waitcnt vmcnt(%0)\" :: \"n\"(2 * N <= 63 ? 2 * N : 63) : \"memory\");\n", "")]),
    "pf_rmwaccum": ("u_12_4_10", [], [("                    if constexpr (PF) {                      // (no value comes back: a persistent wave must not wait for one here)", "                    if constexpr (false) {")]),
    # TLB reach?  a wave's tiles CONTIGUOUS (wave w: tiles w * pf_static ...) instead of grid-strided: valid with prefetch_static = 100 and
    # tiles = grid * pf_static exactly (65 536 / 2 048)
    "pf_blocked": ("u_12_4_10", [], [("        if ((int)blockIdx.x < ntiles) {\n            pf_issue((int)blockIdx.x);", "        if ((int)blockIdx.x < ntiles) {\n            pf_issue((int)blockIdx.x * P.pf_static);"),
                                      ("    for (int tile = blockIdx.x; tile < ntiles;\n         tile = PF ? pf_next", "    for (int tile = PF ? blockIdx.x * P.pf_static : blockIdx.x; tile < ntiles;\n         tile = PF ? pf_next"),
                                      ("            if (pf_i + 1 < P.pf_static) pf_next = tile + (int)gridDim.x;", "            if (pf_i + 1 < P.pf_static) pf_next = tile + 1;")]),
    # round 6: what the PREFETCH form's write-back does to the HBM regime (results unchanged in all of them)
    #   pf_p0: x|u stored with plain stores in the PF form (the plain form keeps its nontemporal ones); + _nt: the LDS-DMA pieces issued
    #   nontemporal (aux = 2); + _am: write-back array by array instead of slot by slot; + _stag: the waves of a CU start 0 ... 7 x 2 us apart
    "pf_p0": ("u_12_4_10", [], [("                        else store_primal(P.prim + off, X[s]);", "                        else if constexpr (PF) P.prim[off] = X[s];\n                        else store_primal(P.prim + off, X[s]);")]),
    "pf_p0_nt": ("u_12_4_10", [], [("                        else store_primal(P.prim + off, X[s]);", "                        else if constexpr (PF) P.prim[off] = X[s];\n                        else store_primal(P.prim + off, X[s]);"),
                                   ("(pf_lptr)(sPF + at + q * 128), 16, 0, 0);", "(pf_lptr)(sPF + at + q * 128), 16, 0, 2);")]),
    "pf_p0_stag": ("u_12_4_10", [], [("                        else store_primal(P.prim + off, X[s]);", "                        else if constexpr (PF) P.prim[off] = X[s];\n                        else store_primal(P.prim + off, X[s]);"),
                                     ("    if constexpr (PF) {\n        if ((int)blockIdx.x < ntiles) {\n            pf_issue(", "    if constexpr (PF) {\n        for (unsigned w = 0; w < ((blockIdx.x >> 8) & 7u) * 32u; ++w) __builtin_amdgcn_s_sleep(127);\n        if ((int)blockIdx.x < ntiles) {\n            pf_issue(")]),
    "pf_p0_am": ("u_12_4_10", [], [("                        else store_primal(P.prim + off, X[s]);", "                        else if constexpr (PF) { }\n                        else store_primal(P.prim + off, X[s]);"),
                                   ("                    if (P.store_mask & 2) P.slack[off] = VN[s];\n                    if (P.store_mask & 4) P.dual[off] = G[s];\n                    if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = VP[s];",
                                    "                    if constexpr (!PF) {\n                    if (P.store_mask & 2) P.slack[off] = VN[s];\n                    if (P.store_mask & 4) P.dual[off] = G[s];\n                    if ((P.store_mask & 8) && vp_touched) P.slack_prev[off] = VP[s];\n                    }"),
                                   ("            // ---- write back (coalesced) -------------------------------------------------------\n",
                                    "            if constexpr (PF) {\n#pragma unroll\n                for (int s = 0; s < N; ++s) { const bool valid = is_state || (is_input && s >= 1); if (valid && (P.store_mask & 2)) P.slack[lbase + s * NZ] = VN[s]; }\n#pragma unroll\n                for (int s = 0; s < N; ++s) { const bool valid = is_state || (is_input && s >= 1); if (valid && (P.store_mask & 4)) P.dual[lbase + s * NZ] = G[s]; }\n#pragma unroll\n                for (int s = 0; s < N; ++s) { const bool valid = is_state || (is_input && s >= 1); if (valid && (P.store_mask & 8) && vp_touched) P.slack_prev[lbase + s * NZ] = VP[s]; }\n#pragma unroll\n                for (int s = 0; s < N; ++s) { const bool valid = is_state || (is_input && s >= 1); if (valid && (P.store_mask & 1) && acc_iter > 0) P.prim[lbase + s * NZ] = X[s]; }\n            }\n            // ---- write back (coalesced) -------------------------------------------------------\n")]),
    "fls0": ("u_12_4_10", ["-DTINYMPC_FULL_LINE_STORES=0"], []),       # record stores as whole knot segments: never / in every box variant (the default: PF only)
    "fls2": ("u_12_4_10", ["-DTINYMPC_FULL_LINE_STORES=2"], []),
    # timing-only ablations of the cone kernel (results are WRONG by construction)
    "abl_fwd_nogc": ("u_6_3_10", [], [("gr[(i + 2) % 3] = sC[cw + (i + 2) * SLOT_D + PL_GC];", "gr[(i + 2) % 3] = 0.0;"),
                                     ("                        gr[0] = sC[cw + PL_GC];\n                        if constexpr (N >= 2) gr[1] = sC[cw + SLOT_D + PL_GC];\n",
                                      "                        gr[0] = 0.0; gr[1] = 0.0;\n")]),
    "abl_bwd_now": ("u_6_3_10", [], [("if (i >= 2) wr[(i - 2) % 3] = sC[cw + (i - 2) * SLOT_D];", "if (i >= 2) wr[(i - 2) % 3] = 0.0;")]),
    "abl_nopass": ("u_6_3_10", [], [("                            if (p > 0 && p >= soc_passes) break;            // wave-uniform\n", "                            break;\n")]),
}


def build(tag):
    unit, defs, patches, *diff = VARIANTS[tag]
    tmp = "/tmp/variant_" + tag
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/_gen")
    for f in os.listdir(SRC):
        if f.endswith((".h", ".hpp", ".hip")):
            shutil.copy(os.path.join(SRC, f), tmp)
    shutil.copy(os.path.join(SRC, "_gen", unit + ".hip"), tmp + "/_gen")
    for d in diff:                                      # a source diff kept under tools/experiments/ goes on first
        subprocess.check_call(["patch", "-s", tmp + "/admm_kernel.hip.h", os.path.join(ROOT, "tools", "experiments", d)])
    applied = [False] * len(patches)
    for hdr in ("admm_kernel.hip.h", "tile_kernel.hip.h"):
        p = tmp + "/" + hdr
        s = open(p).read()
        for i, (a, b) in enumerate(patches):             # in order: a later patch may anchor on what an earlier one put in
            if a in s:
                s = s.replace(a, b)
                applied[i] = True
        open(p, "w").write(s)
    assert all(applied), "a patch no longer applies: %r" % [patches[i][0][:60] for i, ok in enumerate(applied) if not ok]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-c", tmp + "/_gen/" + unit + ".hip", "-o", tmp + "/k.o"])
    objs = [os.path.join(SRC, "_gen", f) for f in os.listdir(SRC + "/_gen") if f.endswith(".o") and f != unit + ".o" and "_chk" not in f]
    out = os.path.join(ROOT, "tinympc_amd", "libtinympc_amd_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o", out, *objs, tmp + "/k.o", "-ldl"])
    print("built", out)


if __name__ == "__main__":
    for t in (sys.argv[1:] or VARIANTS):
        build(t)
