#!/usr/bin/env python3
"""Generates the kernel translation units of libtinympc_amd.so from kernel_dims.txt and tile_dims.txt (called by the Makefile):

  _gen/u_<nx>_<nu>_<N>.hip   ONE unit per (nx, nu, N) shape: its one-row instantiations (kernel_dims.txt: the FULL variant set, or
                             the LEAN one -- the box kernel in its two bound forms; every other variant of a lean shape is
                             instantiated at run time, jit.hip) and every tile-kernel form tile_dims.txt lists for it
  _gen/registry.inc          the tables batch_api.hip looks shapes up in (tile entries in tile_dims.txt order: the FIRST entry of a
                             shape is the one launched)
  _gen/units.mk              UOBJ := the objects to link

A file is only rewritten when its content changes, so make's mtime rule recompiles what an edit really touched."""
import collections
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "_gen")


def rows(path, nmin):
    for line in open(os.path.join(HERE, path)):
        f = line.split("#")[0].split()
        if len(f) >= nmin:
            yield f


def write_if_changed(path, text):
    if os.path.exists(path) and open(path).read() == text:
        return
    with open(path, "w") as f:
        f.write(text)


def main():
    os.makedirs(GEN, exist_ok=True)
    one_row = collections.OrderedDict()
    for f in rows("kernel_dims.txt", 3):
        one_row[tuple(map(int, f[:3]))] = (len(f) > 3 and f[3] == "lean")
    tiles = collections.OrderedDict()
    tile_order = []
    for f in rows("tile_dims.txt", 5):
        nx, nu, N, W, R = map(int, f[:5])
        lm = int(f[5]) if len(f) > 5 else 99
        tiles.setdefault((nx, nu, N), []).append((W, R, lm))
        tile_order.append((nx, nu, N, W, R, lm))
    shapes = list(one_row) + [s for s in tiles if s not in one_row]
    for (nx, nu, N) in shapes:
        forms = tiles.get((nx, nu, N), [])
        t = []
        if (nx, nu, N) in one_row or any(W == 1 for W, _, _ in forms):      # the fused sweep-step blocks are spelled out for ONE (nx, nu) pair per unit
            t += ["#define TINYMPC_FUSED_NX %d" % nx, "#define TINYMPC_FUSED_NU %d" % nu]
        t.append('#include "../kernel_entry.hpp"')
        if forms:
            t.append('#include "../tile_kernel.hip.h"')
        t.append("namespace tinympc_amd {")
        if (nx, nu, N) in one_row:
            macro = "KERNELS_LEAN" if one_row[(nx, nu, N)] else "KERNELS_FOR"
            t.append("extern const KernelEntry kentry_%d_%d_%d = %s(%d, %d, %d);" % (nx, nu, N, macro, nx, nu, N))
        for W, R, lm in forms:
            a = "%d, %d, %d, %d, %d, %d" % (nx, nu, N, W, R, lm)
            t.append("extern const TileEntry tentry_%d_%d_%d_%d_%d_%d = { %s, tile_kernel_or_null<%s, false>(), tile_kernel_or_null<%s, true>(), "
                     "tile_kernel_or_null<%s, false, true>(), tile_kernel_or_null<%s, true, true>() };" % (nx, nu, N, W, R, lm, a, a, a, a, a))
        t.append("}  // namespace tinympc_amd")
        write_if_changed(os.path.join(GEN, "u_%d_%d_%d.hip" % (nx, nu, N)), "\n".join(t) + "\n")
    reg = ["extern const KernelEntry kentry_%d_%d_%d;" % s for s in one_row]
    reg += ["static const KernelEntry* const g_kernels[] = {"] + ["    &kentry_%d_%d_%d," % s for s in one_row] + ["};"]
    reg += ["extern const TileEntry tentry_%d_%d_%d_%d_%d_%d;" % e for e in tile_order]
    reg += ["static const TileEntry* const g_tiles[] = {"] + ["    &tentry_%d_%d_%d_%d_%d_%d," % e for e in tile_order] + ["};"]
    write_if_changed(os.path.join(GEN, "registry.inc"), "\n".join(reg) + "\n")
    with open(os.path.join(GEN, "units.mk"), "w") as f:          # (always: its mtime tells make that this generator has run)
        f.write("UOBJ := " + " ".join("_gen/u_%d_%d_%d.o" % s for s in shapes) + "\n")
    # units of shapes that left the lists
    keep = {"u_%d_%d_%d" % s for s in shapes}
    for fn in os.listdir(GEN):
        base, ext = os.path.splitext(fn)
        if ext in (".hip", ".o") and (base.startswith(("k_", "t_")) or (base.startswith("u_") and base not in keep)):
            os.remove(os.path.join(GEN, fn))


if __name__ == "__main__":
    main()
