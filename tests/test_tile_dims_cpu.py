"""CPU: invariants of tinympc_amd/csrc/tile_dims.txt (the compiled-in forms of the tile kernel; the FIRST entry of a shape is the one
launched, the LAST one's R serves the run-time instantiated cone / half-space variants)."""
import collections
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QX, DN, REGEN, VP, VPG, QXR = 1, 2, 4, 8, 16, 32


def entries():
    out = collections.OrderedDict()
    for line in open(os.path.join(ROOT, "tinympc_amd", "csrc", "tile_dims.txt")):
        f = line.split("#")[0].split()
        if len(f) >= 5:
            nx, nu, N, W, R = map(int, f[:5])
            out.setdefault((nx, nu, N), []).append((W, R, int(f[5]) if len(f) > 5 else 99))
    return out


def test_every_entry_is_a_legal_form():
    for (nx, nu, N), forms in entries().items():
        nz = nx + nu
        assert len(set(forms)) == len(forms), ("duplicate entry", nx, nu, N)
        for W, R, lm in forms:
            assert W in (0, 1, 2) and R in (1, 2, 4) and max(W, 1) * R <= 4 and N % R == 0 and N // R >= 2, (nx, nu, N, W, R)
            assert nz <= (8 if W == 0 else 16 * W), ("the knot vector does not fit the rows", nx, nu, N, W)
            if W == 2:
                assert nz > 16, ("a shape that fits one row must not take two", nx, nu, N)
            if lm != 99:
                assert 0 <= lm < 64
                assert not (lm & VP and lm & VPG), "v|z: LDS or its record, not both"
                assert not (lm & QX and lm & QXR), "QX: LDS or the reference record, not both"
                assert not (lm & QXR) or lm & VPG, "QXR rides on the VPG record pointers"
                # a form that regenerates the trajectory needs no x|u kept; one that streams v|z keeps nothing a cone variant could use
                assert not (lm & (VPG | QXR)) or lm & REGEN or N <= 10, (nx, nu, N, lm)


def one_row_shapes():
    out = set()
    for line in open(os.path.join(ROOT, "tinympc_amd", "csrc", "kernel_dims.txt")):
        f = line.split("#")[0].split()
        if len(f) >= 3:
            out.add(tuple(map(int, f[:3])))
    return out


def test_the_last_entry_can_serve_the_runtime_instantiated_variants():
    """Cone / half-space variants of a tile-only shape keep ALL their arrays in registers (9 L-long arrays with a cone, L = N / R) and
    inherit R from the shape's LAST entry (batch_dispatch.hip: variant_tile_r): that must be the most register-frugal split the shape has,
    and never a half-row form (box constraints only)."""
    regs = one_row_shapes()
    for (nx, nu, N), forms in entries().items():
        if (nx, nu, N) in regs:
            continue                                              # (their variants run on the one-row kernel)
        W, R, lm = forms[-1]
        assert W != 0, ("half rows serve box constraints only; the last entry must not be one", nx, nu, N)
        assert R == max(r for _, r, _ in forms), (nx, nu, N, forms)
        assert not (lm != 99 and lm & (VPG | QXR)), ("the last entry is the reference form for experiments too", nx, nu, N)


def test_sweep_cells_of_config5_are_all_served():
    served = set(entries()) | one_row_shapes()
    for nx in (4, 8, 12, 20):
        for nu in (2, 4, 8):
            for N in (10, 30, 50):
                assert (nx, nu, N) in served, (nx, nu, N)
