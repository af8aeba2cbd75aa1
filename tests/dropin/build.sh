#!/bin/bash
# Drop-in boundary check: compile the REFERENCE's own example mains
# (by path, nothing copied) against the REFERENCE's headers, but link them against
# libtinympc_amd.so instead of libtinympcstatic.a.  Needs /root/reference, so it only runs in the
# build container; the binaries land in tests/dropin/_build/ (git-ignored, travel to the GPU box).
#   build.sh           -> tests/dropin/_build/<example>          (linked against OUR library)
#   build.sh --golden  -> also builds each example against the real reference sources and records
#                         its stdout in tests/golden/stdout_<example>.txt
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
REF=${TINYMPC_REFERENCE:-/root/reference}
[ -d "$REF/examples" ] || { echo "no $REF: keeping prebuilt binaries"; exit 0; }
OUT=$HERE/_build; mkdir -p "$OUT"
INC="-I$REF/src -I$REF/include/Eigen -I$REF/include -I$REF/examples"
# default alignment flags on purpose: Eigen must use plain malloc/free (SURVEY.md 8(b) ownership row)
CXXFLAGS="-O2 -DNDEBUG -std=c++17 -w"
EXAMPLES="cartpole_example quadrotor_hovering quadrotor_tracking rocket_landing_mpc quadrotor_linear_constraints quadrotor_tv_linear_constraints"
for ex in $EXAMPLES; do
  g++ $CXXFLAGS $INC -o "$OUT/$ex" "$REF/examples/$ex.cpp" -L"$ROOT/tinympc_amd" -ltinympc_amd \
      -Wl,-rpath,'$ORIGIN/../../../tinympc_amd' &
done
# the reference's two code-generation examples (examples/codegen_random.cpp, codegen_cartpole.cpp): tiny_codegen of a
# TinySolver built with real Eigen types; what they generate is checked by tests/test_gpu_dropin.py
for ex in codegen_random codegen_cartpole; do
  g++ $CXXFLAGS $INC -o "$OUT/$ex" "$REF/examples/$ex.cpp" -L"$ROOT/tinympc_amd" -ltinympc_amd \
      -Wl,-rpath,'$ORIGIN/../../../tinympc_amd' &
done
# our own caller of the exported phase functions (admm.hpp:12-34), real Eigen types across the boundary
g++ $CXXFLAGS $INC -o "$OUT/phase_driver" "$HERE/phase_driver.cpp" -L"$ROOT/tinympc_amd" -ltinympc_amd \
    -Wl,-rpath,'$ORIGIN/../../../tinympc_amd' &
# our own caller of the adaptive-rho path (settings->adaptive_rho = 1 + tiny_initialize_sensitivity_matrices)
g++ $CXXFLAGS $INC -o "$OUT/adaptive_driver" "$HERE/adaptive_driver.cpp" -L"$ROOT/tinympc_amd" -ltinympc_amd \
    -Wl,-rpath,'$ORIGIN/../../../tinympc_amd' &
# our own caller of the C++-mangled helpers of rho_benchmark.hpp (rho_api.hip)
g++ $CXXFLAGS $INC -o "$OUT/rho_driver" "$HERE/rho_driver.cpp" -L"$ROOT/tinympc_amd" -ltinympc_amd \
    -Wl,-rpath,'$ORIGIN/../../../tinympc_amd' &
wait
if [ "$1" = "--golden" ]; then
  g++ $CXXFLAGS $INC -o "$OUT/ref_rho_driver" "$HERE/rho_driver.cpp" "$REF/src/tinympc/admm.cpp" \
      "$REF/src/tinympc/tiny_api.cpp" "$REF/src/tinympc/rho_benchmark.cpp"
  (cd "$OUT" && ./ref_rho_driver > "$ROOT/tests/golden/stdout_rho_driver.txt"); rm -f "$OUT/ref_rho_driver"
  g++ $CXXFLAGS $INC -o "$OUT/ref_phase_driver" "$HERE/phase_driver.cpp" "$REF/src/tinympc/admm.cpp" \
      "$REF/src/tinympc/tiny_api.cpp" "$REF/src/tinympc/rho_benchmark.cpp"
  (cd "$OUT" && ./ref_phase_driver > "$ROOT/tests/golden/stdout_phase_driver.txt"); rm -f "$OUT/ref_phase_driver"
  g++ $CXXFLAGS $INC -o "$OUT/ref_adaptive_driver" "$HERE/adaptive_driver.cpp" "$REF/src/tinympc/admm.cpp" \
      "$REF/src/tinympc/tiny_api.cpp" "$REF/src/tinympc/rho_benchmark.cpp"
  (cd "$OUT" && ./ref_adaptive_driver > "$ROOT/tests/golden/stdout_adaptive_driver.txt"); rm -f "$OUT/ref_adaptive_driver"
  for ex in $EXAMPLES; do
    g++ $CXXFLAGS $INC -o "$OUT/ref_$ex" "$REF/examples/$ex.cpp" "$REF/src/tinympc/admm.cpp" \
        "$REF/src/tinympc/tiny_api.cpp" "$REF/src/tinympc/rho_benchmark.cpp" &
  done
  wait
  for ex in $EXAMPLES; do (cd "$OUT" && ./ref_$ex > "$ROOT/tests/golden/stdout_$ex.txt"); rm -f "$OUT/ref_$ex"; done
fi
ls -la "$OUT"
