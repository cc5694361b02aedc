#!/bin/sh
# Compile the UNMODIFIED reference tracer (main.cpp + objloader.cpp, where they lie)
# together with oracle/ref_harness.cpp into oracle/_ref/.  Two builds:
#   libref_O0.so  -O0, the Xcode Debug equivalent: canonical 8-bit frame of setup()
#   libref.so     -O2 -ffp-contract=off: same float results, faster, for scene sweeps
set -e
REF_DIR="${1:-/root/reference/Raytracer}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
mkdir -p "$OUT"
COMMON="-std=gnu++11 -fPIC -shared -w -ffp-contract=off -fno-fast-math -I$REF_DIR"
g++ -O0 $COMMON -o "$OUT/libref_O0.so" "$HERE/ref_harness.cpp" "$REF_DIR/objloader.cpp"
g++ -O2 $COMMON -o "$OUT/libref.so"    "$HERE/ref_harness.cpp" "$REF_DIR/objloader.cpp"
echo "built $OUT/libref_O0.so $OUT/libref.so"
