#!/bin/bash
# profiles/builder_libs.sh WORKLOAD lib1.so lib2.so ...: builder_parity.py under each library
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
wl="$1"; shift
for l in "$@"; do
  echo "== $l"
  RT_AMD_LIB="$R/raytracer_amd/variants/$l" timeout -k 10 200 python3 "$R/profiles/builder_parity.py" "$wl" || exit 1
done
