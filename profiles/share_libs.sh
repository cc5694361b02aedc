#!/bin/bash
# profiles/share_libs.sh PARTS lib1.so lib2.so ...: one rank's share of the frame (part 0 of PARTS) with several builds
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
P="$1"; shift
for lib in "$@"; do
  printf "%-40s " "$lib"
  RT_AMD_LIB="$R/$lib" timeout -k 10 100 python3 "$R/profiles/share_frames.py" "$P" 20 2>/dev/null || { echo FAILED; exit 1; }
done
