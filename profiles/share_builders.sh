#!/bin/bash
# profiles/share_builders.sh PARTS... : ms per frame of part 0 of PARTS, for each hierarchy builder
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for parts in "$@"; do
  for b in 2 3; do
    ms=$(RT_SHARE_BUILDER=$b timeout -k 10 120 python3 "$R/profiles/share_frames.py" "$parts" 2>/dev/null) || { echo "builder $b parts $parts FAILED"; exit 1; }
    echo "parts $parts builder $b: $ms"
  done
done
