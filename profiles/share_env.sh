#!/bin/bash
# profiles/share_env.sh PARTS VAR v1 v2 ...: one rank's share of the frame (part 0 of PARTS) under values of one env knob
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
P="$1"; VAR="$2"; shift; shift
for v in "$@"; do
  printf "%s=%s  " "$VAR" "$v"
  env "$VAR=$v" timeout -k 10 100 python3 "$R/profiles/share_frames.py" "$P" 20 2>/dev/null || { echo FAILED; exit 1; }
done
