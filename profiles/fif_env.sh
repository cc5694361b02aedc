#!/bin/bash
# profiles/fif_env.sh VAR "v1 v2 ..." "PARTS F" ...: frames_in_flight.py under values of one env knob
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
VAR="$1"; VALS="$2"; shift; shift
for v in $VALS; do for a in "$@"; do printf "%s=%s  " "$VAR" "$v"; env "$VAR=$v" timeout -k 10 120 python3 "$R/profiles/frames_in_flight.py" $a 2>/dev/null || { echo FAILED; exit 1; }; done; done
