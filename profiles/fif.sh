#!/bin/bash
# profiles/fif.sh "PARTS F" ... : frames_in_flight.py for each pair
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for a in "$@"; do timeout -k 10 120 python3 "$R/profiles/frames_in_flight.py" $a 2>/dev/null || { echo "$a FAILED"; exit 1; }; done
