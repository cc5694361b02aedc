#!/bin/bash
# profiles/pmc_once.sh <tag> "<counters>" [bench args]: one PMC pass, per-kernel sums
TAG="$1"; PMC="$2"; shift; shift
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc1" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/log.txt" 2>&1 || { echo "pass failed"; tail -3 "$OUT/log.txt"; exit 1; }
python3 "$R/profiles/summarize.py" "$OUT" | cut -c1-400
