#!/bin/bash
# profiles/ktimes.sh [bench args...] : per-kernel average durations of the default bench loop (kernel trace only,
# one frame at a time), for a quick before / after of a kernel change.  Raw output goes to gpurun_out/ktimes/.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/ktimes"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pipelined --frames-in-flight 1 "$@" > "$OUT/log.txt" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    if float(r["Percentage"]) >= 0.3:
        print("%-46s calls %4s  avg %9.1f us  %6s %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
