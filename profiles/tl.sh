#!/bin/bash
# profiles/tl.sh lib.so [bench args...] : kernel timeline of one frame of a library build
# (rocprofv3 kernel trace only; output under gpurun_out/tl_<name>/)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd /tmp
lib="$1"; shift
name="$(basename "$lib" .so)"
OUT="$R/gpurun_out/tl_$name"
rm -rf "$OUT"; mkdir -p "$OUT"
export RT_AMD_LIB="$R/$lib"
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-graph "$@" > "$OUT/log.txt" 2>&1 || { echo "$lib: trace failed"; tail -3 "$OUT/log.txt"; exit 1; }
echo "== $lib $*"
python3 "$R/profiles/timeline.py" "$OUT" "${TL_BACK:-2}"
