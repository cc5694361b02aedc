#!/bin/bash
# A/B several builds of librt_amd.so in one GPU session: profiles/ab.sh lib1.so lib2.so ...
# Stops at the first build that fails (a guarded build fails when it caught an out-of-range index):
# nothing else is started on the GPU after a failure.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
ARGS="${AB_ARGS:---steps 20 --warmup 3}"
for lib in "$@"; do
  RT_AMD_LIB="$R/$lib" timeout -k 10 150 python3 "$R/bench.py" $ARGS --no-cpu-baseline --no-extras --no-pipelined 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s %8.1f Mrays/s  %7.3f ms/frame  trace %7.3f ms  frac %.4f' % ('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['roofline']['frac']))
" || { echo "$lib FAILED"; exit 1; }
done
