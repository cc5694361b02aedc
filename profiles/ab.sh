#!/bin/bash
# A/B several builds of librt_amd.so in one GPU session: profiles/ab.sh lib1.so lib2.so ...
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for lib in "$@"; do
  RT_AMD_LIB="$R/$lib" timeout -k 10 120 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s %8.1f Mrays/s  %7.3f ms/frame  trace %7.3f ms' % ('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame']))
" || echo "$lib FAILED"
done
