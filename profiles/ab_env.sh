#!/bin/bash
# profiles/ab_env.sh VAR v1 v2 ... : bench the default library under different values of one env knob
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
VAR="$1"; shift
for v in "$@"; do
  env "$VAR=$v" timeout -k 10 120 python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-30s %8.1f Mrays/s  %7.3f ms/frame  trace %7.3f ms' % ('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame']))
" || echo "$VAR=$v FAILED"
done
