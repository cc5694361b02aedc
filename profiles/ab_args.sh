#!/bin/bash
# profiles/ab_args.sh "args1" "args2" ... : bench the default library with different bench.py arguments
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for a in "$@"; do
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['config']['rays_per_frame']
print('%-40s %8.1f Mrays/s  %7.3f ms/frame  trace %7.3f ms  rays %.2fM' % ('$a', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], (r['primary']+r['shadow']+r['reflect'])/1e6))
" || echo "$a FAILED"
done
