#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:
#   profiles/collect.sh <tag> [bench args...]
# One frame at a time (--frames-in-flight 1): with several in flight kernels of different frames overlap and the
# profiler's durations are no longer a kernel's own.
# Pass 0 is the kernel trace (+ --stats); every counter pass is its own run (PMC is never
# combined with a trace domain other than the kernel trace).  Raw output goes to
# gpurun_out/prof_<tag>/, summaries are written by profiles/summarize.py.
set -u
TAG="${1:-r01}"; shift || true
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipelined --frames-in-flight 1 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; exit 1; }
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$OUT/pmc$i.log"; exit 1; }
done
python3 "$R/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
echo "== derived (profiles/derive.py)" >> "$OUT/summary.txt"
python3 "$R/profiles/derive.py" "$OUT" >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
