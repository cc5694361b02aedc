#!/bin/bash
# profiles/try_main.sh : the committed sources' turn on the GPU, in the order kernel changes are allowed there:
# variants/guard.so (-DRT_GUARD=1) through the parity subset and the bench, then librt_amd.so through the same.
# Extra arguments are environment assignments for an A/B of the plain library at the end (e.g. RT_WF_PACKET=0).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R" || exit 1
T="tests/test_gpu_fullsize.py::test_sponza_1080p_equals_cpu_oracle tests/test_gpu_parity.py::test_golden_scenes tests/test_gpu_parity.py::test_light_counts_across_the_32_light_chunks"
timeout -k 10 300 python3 -u -m pytest $T -x -q --timeout 200 --timeout-method thread --rt-lib raytracer_amd/variants/guard.so 2>&1 | tail -2 | head -1
[ "${PIPESTATUS[0]}" = 0 ] || { echo "guarded parity FAILED"; exit 1; }
bash profiles/ab.sh raytracer_amd/variants/guard.so || exit 1
timeout -k 10 300 python3 -u -m pytest $T -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 | head -1
[ "${PIPESTATUS[0]}" = 0 ] || { echo "parity FAILED"; exit 1; }
bash profiles/ab.sh raytracer_amd/librt_amd.so || exit 1
for kv in "$@"; do
  echo "with $kv:"
  env "$kv" bash profiles/ab.sh raytracer_amd/librt_amd.so || exit 1
  bash profiles/ab.sh raytracer_amd/librt_amd.so || exit 1
done
