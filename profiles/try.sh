#!/bin/bash
# profiles/try.sh NAME : the order in which a kernel variant is allowed onto the GPU.
#   1. variants/NAME_g.so (the same change built with -DRT_GUARD=1) through the full-frame parity test and the
#      bench (rt_get_stats fails on any caught index); 2. only then variants/NAME.so through the same test and
#      the bench, beside the committed library.  Stops at the first failure.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
N="$1"; shift
cd "$R" || exit 1
T="tests/test_gpu_fullsize.py::test_sponza_1080p_equals_cpu_oracle tests/test_gpu_parity.py::test_golden_scenes tests/test_gpu_parity.py::test_light_counts_across_the_32_light_chunks"
timeout -k 10 300 python3 -u -m pytest $T -x -q --timeout 200 --timeout-method thread --rt-lib raytracer_amd/variants/${N}_g.so 2>&1 | tail -2 | head -1
[ "${PIPESTATUS[0]}" = 0 ] || { echo "guarded parity FAILED"; exit 1; }
bash profiles/ab.sh raytracer_amd/variants/${N}_g.so || exit 1
timeout -k 10 300 python3 -u -m pytest $T -x -q --timeout 200 --timeout-method thread --rt-lib raytracer_amd/variants/${N}.so 2>&1 | tail -2 | head -1
[ "${PIPESTATUS[0]}" = 0 ] || { echo "parity FAILED"; exit 1; }
bash profiles/ab.sh raytracer_amd/librt_amd.so raytracer_amd/variants/${N}.so raytracer_amd/librt_amd.so raytracer_amd/variants/${N}.so "$@"
